#!/bin/bash
# one-off GPU job (round 4): the general fit_beta with the least-squares rows in registers (QRM 2) and the batched wave
# reductions, against the build before (DSQ_BETA_QRREG=0: rows in LDS from p = 10, replay below) and an alternate library
# (p = 7..9 at one wave per SIMD)
cd "${GRAFT_REPO_ROOT:-.}"
S=${1:-3,4,5,6,7,8,10,0,2}
echo "=== tree"
CONTBENCH_ONLY=$S DSQ_VERBOSE=1 timeout 800 python tools/contbench.py 2>&1 | grep -E "^p=|fit_beta<" | sort | uniq
echo "=== DSQ_BETA_QRREG=0"
CONTBENCH_ONLY=$S DSQ_BETA_QRREG=0 timeout 800 python tools/contbench.py 2>&1 | grep -E "^p="
if [ -f deseq2_amd/libalt_qrreg1.so ]; then
echo "=== p = 7..9 at one wave per SIMD"
CONTBENCH_ONLY=6,7,8 DSQ_LIB=$PWD/deseq2_amd/libalt_qrreg1.so timeout 800 python tools/contbench.py 2>&1 | grep -E "^p="
fi
