#!/bin/bash
# one-off GPU job (round 4): the LU::solve probe, the select form inside the real kernels (alt build), the D2H probe
cd "${GRAFT_REPO_ROOT:-.}"
echo "=== lu_probe"; timeout 300 tools/lu_probe; echo "rc=$?"
echo "=== parity tests on the alt library (LU_SELECT at every width in fit_beta / fit_disp p = 5, 6, 10)"
DSQ_LIB=$PWD/deseq2_amd/libdeseq2_alt.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge.py tests/test_gpu_wide.py tests/test_gpu_fuzz.py -m gpu -q 2>&1 | tail -30
echo "=== d2h_probe"; timeout 300 tools/d2h_probe; echo "rc=$?"
