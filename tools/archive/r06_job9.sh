#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r06k}; mkdir -p $O; cd $R; export TMPDIR=/tmp
DSQ_ROLLED_MINP=5 timeout 900 python -m pytest tests/test_gpu_wide.py::test_general_path_kernels_match_oracle tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -4
echo "== default"; CONTBENCH_ONLY=0,1,2,3,7,8,9,10,11 timeout 600 python tools/contbench.py 2>&1 | grep "^p="
echo "== rolled from p=5"; DSQ_ROLLED_MINP=5 DSQ_VERBOSE=1 CONTBENCH_ONLY=0,1,2,3,7,8,9,10,11 timeout 600 python tools/contbench.py 2>&1 | grep -E "^p=|rolled" | sort -u
