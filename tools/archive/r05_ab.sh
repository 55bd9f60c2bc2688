#!/bin/bash
# one-off (round 5): parity tests of the fit kernels + bench A/B with an alternate library -- r05_ab.sh "LIBS" "CONFIGS"
cd "${GRAFT_REPO_ROOT:-.}"
export DSQ_LIB=$PWD/deseq2_amd/libdeseq2_test.so
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge.py tests/test_gpu_fused.py tests/test_gpu_pipeline.py -m gpu -q -x 2>&1 | tail -5
unset DSQ_LIB
bash tools/r04y.sh "$1" "$2"
