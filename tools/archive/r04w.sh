#!/bin/bash
# one-off GPU job (round 4): contrast_denom = cd instead of sqrt(cd) in the five-trip register builds (p = 16 / 24)
cd "${GRAFT_REPO_ROOT:-.}"
T="tests/test_gpu_wide.py::test_general_path_kernels_match_oracle tests/test_gpu_fuzz.py::test_fuzzed_configuration[4045]"
for lib in libdeseq2_mi355x libalt_A libalt_B; do
  echo "=== $lib"
  DSQ_LIB=$PWD/deseq2_amd/$lib.so timeout 600 python -m pytest $T -m gpu -q 2>&1 | tail -4
done
