#!/bin/bash
# one-off GPU job (round 4): resident waves per SIMD of the p = 4 fit kernels (launch-bounds A/B), kbench at the C3 shape
cd "${GRAFT_REPO_ROOT:-.}"
for lib in libdeseq2_mi355x libalt_disp2 libalt_beta2 libalt_beta4; do
  echo "=== $lib"
  DSQ_LIB=$PWD/deseq2_amd/$lib.so DSQ_VERBOSE=1 timeout 300 python tools/kbench.py --reps 4 2>&1 | grep -E "KBENCH|dsq\]" | sort | uniq | tail -6
done
