#!/bin/bash
# one-off GPU job (round 4): scheduler / optimisation-level variants of the p = 4 fit kernels (same source, same IEEE semantics)
# a max-ilp strategy, b max-memory-clause, c schedule-metric-bias 100, d prealloc-sgpr-spill-vgprs, e amdgpu RP trackers, f -O2, g bias 0
cd "${GRAFT_REPO_ROOT:-.}"
for lib in libdeseq2_mi355x libalt_h libalt_i libalt_j libalt_k libdeseq2_mi355x libalt_k; do
  echo "=== $lib $(DSQ_LIB=$PWD/deseq2_amd/$lib.so timeout 300 python tools/kbench.py --reps 5 2>&1 | grep KBENCH | cut -c1-110)"
done
