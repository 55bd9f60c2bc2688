#!/bin/bash
# one-off (round 5): fixed per-gene cost vs cost per IRLS iteration / per search evaluation of the p = 4 fit kernels at the C3
# shape -- kbench with the trip count forced (DSQ_FORCE_ITERS), fit_beta and fit_disp launch times
cd "${GRAFT_REPO_ROOT:-.}"
for it in 1 2 4 8 12; do
  echo "== DSQ_FORCE_ITERS=$it"
  DSQ_FORCE_ITERS=$it python tools/kbench.py --reps 3 2>&1 | grep -v amdgpu.ids | tail -4
done
echo "== production"
python tools/kbench.py --reps 3 2>&1 | grep -v amdgpu.ids | tail -4
