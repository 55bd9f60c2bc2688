#!/bin/bash
# one-off GPU job (round 4): which variant of LU::solve's pivot swap breaks the round-2 fit_disp<6> kernel (commit 9b95dfa)?
# v0 select chain (the form that 'came out wrong'), v1 pairwise select, v2 select chain + wave-uniform pivot row (readfirstlane),
# v3 select chain + opaque identity columns (no constant folding), v4 conditional swap (what the commit shipped), v5 = v0 with
# SGPR spills to memory instead of VGPR lanes
cd "${GRAFT_REPO_ROOT:-.}/_old9b"
for v in 0 1 2 3 4 5; do
  echo "=== variant v$v"
  DSQ_LIB=$PWD/deseq2_amd/libold_v$v.so timeout 300 python -m pytest "tests/test_gpu_parity.py::test_fit_disp_matches_oracle" tests/test_gpu_edge.py::test_seeded_shape_sweep -m gpu -q 2>&1 | tail -6
done
