#!/bin/bash
# one-off GPU job (round 4): the LU probe (fixed), the round-2 commit 9b95dfa rebuilt with the select form up to P = 6
cd "${GRAFT_REPO_ROOT:-.}"
echo "=== lu_probe"; timeout 300 tools/lu_probe | grep -v "mismatches    0  det/trace mismatches    0" ; echo "rc=${PIPESTATUS[0]}"
echo "=== commit 9b95dfa with -DDSQ_LU_SELECT_MAXP=6 (the state that 'came out wrong'): its own parity tests"
(cd _old9b && timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge.py tests/test_gpu_wide.py -m gpu -q -x --maxfail=5 2>&1 | tail -40)
