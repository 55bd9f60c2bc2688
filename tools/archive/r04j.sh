#!/bin/bash
# one-off GPU job (round 4): stored-row QR of the general fit_beta at p = 16 / 24 (two resident waves per CU instead of replay)
cd "${GRAFT_REPO_ROOT:-.}"
echo "=== stored rows allowed at low occupancy (tree)"
CONTBENCH_ONLY=3,4,5 DSQ_VERBOSE=1 timeout 600 python tools/contbench.py 2>&1 | grep -E "^p=|fit_beta<" | sort | uniq
echo "=== replay (round 3: stored rows only from 4 waves per CU)"
CONTBENCH_ONLY=4,5 DSQ_BETA_QRROWS_MINWPC=4 timeout 600 python tools/contbench.py 2>&1 | grep -E "^p="
