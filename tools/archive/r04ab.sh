#!/bin/bash
# one-off GPU job (round 4): the general fit_beta with eight trips of rows in registers (QRM 3, 256 < m + p <= 512, p <= 10)
cd "${GRAFT_REPO_ROOT:-.}"
S=${1:-1,9,10}
echo "=== tree"
CONTBENCH_ONLY=$S DSQ_VERBOSE=1 timeout 800 python tools/contbench.py 2>&1 | grep -E "^p=|fit_beta<.*stored-rows=3" | sort | uniq
echo "=== DSQ_BETA_QRREG=2 (no eight-trip form: replay at p = 7..9, rows in LDS at p = 10)"
CONTBENCH_ONLY=$S DSQ_BETA_QRREG=2 timeout 800 python tools/contbench.py 2>&1 | grep -E "^p="
