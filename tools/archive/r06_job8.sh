#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r06j}; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_wide.py -m gpu --maxfail=8 -q > $O/tests_wide.log 2>&1; echo "wide tests rc=$? $(tail -1 $O/tests_wide.log)"
echo "== default"; DSQ_VERBOSE=1 timeout 900 python tools/widebench.py 20 32 40 48 31 46 2>&1 | grep -E "^p=|rolled" | sort -u
for nw in 1 2 4; do echo "== DSQ_WIDE_NW=$nw"; DSQ_WIDE_NW=$nw timeout 900 python tools/widebench.py 40 31 46 2>&1 | grep -E "^p=" ; done
