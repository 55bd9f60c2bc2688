#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r06f}; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_wide.py -m gpu --maxfail=8 -q > $O/tests_wide.log 2>&1; echo "wide tests rc=$? $(tail -1 $O/tests_wide.log)"
grep -E "FAILED|Error|error" $O/tests_wide.log | head -20
DSQ_VERBOSE=1 timeout 900 python tools/widebench.py > $O/wide.txt 2>&1; grep -E "^p=|rolled" $O/wide.txt
