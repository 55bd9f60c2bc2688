#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r06g}; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_wide.py -m gpu --maxfail=8 -q > $O/tests_wide.log 2>&1; echo "wide tests rc=$? $(tail -1 $O/tests_wide.log)"
for mode in -1 0 1; do
  echo "== DSQ_WIDE_LDS=$mode"
  DSQ_WIDE_LDS=$mode DSQ_VERBOSE=1 timeout 900 python tools/widebench.py 20 40 48 31 46 2>&1 | grep -E "^p=|rolled" | sort -u
done
