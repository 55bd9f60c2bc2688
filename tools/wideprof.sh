#!/bin/bash
# kernel stats (rocprofv3 --kernel-trace --stats) of tools/widebench.py for a few wide designs: durations, registers, LDS and
# scratch of the rolled kernels  ->  <out>/wide_stats.md
cd "$(dirname "$0")/.."
R=$PWD; O=${1:-$R/gpurun_out/wideprof}; shift || true; case $O in /*) ;; *) O=$R/$O;; esac
CASES=${*:-31 46 48 56}
mkdir -p "$O"
(cd /tmp && TMPDIR=/tmp timeout 900 rocprofv3 --kernel-trace --stats -d "$O/prof" -o trace -- python "$R/tools/widebench.py" $CASES > "$O/run.log" 2>&1)
grep "^p=" "$O/run.log"
python profiles/summarize_rocpd.py "$(find "$O/prof" -name '*.db' | head -1)" | head -24 > "$O/wide_stats.md"; cat "$O/wide_stats.md"
rm -rf "$O/prof"
