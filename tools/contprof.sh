#!/bin/bash
# kernel stats of tools/contbench.py for one shape (CONTBENCH_ONLY): registers / LDS / scratch of the general kernels
cd "$(dirname "$0")/.."
R=$PWD; O=${1:-$R/gpurun_out/contprof}; mkdir -p "$O"
export CONTBENCH_ONLY=${CONTBENCH_ONLY:-3}
(cd /tmp && TMPDIR=/tmp timeout 600 rocprofv3 --kernel-trace --stats -d "$O/prof" -o trace -- python "$R/tools/contbench.py" > "$O/run.log" 2>&1)
python profiles/summarize_rocpd.py "$(find "$O/prof" -name '*.db' | head -1)" | head -16
rm -rf "$O/prof"
