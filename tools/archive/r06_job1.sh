#!/bin/bash
# round 6, first GPU job: the GPU suite, the default bench line, a kernel timeline of one rank's share (6 250 genes)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06a; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu --maxfail=10 -q -x > $O/tests.log 2>&1; echo "tests rc=$? $(tail -1 $O/tests.log)"
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
timeout 300 python bench.py --genes 6250 --pipeline 1 --no-cpu-baseline --no-hostpath --no-variants > $O/bench_6250.json 2> $O/bench_6250.err; echo "bench6250 rc=$?"
d=$O/prof6250; rm -rf $d
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $d -o trace -- python $R/bench.py --genes 6250 --steps 5 --warmup 2 --pipeline 1 --no-cpu-baseline --no-hostpath --no-variants --no-parity > $O/stats6250.log 2>&1)
python tools/timeline.py "$(find $d -name '*.db' | head -1)" > $O/timeline_6250.txt 2>&1
python profiles/summarize_rocpd.py "$(find $d -name '*.db' | head -1)" > $O/stats_6250.md 2>> $O/stats6250.log
rm -rf $d
tail -3 $O/timeline_6250.txt
