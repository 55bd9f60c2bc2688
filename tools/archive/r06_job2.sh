#!/bin/bash
# round 6: chain tests + the 6 250-gene step (bench line, timeline) + the default C3 line
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r06b}; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_fused.py tests/test_gpu_pipeline.py tests/test_gpu_deseq_host.py tests/test_gpu_outliers.py tests/test_gpu_wide.py -m gpu --maxfail=5 -q -x > $O/tests.log 2>&1; echo "tests rc=$? $(tail -1 $O/tests.log)"
timeout 300 python bench.py --genes 6250 --pipeline 1 --no-cpu-baseline --no-hostpath --no-variants > $O/bench_6250.json 2> $O/bench_6250.err; echo "bench6250 rc=$?"
timeout 300 python bench.py --no-cpu-baseline --no-hostpath --no-variants --no-configs > $O/bench_C3.json 2> $O/bench_C3.err; echo "benchC3 rc=$?"
d=$O/prof6250; rm -rf $d
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $d -o trace -- python $R/bench.py --genes 6250 --steps 5 --warmup 2 --pipeline 1 --no-cpu-baseline --no-hostpath --no-variants --no-parity > $O/stats6250.log 2>&1)
python tools/timeline.py "$(find $d -name '*.db' | head -1)" > $O/timeline_6250.txt 2>&1
rm -rf $d
python - <<PY
import json
for f in ("bench_6250","bench_C3"):
    try:
        j=json.loads(open("$O/%s.json"%f).read().strip().splitlines()[-1])
        print(f, "ms/step %.3f"%j["ms_per_step"], "onecall", (j.get("one_call_at_a_time") or {}).get("ms_per_step"), j["result_digest"][:12], "parity", {k:j["parity"].get(k) for k in ("rows","replaced_rows","iter_equal","max_rel","error")} if j.get("parity") else None, "allocs", j["driver_allocs_in_timed_region"])
    except Exception as e:
        print(f, "FAILED", e)
PY
tail -3 $O/timeline_6250.txt
