# round-2 first GPU job: full GPU test suite, parity report, bench lines of all four configs (kernels of round 1)
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02a; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/t.log
timeout 300 python tools/parity_report.py hip > $O/parity_hip.md 2> $O/parity_hip.err
timeout 400 python bench.py --steps 10 --warmup 3 > $O/bench_C3.json 2> $O/bench_C3.err
timeout 300 python bench.py --config C2 --steps 10 --warmup 3 > $O/bench_C2.json 2> $O/bench_C2.err
timeout 400 python bench.py --config C5 --steps 5 --warmup 2 > $O/bench_C5.json 2> $O/bench_C5.err
timeout 120 python bench.py --profile-host --no-cpu-baseline > $O/hostprofile_C3.txt 2>&1
cat $O/t.log; tail -3 $O/*.err; for f in $O/bench_C*.json; do python - "$f" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], j["value"], j["ms_per_step"], j["roofline"]["kernel"], j["roofline"]["frac"], {k:(round(v["avg_ms"],3)) for k,v in j["kernels"].items()}, j.get("hostpath_ms"), j.get("cpu_baseline",{}).get("value"), j.get("cpu_baseline",{}).get("all_cores",{}))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
