# final round-2 state: full GPU suite, HIP-vs-reference parity table, bench lines of all four configs (cpu baselines, hostpath), small-n and 2-rank lines
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02p; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $O/t.log
cat $O/t.log
timeout 300 python tools/parity_report.py hip > $O/parity_hip.md 2> $O/parity_hip.err
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_C3.json 2> $O/bench_C3.err
timeout 300 python bench.py --config C2 --steps 20 --warmup 5 > $O/bench_C2.json 2> $O/bench_C2.err
timeout 400 python bench.py --config C5 --steps 10 --warmup 3 > $O/bench_C5.json 2> $O/bench_C5.err
timeout 900 python bench.py --config C4 --steps 3 --warmup 1 > $O/bench_C4.json 2> $O/bench_C4.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-hostpath --genes 6250 > $O/bench_C3_6250.json 2> $O/bench_C3_6250.err
DSQ_BENCH_ONE_DEVICE=1 timeout 300 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-hostpath > $O/bench_2rank.json 2> $O/bench_2rank.err
tail -n 3 $O/*.err
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    cb=j.get("cpu_baseline",{})
    print(sys.argv[1].split('/')[-1], j["n_gpus"], round(j["value"]), round(j["ms_per_step"],2), j["roofline"]["kernel"], round(j["roofline"]["frac"],5), {k:(round(v["avg_ms"],3)) for k,v in j["kernels"].items()}, "hostpath", j.get("hostpath_ms"), "cpu", cb.get("value"), cb.get("all_cores",{}).get("value"), cb.get("all_cores",{}).get("cores"), j.get("weak"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
