# folded logarithms in the IRLS constants, low-lane reductions in the collapsed QR: tests + bench
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02r; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $O/t.log
cat $O/t.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-hostpath > $O/bench_C3.json 2> $O/bench_C3.err
timeout 300 python bench.py --config C2 --steps 20 --warmup 5 --no-cpu-baseline --no-hostpath > $O/bench_C2.json 2> $O/bench_C2.err
timeout 300 python bench.py --config C5 --steps 10 --warmup 3 --no-cpu-baseline --no-hostpath > $O/bench_C5.json 2> $O/bench_C5.err
timeout 600 python bench.py --config C4 --steps 3 --warmup 1 --no-cpu-baseline --no-hostpath > $O/bench_C4.json 2> $O/bench_C4.err
tail -n 3 $O/*.err
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], j["n_gpus"], round(j["value"]), round(j["ms_per_step"],2), j["roofline"]["kernel"], round(j["roofline"]["frac"],5), {k:(round(v["avg_ms"],3)) for k,v in j["kernels"].items()})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
