# round-2 (b): fused chain tests first, then the suite, then bench lines (fused vs call-by-call)
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02b; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_fused.py -x -q 2>&1 | tail -40 > $O/t_fused.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-hostpath > $O/bench_C3.json 2> $O/bench_C3.err
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-hostpath --call-by-call > $O/bench_C3_cbc.json 2> $O/bench_C3_cbc.err
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-hostpath --genes 6250 > $O/bench_C3_6250.json 2> $O/bench_C3_6250.err
timeout 300 python bench.py --config C2 --steps 10 --warmup 3 --no-cpu-baseline --no-hostpath > $O/bench_C2.json 2> $O/bench_C2.err
DSQ_BENCH_ONE_DEVICE=1 timeout 300 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-hostpath --genes 8000 > $O/bench_2rank.json 2> $O/bench_2rank.err
cat $O/t_fused.log; tail -n 5 $O/*.err
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], j["n_gpus"], round(j["value"]), round(j["ms_per_step"],2), j["config"]["chain"], {k:(round(v["avg_ms"],3)) for k,v in j["kernels"].items()}, {k:(round(v["avg_ms"],3)) for k,v in j["kernels_outlier_refit"].items()}, j.get("weak"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
