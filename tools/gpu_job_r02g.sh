set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02g; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_pipeline.py -x -q 2>&1 | tail -12 > $O/t1.log
cat $O/t1.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-hostpath > $O/bench_C3.json 2> $O/bench_C3.err
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-hostpath --genes 6250 > $O/bench_C3_6250.json 2> $O/bench_C3_6250.err
DSQ_BENCH_ONE_DEVICE=1 timeout 300 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-hostpath --genes 12500 > $O/bench_2rank.json 2> $O/bench_2rank.err
tail -n 3 $O/*.err
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], j["n_gpus"], round(j["value"]), round(j["ms_per_step"],2), {k:(round(v["avg_ms"],3)) for k,v in j["kernels"].items()}, j.get("weak"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
