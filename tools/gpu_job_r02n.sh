# experiments: fitDisp cell mode from p = 2 (timing only), small-n step
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02n; mkdir -p $O
cd $R
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-hostpath > $O/bench_C3.json 2> $O/bench_C3.err
DSQ_DISP_CELL_MINP=2 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-hostpath > $O/bench_C3_cell2.json 2> $O/bench_C3_cell2.err
DSQ_DISP_CELL_MINP=2 timeout 300 python bench.py --config C2 --steps 20 --warmup 5 --no-cpu-baseline --no-hostpath > $O/bench_C2_cell2.json 2> $O/bench_C2_cell2.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-hostpath --genes 6250 > $O/bench_C3_6250.json 2> $O/bench_C3_6250.err
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], j["n_gpus"], round(j["value"]), round(j["ms_per_step"],2), {k:(round(v["avg_ms"],3)) for k,v in j["kernels"].items()}, {k:(round(v["avg_ms"],3)) for k,v in j["kernels_outlier_refit"].items()})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
