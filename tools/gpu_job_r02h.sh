set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02h; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_edge.py tests/test_gpu_parity.py -x -q 2>&1 | tail -12 > $O/t1.log
cat $O/t1.log
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_C3.json 2> $O/bench_C3.err
DSQ_HOST_SHARDS=4 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_C3_s4.json 2> $O/bench_C3_s4.err
tail -n 3 $O/*.err
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], round(j["value"]), round(j["ms_per_step"],2), j.get("hostpath"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
