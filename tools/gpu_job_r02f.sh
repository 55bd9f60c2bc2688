set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02f; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wide.py tests/test_gpu_fused.py -x -q 2>&1 | tail -12 > $O/t1.log
cat $O/t1.log
timeout 400 python bench.py --config C4 --genes 15000 --steps 3 --warmup 1 --no-cpu-baseline --no-hostpath > $O/bench_C4_main.json 2> $O/bench_C4_main.err
DSQ_LIB=$R/deseq2_amd/libdeseq2_alt.so timeout 400 python bench.py --config C4 --genes 15000 --steps 3 --warmup 1 --no-cpu-baseline --no-hostpath > $O/bench_C4_alt.json 2> $O/bench_C4_alt.err
DSQ_LIB=$R/deseq2_amd/libdeseq2_alt.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "factor" 2>&1 | tail -5 > $O/t_alt.log
cat $O/t_alt.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-hostpath > $O/bench_C3.json 2> $O/bench_C3.err
tail -n 3 $O/*.err
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], j["n_gpus"], round(j["value"]), round(j["ms_per_step"],2), {k:(round(v["avg_ms"],3)) for k,v in j["kernels"].items()})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
