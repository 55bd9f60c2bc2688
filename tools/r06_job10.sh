#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r06m}; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_pipeline.py tests/test_gpu_deseq_host.py -m gpu --maxfail=5 -q -x > $O/tests.log 2>&1; echo "tests rc=$? $(tail -1 $O/tests.log)"
for lpt in 1 0; do
  DSQ_LPT=$lpt timeout 300 python bench.py --genes 6250 --pipeline 1 --no-cpu-baseline --no-hostpath --no-variants --no-parity > $O/bench_6250_lpt$lpt.json 2> $O/bench_6250_lpt$lpt.err
  DSQ_LPT=$lpt timeout 300 python bench.py --config C2 --pipeline 1 --no-cpu-baseline --no-hostpath --no-variants --no-parity > $O/bench_C2_lpt$lpt.json 2> $O/bench_C2_lpt$lpt.err
done
python - <<PY
import json
for f in ("bench_6250_lpt1","bench_6250_lpt0","bench_C2_lpt1","bench_C2_lpt0"):
    try:
        j=json.loads(open("$O/%s.json"%f).read().strip().splitlines()[-1])
        print(f, "ms/step %.3f"%j["ms_per_step"], j["result_digest"][:12], {k:round(v["avg_ms"],3) for k,v in j["kernels"].items() if k in ("fit_beta","fit_disp")}, j["step_ms"])
    except Exception as e:
        print(f, "FAILED", e)
PY
