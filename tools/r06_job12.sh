#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r06q}; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_fused.py tests/test_gpu_pipeline.py tests/test_gpu_deseq_host.py tests/test_gpu_testthat.py -m gpu --maxfail=5 -q -x > $O/tests.log 2>&1; echo "tests rc=$? $(tail -1 $O/tests.log)"
for spec in "6250:--genes=6250" "C3:--config=C3" "C2:--config=C2" "C5:--config=C5" "C4:--config=C4"; do
  IFS=':' read -r name arg <<< "$spec"
  timeout 300 python bench.py $arg --pipeline 1 --no-cpu-baseline --no-hostpath --no-variants --no-configs > $O/b_$name.json 2> $O/b_$name.err
  python - <<PY
import json
try:
    j=json.loads(open("$O/b_$name.json").read().strip().splitlines()[-1])
    print("$name", "ms/step %.3f"%j["ms_per_step"], j["result_digest"][:12], {k:round(v["avg_ms"],3) for k,v in j["kernels"].items() if k in ("fit_beta","fit_disp","trend_fit","prior_var")}, {k:j["parity"].get(k) for k in ("rows","iter_equal","max_rel")})
except Exception as e:
    print("$name", "FAILED", e)
PY
done
