#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r06d}; mkdir -p $O; cd $R; export TMPDIR=/tmp
for w in 3 4 6; do
  timeout 300 python bench.py --warmup $w --no-cpu-baseline --no-hostpath --no-variants --no-configs --no-parity > $O/bench_w$w.json 2> $O/bench_w$w.err
  python - <<PY
import json
j=json.loads(open("$O/bench_w$w.json").read().strip().splitlines()[-1])
print("warmup $w", "ms/step %.3f"%j["ms_per_step"], j["driver_allocs_in_timed_region"], j["step_ms"], (j.get("one_call_at_a_time") or {}).get("ms_per_step"))
PY
done
