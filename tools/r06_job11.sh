#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r06n}; mkdir -p $O; cd $R; export TMPDIR=/tmp
run() { # name, env..., -- args
  name=$1; shift
  env "$@" > /dev/null 2>&1
}
for spec in "6250_key2_0:DSQ_LPT_KEY2=0:--genes=6250" "6250_key2_1:DSQ_LPT_KEY2=1:--genes=6250" "12500_k1:DSQ_LPT_KEY2=1:--genes=12500" "12500_off:DSQ_LPT=0:--genes=12500" "C2_k1_max:DSQ_LPT_KEY2=1,DSQ_LPT_MAXN=100000:--config=C2" "C2_off:DSQ_LPT=0:--config=C2" "C3_k1_max:DSQ_LPT_KEY2=1,DSQ_LPT_MAXN=100000:--config=C3" "C3_off:DSQ_LPT=0:--config=C3" "C5_k1_max:DSQ_LPT_KEY2=1,DSQ_LPT_MAXN=100000:--config=C5" "C5_off:DSQ_LPT=0:--config=C5"; do
  IFS=':' read -r name envs arg <<< "$spec"
  (export ${envs//,/ }; timeout 300 python bench.py $arg --pipeline 1 --no-cpu-baseline --no-hostpath --no-variants --no-parity --no-configs > $O/b_$name.json 2> $O/b_$name.err)
  python - <<PY
import json
try:
    j=json.loads(open("$O/b_$name.json").read().strip().splitlines()[-1])
    print("$name", "ms/step %.3f"%j["ms_per_step"], j["result_digest"][:12], {k:round(v["avg_ms"],3) for k,v in j["kernels"].items() if k in ("fit_beta","fit_disp","fit_beta_mle","fit_beta_prior")})
except Exception as e:
    print("$name", "FAILED", e)
PY
done
