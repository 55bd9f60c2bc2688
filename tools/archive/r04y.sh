#!/bin/bash
# one-off GPU job (round 4): A/B of alternate libraries on bench.py configs -- r04y.sh "LIBS" "CONFIGS"
cd "${GRAFT_REPO_ROOT:-.}"
for rep in 1 2; do
for lib in $1; do
  for cfg in $2; do
  DSQ_LIB=$PWD/deseq2_amd/$lib.so timeout 300 python bench.py --config $cfg --steps 10 --warmup 2 --no-cpu-baseline --no-hostpath --no-variants --no-parity 2> /dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib $cfg step %.3f ms' % j['ms_per_step'], {k:round(v['avg_ms'],3) for k,v in j['kernels'].items() if k in ('fit_disp','fit_beta','nbinom_loglike')}, j['result_digest'][:10])"
  done
done
done
