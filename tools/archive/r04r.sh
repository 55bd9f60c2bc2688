#!/bin/bash
# one-off GPU job (round 4): resident waves of the p = 10 kernels at C4 (registers AND LDS lifted together)
cd "${GRAFT_REPO_ROOT:-.}"
for lib in libdeseq2_mi355x libalt_disp_g2 libalt_disp_g3 libalt_disp_g4 libalt_beta_3 libalt_beta_4; do
  echo "=== $lib"
  DSQ_LIB=$PWD/deseq2_amd/$lib.so DSQ_VERBOSE=1 timeout 300 python bench.py --config C4 --steps 4 --warmup 1 --no-cpu-baseline --no-hostpath 2> /tmp/err.txt | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('   step %.2f ms' % j['ms_per_step'], {k:round(v['avg_ms'],2) for k,v in j['kernels'].items() if k in ('fit_disp','fit_beta')}, j['result_digest'][:10])"
  grep "fit_disp<P=10,mode=0>\|fit_beta_cells<P=10>" /tmp/err.txt | sort | uniq
done
