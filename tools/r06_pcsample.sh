#!/bin/bash
# round 6: rocprofv3 PC sampling (beta, host-trap method) of the C3 step on the line-table build of the p = 4 fit kernels
# (libdeseq2_alt.so: the production objects, fit_disp_p4 / fit_beta_p4 recompiled with -gline-tables-only) -> samples per
# source line of the two dominant kernels.  Bounded: 3 steps, its own timeout.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r06s}; mkdir -p $O; cd $R; export TMPDIR=/tmp
d=$O/pcs; rm -rf $d
(cd /tmp && DSQ_LIB=$R/deseq2_amd/libdeseq2_alt.so timeout 300 rocprofv3 --kernel-trace --pc-sampling-beta-enabled --pc-sampling-unit time --pc-sampling-method host_trap --pc-sampling-interval ${2:-20} --output-format csv -d $d -o pcs -- python $R/bench.py --steps 3 --warmup 1 --pipeline 1 --no-cpu-baseline --no-hostpath --no-variants --no-parity --no-configs > $O/pcs.log 2>&1); echo "pc sampling rc=$?"
tail -5 $O/pcs.log
find $d -type f | head; for f in $(find $d -name "*pc_sampling*.csv" | head -2); do echo $f; wc -l $f; head -3 $f; done
python tools/pcs_summary.py $d > $O/hot_regions.txt 2>&1; head -70 $O/hot_regions.txt
rm -rf $d
