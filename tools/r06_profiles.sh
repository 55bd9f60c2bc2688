#!/bin/bash
# round 6: the profile set of the shipped library -- kernel stats + timeline (C3, C4, the 6 250-gene share), five PMC passes
# (C3, C4), the bench records of every config, the wide-design table.  Run through gpurun; results under gpurun_out/<TAG>/.
set -u
TAG=${1:-r06p}
bash tools/gpu_job.sh $TAG stats:C3 stats:C4 pmc:C3 pmc:C4 bench:C2 bench:C4 bench:C4R bench:C5 bench:C3:--genes=6250:--pipeline=1:name=C3_6250
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; cd $R
timeout 900 python tools/widebench.py > $O/wide.txt 2>&1; grep "^p=" $O/wide.txt
timeout 600 python tools/contbench.py > $O/general_path.txt 2>&1; tail -5 $O/general_path.txt
sha256sum deseq2_amd/libdeseq2_mi355x.so > $O/library.sha256
# the phase shares of the rolled wide kernels (make prof), when the profiling build travelled
if [ -f deseq2_amd/libdeseq2_prof.so ]; then
  DSQ_LIB=$R/deseq2_amd/libdeseq2_prof.so timeout 600 python tools/widebench.py 48 31 2>&1 | grep -E "(dispw|betaw)_prof\]" | sort | uniq > $O/wide_phases.txt; cat $O/wide_phases.txt
  # ... and of the two C3 fit kernels (fit_disp<4>, fit_beta_cell<4>): the full-size launches of one timed step
  DSQ_LIB=$R/deseq2_amd/libdeseq2_prof.so timeout 600 python bench.py --config C3 --steps 2 --warmup 1 --pipeline 1 --no-cpu-baseline --no-hostpath --no-variants --no-parity --no-configs 2> $O/phases_C3.raw > $O/phases_C3.json
  grep -E "_prof\]" $O/phases_C3.raw | grep -vE "\(([0-9]|[0-9][0-9]) Mcycles" | tail -4 > $O/phases_C3.txt; cat $O/phases_C3.txt
  python - <<PY
import json
d = json.loads(open("$O/phases_C3.json").read().strip().splitlines()[-1])
print("profiling build: ms_per_step", d["ms_per_step"], "kernel_sum_ms", d.get("kernel_sum_ms"))
PY
fi
# the default bench line with the PMC files of THIS library next to it (bench.py binds them by the library's sha256)
cp $O/pmc_C3.json profiles/r06_pmc_C3.json; cp $O/pmc_C4.json profiles/r06_pmc_C4.json
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.log; python - <<PY
import json
d = json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("default bench:", d["value"], d["unit"], d["ms_per_step"], "ms/step; roofline", d.get("roofline"), "; pmc_matches_library", d.get("pmc_matches_library", d.get("roofline", {}).get("pmc_matches_library")))
PY
