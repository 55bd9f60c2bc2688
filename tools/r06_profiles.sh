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
