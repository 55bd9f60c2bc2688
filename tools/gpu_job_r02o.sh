# profiles of the final round-2 kernels: C3 and C4 (kernel trace + stats, three PMC passes each)
set -x
cd $GRAFT_REPO_ROOT
TAG=r02o_C3 CFG=C3 bash tools/gpu_profile_r02.sh > gpurun_out/r02o_C3.log 2>&1
TAG=r02o_C4 CFG=C4 bash tools/gpu_profile_r02.sh > gpurun_out/r02o_C4.log 2>&1
tail -5 gpurun_out/r02o_C3.log gpurun_out/r02o_C4.log
head -12 gpurun_out/r02o_C3/kernel_stats.md
