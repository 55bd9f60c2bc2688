# closed-form deviance in the general fitBeta kernel: full GPU suite, general-path timings before / after
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02s; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $O/t.log
cat $O/t.log
DSQ_LIB=$R/deseq2_amd/_variants/prev.so timeout 300 python tools/contbench.py 2>&1 | grep -v amdgpu > $O/cont_prev.log
timeout 300 python tools/contbench.py 2>&1 | grep -v amdgpu > $O/cont_new.log
cat $O/cont_prev.log $O/cont_new.log
