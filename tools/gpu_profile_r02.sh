# rocprofv3 evidence for the round-2 kernels: kernel trace + stats, then separate PMC passes (SQ counters, FETCH_SIZE,
# WRITE_SIZE) of the same bench command; TAG names the output directory under gpurun_out/
set -x
TAG=${TAG:-r02p}; CFG=${CFG:-C3}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-hostpath"
B2="python $R/bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --no-hostpath"
rm -rf $O/prof $O/pmc1 $O/pmc2 $O/pmc3
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o t -- $B > $O/prof.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc1 -o p -- $B2 > $O/pmc1.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc2 -o p -- $B2 > $O/pmc2.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc3 -o p -- $B2 > $O/pmc3.log 2>&1
cd $R
python profiles/summarize_rocpd.py $(find $O/prof -name "*.db" | head -1) > $O/kernel_stats.md 2> $O/kernel_stats.err
P1=$(dirname $(find $O/pmc1 -name "p_counter_collection.csv" | head -1)); P2=$(dirname $(find $O/pmc2 -name "p_counter_collection.csv" | head -1)); P3=$(dirname $(find $O/pmc3 -name "p_counter_collection.csv" | head -1))
NG=$(python -c "import bench; print(bench.CONFIGS['$CFG']['genes'])")
python tools/pmc_summary.py $P2 $P3 $P1 $O/pmc.json $NG "rocprofv3 --pmc passes (SQ_*, FETCH_SIZE, WRITE_SIZE separately) of bench.py --config $CFG --steps 2 --warmup 1, state $TAG; means over the launches with the largest grid of each kernel; FETCH_SIZE x2 (gfx950 note) + WRITE_SIZE; tools/gpu_profile_r02.sh" > $O/pmc_summary.log 2>&1
# keep the merge small: drop the raw traces
rm -rf $O/prof $O/pmc1 $O/pmc2 $O/pmc3
head -30 $O/kernel_stats.md; tail -5 $O/pmc_summary.log; tail -2 $O/prof.log
