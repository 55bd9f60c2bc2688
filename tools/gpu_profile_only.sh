set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
B2="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
rm -rf $O/prof_r1d $O/pmcd1 $O/pmcd2 $O/pmcd3
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_r1d -o r1d -- $B > $O/prof_r1d.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmcd1 -o p -- $B2 > $O/pmcd1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmcd2 -o p -- $B2 > $O/pmcd2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmcd3 -o p -- $B2 > $O/pmcd3.log 2>&1
cd $R && timeout 400 python bench.py --steps 10 --warmup 3 > $O/bench_r1d.json 2> $O/bench_r1d.err
tail -2 $O/prof_r1d.log; ls $O/prof_r1d $O/pmcd1 $O/pmcd2 $O/pmcd3; cat $O/bench_r1d.json
