set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02q; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o t -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-hostpath > $O/prof.log 2>&1
cd $R
python tools/timeline.py $(find $O/prof -name "*.db" | head -1) > $O/timeline.txt 2>&1
rm -rf $O/prof
tail -4 $O/timeline.txt
