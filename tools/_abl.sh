cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r02a -o r02a -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/prof_r02a.log 2>&1
tail -2 $GRAFT_REPO_ROOT/gpurun_out/prof_r02a.log | cut -c1-300
find $GRAFT_REPO_ROOT/gpurun_out/prof_r02a -name "*kernel_stats*" | head
