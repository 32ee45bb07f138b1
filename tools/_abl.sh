timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r02b -o r02b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/prof_r02b.log 2>&1
