timeout 600 python tools/wgrad_l2_bench.py 2>&1 | tail -16 | cut -c1-190
