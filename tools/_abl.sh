timeout 300 python tools/wgrad_l2_bench.py 2>&1 | grep Cin | cut -c1-150
timeout 600 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "conv_fwd_bwd or full_size or limb" 2>&1 | tail -3
