timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "limb_planes or batchnorm or conv_fwd_bwd" 2>&1 | tail -15
