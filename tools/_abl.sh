for d in 6 0; do echo "== ONDA_L2_DEBUG=$d"; ONDA_L2_DEBUG=$d timeout 300 python tools/conv_l2_bench.py 2>&1 | grep Cin | cut -c1-175; done
timeout 200 python tools/l2_stamps.py 2>&1 | grep Cin
