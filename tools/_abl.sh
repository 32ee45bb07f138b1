for x in 1 0; do echo "== ONDA_L2_NOSKIP=$x"; ONDA_L2_NOSKIP=$x QUICK=1 timeout 300 python tools/conv_l2_bench.py 2>&1 | grep Cin | cut -c1-120; done
