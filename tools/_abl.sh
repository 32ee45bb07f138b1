timeout 900 python -m pytest tests/test_hip_model.py -m gpu -x -q -k "sharded" 2>&1 | tail -8
timeout 600 python bench.py --steps 6 --warmup 2 2>&1 | tail -1 > gpurun_out/bench_default.json; cut -c1-600 gpurun_out/bench_default.json
timeout 300 python bench.py --config 1 --steps 3 --warmup 1 2>&1 | tail -1 | cut -c1-400
timeout 300 python bench.py --config 2 --steps 5 --warmup 2 2>&1 | tail -1 | cut -c1-400
timeout 300 python bench.py --global-batch 8 --steps 3 --warmup 1 --no-roofline 2>&1 | tail -1 | cut -c1-400
timeout 400 python bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400
