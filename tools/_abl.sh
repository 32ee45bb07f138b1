timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-exact-f32 2>&1 | tail -1 | cut -c1-200
