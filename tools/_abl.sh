timeout 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_default.json; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_default.json')); c=d['config']
print(d['value'], d['ms_per_step'], c.get('eager_rocm_committed'), c.get('exact_f32'))
PY
