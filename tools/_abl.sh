cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r02d -o r02d -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-exact-f32 > $R/gpurun_out/prof_r02d.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -o f -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-exact-f32 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -o w -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-exact-f32 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_sq -o sq -- python3 $R/tools/one_pass.py > /dev/null 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/r02_d_hbm_traffic.json | head -10
python tools/sq_summary.py $(find gpurun_out/pmc_sq -name "*counter_collection.csv" | head -1) 12 | tee gpurun_out/r02_d_sq_counters.txt
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq
timeout 400 python bench.py --steps 8 --warmup 2 2>&1 | tail -1 > gpurun_out/r02_d_bench_line.json
timeout 300 python bench.py --branch static --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-exact-f32 2>&1 | tail -1 > gpurun_out/r02_d_bench_line_static_branch.json
timeout 300 python bench.py --config 1 --steps 3 --warmup 1 --no-roofline 2>&1 | tail -1 > gpurun_out/r02_d_bench_line_config1.json
timeout 300 python bench.py --config 2 --steps 6 --warmup 2 --no-roofline 2>&1 | tail -1 > gpurun_out/r02_d_bench_line_config2.json
timeout 400 python bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-exact-f32 2>&1 | tail -1 > gpurun_out/r02_d_bench_line_config5.json
timeout 300 python bench.py --global-batch 32 --steps 2 --warmup 1 --no-roofline 2>&1 | tail -1 > gpurun_out/r02_d_bench_line_strong_gb32_n1.json
for f in gpurun_out/r02_d_bench_line*.json; do python -c "
import json,sys; d=json.load(open('$f')); print('$f', d['value'], d['unit'], d['ms_per_step'])"; done
