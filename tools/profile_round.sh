#!/bin/bash
# One evidence run for profiles/ (run on the GPU box through gpurun): bench lines of every configuration, the rocprofv3
# kernel-trace summary of the default bench command, the two PMC traffic passes and the SQ counter pass (each in its OWN
# run, with --kernel-trace only, as MI355X_MICROARCH.md prescribes), per-shape conv rates.
#   tools/profile_round.sh r03_a      -> gpurun_out/r03_a_*
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
tag=${1:-r05_x}; G=gpurun_out; mkdir -p $G
common="--no-cpu-baseline --no-eager --no-other-configs --no-h2d-leg --no-exchange-probe"
# the headline line: bench.py alone, nothing wrapped around it (round-5 advisor); board power is logged in a run of its own below
python3 bench.py --steps 20 --warmup 5 > $G/${tag}_bench_line.json 2> $G/${tag}_bench_line.err
python3 tools/power_log.py $G/${tag}_power_during_bench.csv -- python3 bench.py --steps 40 --warmup 5 $common --no-exact-f32 --no-roofline > /dev/null 2> $G/${tag}_power_summary.txt
rm -f $G/${tag}_power_during_bench.csv   # (20 ms samples, thousands of lines: the one-line summary on stderr is what is kept)
python3 bench.py --steps 10 --warmup 3 --branch static $common --no-exact-f32 > $G/${tag}_bench_line_static_branch.json 2>/dev/null
python3 bench.py --steps 10 --warmup 3 --config 1 $common > $G/${tag}_bench_line_config1.json 2>/dev/null
python3 bench.py --steps 10 --warmup 3 --config 1 --eval-batch 1 $common > $G/${tag}_bench_line_config1_one_frame_at_a_time.json 2>/dev/null
python3 bench.py --steps 10 --warmup 3 --config 2 $common > $G/${tag}_bench_line_config2.json 2>/dev/null
python3 bench.py --steps 5 --warmup 2 --config 5 $common --no-exact-f32 > $G/${tag}_bench_line_config5.json 2>/dev/null
python3 bench.py --steps 3 --warmup 1 --global-batch 32 $common --no-exact-f32 --no-roofline > $G/${tag}_bench_line_strong_gb32_n1.json 2>/dev/null
python3 tools/conv_shapes.py > $G/${tag}_conv_shapes.txt 2>&1
rm -rf $G/prof_$tag && mkdir -p $G/prof_$tag
rocprofv3 --kernel-trace --stats -d $G/prof_$tag/kt -- python3 bench.py --steps 10 --warmup 3 $common --no-exact-f32 --no-roofline > $G/${tag}_profiled_bench_line.json 2> $G/prof_$tag/kt.err
python3 tools/kstats.py $G/prof_$tag/kt > $G/${tag}_bench_kernel_stats.csv
# the same command with every pass on ONE stream: per-kernel durations of kernels that have the GPU to themselves (what the
# bench line's `roofline` is measured on; beside other streams' launches a kernel's duration stretches)
ONDA_SIDE_STREAMS=0 rocprofv3 --kernel-trace --stats -d $G/prof_$tag/kt1 -- python3 bench.py --steps 10 --warmup 3 $common --no-exact-f32 --no-roofline > $G/${tag}_profiled_bench_line_one_stream.json 2> $G/prof_$tag/kt1.err
python3 tools/kstats.py $G/prof_$tag/kt1 > $G/${tag}_bench_kernel_stats_one_stream.csv
# per-step kernel table (launches and time per kernel and step, idle time) from a kernel trace of the one-stream schedule
ONDA_SIDE_STREAMS=0 rocprofv3 --kernel-trace --output-format csv -d $G/prof_$tag/tr -o t -- python3 bench.py --steps 8 --warmup 3 $common --no-exact-f32 --no-roofline > /dev/null 2> $G/prof_$tag/tr.err
KERNELS=40 python3 tools/trace_gaps.py $(find $G/prof_$tag/tr -name "*kernel_trace.csv" | head -1) > $G/${tag}_step_kernel_table.txt 2>&1
# HBM-side traffic ON THE BENCH STEP ITSELF (round-4 verdict: bytes and flops per launch of one `roofline` block must describe the
# same launches), every pass on one stream as the roofline leg measures it; separate --pmc passes with --kernel-trace only
pmc_cmd="bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-eager --no-other-configs --no-exact-f32 --no-roofline --no-h2d-leg --no-exchange-probe"
ONDA_SIDE_STREAMS=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $G/prof_$tag/fetch -- python3 $pmc_cmd > /dev/null 2> $G/prof_$tag/fetch.err
ONDA_SIDE_STREAMS=0 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $G/prof_$tag/write -- python3 $pmc_cmd > /dev/null 2> $G/prof_$tag/write.err
python3 tools/pmc_summary.py $G/prof_$tag/fetch $G/prof_$tag/write $G/${tag}_hbm_traffic.json "ONDA_SIDE_STREAMS=0 python3 $pmc_cmd (the bench step: 5 steps + set-up passes)" > $G/${tag}_hbm_traffic_top.txt
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $G/prof_$tag/sq -o sq -- python3 tools/one_pass.py > /dev/null 2> $G/prof_$tag/sq.err
python3 tools/sq_summary.py $(find $G/prof_$tag/sq -name "*counter_collection.csv" | head -1) 12 > $G/${tag}_sq_counters.txt 2>&1
# the exact-fp32 mode (ONDA_CONV_MODE=f32): bench line, per-shape rates, SQ counters, board power -- each in a run of its own
ONDA_CONV_MODE=f32 python3 bench.py --steps 6 --warmup 2 $common --no-exact-f32 --no-roofline > $G/${tag}_f32_bench_line.json 2>/dev/null
ONDA_CONV_MODE=f32 python3 tools/conv_shapes.py > $G/${tag}_f32_conv_shapes.txt 2>&1
ONDA_CONV_MODE=f32 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $G/prof_$tag/sqf -o sq -- python3 tools/one_pass.py > /dev/null 2> $G/prof_$tag/sqf.err
python3 tools/sq_summary.py $(find $G/prof_$tag/sqf -name "*counter_collection.csv" | head -1) 12 > $G/${tag}_f32_sq_counters.txt 2>&1
ONDA_CONV_MODE=f32 python3 tools/power_log.py $G/${tag}_f32_power.csv -- python3 bench.py --steps 12 --warmup 3 $common --no-exact-f32 --no-roofline > /dev/null 2> $G/${tag}_f32_power_summary.txt
rm -rf $G/${tag}_f32_power.csv $G/prof_$tag/sqf
rm -rf $G/prof_$tag/fetch $G/prof_$tag/write $G/prof_$tag/sq $G/prof_$tag/kt $G/prof_$tag/kt1 $G/prof_$tag/tr   # (raw traces: hundreds of MB)
head -c 400 $G/${tag}_bench_line.json; echo; head -12 $G/${tag}_bench_kernel_stats.csv; cat $G/${tag}_sq_counters.txt | head -8
