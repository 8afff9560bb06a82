#!/bin/bash
# round 6: the evidence set of the round in one GPU-box call (outputs under gpurun_out/r6/final; what is to be judged is copied into profiles/)
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$root"; out=gpurun_out/r6/final; mkdir -p $out; export TMPDIR=/tmp
common="--no-cpu-baseline --dense-steps 0"
# 1. rocprof summaries: serialised and two-stream
WJ_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_serial -o run -- python3 bench.py --steps 10 --warmup 5 --no-profile --no-calibration $common > $out/prof_serial.json 2> $out/prof_serial.err
cp $(find $out/prof_serial -name "run_kernel_stats.csv" | head -1) $out/bench_kernel_stats_serial.csv; rm -rf $out/prof_serial
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_2s -o run -- python3 bench.py --steps 10 --warmup 5 --no-profile --no-calibration $common > $out/prof_2s.json 2> $out/prof_2s.err
cp $(find $out/prof_2s -name "run_kernel_stats.csv" | head -1) $out/bench_kernel_stats.csv
python3 tools/trace_gaps.py $(find $out/prof_2s -name "run_kernel_trace.csv" | head -1) > $out/trace_gaps.txt 2>&1; rm -rf $out/prof_2s
python3 tools/serial_sum.py $out/bench_kernel_stats_serial.csv > $out/serial_sum.txt
# 2. PMC traffic: two separate passes (kernel-trace only)
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/pmc_fetch -o runc --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-profile --no-calibration $common > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/pmc_write -o runc --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-profile --no-calibration $common > $out/pmc_write.log 2>&1
python3 tools/pmc_traffic.py $out/pmc_fetch $out/pmc_write $out/pmc_traffic.json > $out/pmc_traffic.txt 2>&1; rm -rf $out/pmc_fetch $out/pmc_write
# 3. the default bench line (with calibration and the CPU baseline), three more without the baseline, and one stream only
python3 bench.py > $out/bench.json 2> $out/bench.err
cp gpurun_out/bench_gemm_shapes.json $out/bench_gemm_shapes.json; cp gpurun_out/bench_kernel_classes.json $out/bench_kernel_classes.json
for i in 1 2 3; do python3 bench.py --no-cpu-baseline --no-profile --dense-steps 0 2>/dev/null | tail -1; done > $out/bench_same_box_x3.json
WJ_SIDE_STREAM=0 python3 bench.py $common --no-profile > $out/bench_one_stream.json 2> $out/bench_one_stream.err
# 4. GEMM screens (laboratory library)
timeout 600 python3 tools/gemm_check.py 3 4 4 > $out/gemm_check_3_4.log 2>&1
timeout 200 python3 tools/panel_bench.py > $out/panel_bench.log 2>&1
# 5. the GPU suite, plain and with the arena NaN-poisoned
timeout 1500 python3 -m pytest tests -m gpu -q > $out/gputest.log 2>&1; tail -3 $out/gputest.log
WJ_ARENA_FILL=nan timeout 1500 python3 -m pytest tests -m gpu -q > $out/gputest_nan_arena.log 2>&1; tail -3 $out/gputest_nan_arena.log
# 6. the other workloads and the trainer
for w in 4s-bf16 4s-fp8 2s-nat; do python3 bench.py --workload $w --steps 10 --no-cpu-baseline --no-profile --dense-steps 0 2>/dev/null | tail -1; done > $out/other_workloads.json
timeout 600 python3 train.py trainer.steps=3000 trainer.warmup_steps=500 trainer.log_every_n_steps=250 save_dir=/tmp/r6_runs > $out/train_3000steps.log 2>&1
head -2 $out/serial_sum.txt; tail -1 $out/bench.json | cut -c1-300; cat $out/bench_same_box_x3.json | cut -c1-120; tail -1 $out/bench_one_stream.json | cut -c1-200
