#!/bin/bash
# round 5: the evidence set of the round in one GPU-box call (outputs under gpurun_out/r5/final; copy what is to be judged into profiles/)
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$root"; out=gpurun_out/r5/final; mkdir -p $out; export TMPDIR=/tmp
common="--no-cpu-baseline --dense-steps 0"
# 1. rocprof summaries: serialised and two-stream
WJ_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_serial -o run -- python3 bench.py --steps 10 --warmup 5 --no-profile $common > $out/prof_serial.json 2> $out/prof_serial.err
cp $(find $out/prof_serial -name "run_kernel_stats.csv" | head -1) $out/bench_kernel_stats_serial.csv; rm -rf $out/prof_serial
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_2s -o run -- python3 bench.py --steps 10 --warmup 5 --no-profile $common > $out/prof_2s.json 2> $out/prof_2s.err
cp $(find $out/prof_2s -name "run_kernel_stats.csv" | head -1) $out/bench_kernel_stats.csv
python3 tools/trace_gaps.py $(find $out/prof_2s -name "run_kernel_trace.csv" | head -1) > $out/trace_gaps.txt 2>&1; rm -rf $out/prof_2s
python3 tools/serial_sum.py $out/bench_kernel_stats_serial.csv > $out/serial_sum.txt
# 2. PMC traffic: two separate passes (kernel-trace only)
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/pmc_fetch -o runc --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-profile $common > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/pmc_write -o runc --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-profile $common > $out/pmc_write.log 2>&1
python3 tools/pmc_traffic.py $out/pmc_fetch $out/pmc_write $out/pmc_traffic.json > $out/pmc_traffic.txt 2>&1; rm -rf $out/pmc_fetch $out/pmc_write
# ... and the fetch side without the W blocking (what the blocking changes for the N = 3072 launches)
WJ_PERSIST_WBLOCK=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/pmc_fetch0 -o runc --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-profile $common > $out/pmc_fetch0.log 2>&1
python3 tools/pmc_traffic.py $out/pmc_fetch0 $out/pmc_fetch0 $out/pmc_traffic_wblock_off.json > /dev/null 2>&1; rm -rf $out/pmc_fetch0
cp $out/pmc_traffic.json profiles/r05_pmc_traffic.json      # what the bench line's `traffic` reads
# 3. the default bench line (with the CPU baseline), and one stream only
python3 bench.py > $out/bench.json 2> $out/bench.err
cp gpurun_out/bench_gemm_shapes.json $out/bench_gemm_shapes.json; cp gpurun_out/bench_kernel_classes.json $out/bench_kernel_classes.json
WJ_SIDE_STREAM=0 python3 bench.py $common --no-profile > $out/bench_one_stream.json 2> $out/bench_one_stream.err
# 4. GEMM screens
timeout 900 python3 tools/gemm_check.py 3 4 4 > $out/gemm_check_3_4.log 2>&1
timeout 300 python3 tools/gemm_repeat.py > $out/gemm_repeat.log 2>&1
WJ_PAIR_MIN_K=256 timeout 300 python3 tools/gemm_small.py 9945 > $out/gemm_small.log 2>&1
# 5. the GPU suite
timeout 1500 python3 -m pytest tests -m gpu -q > $out/gputest.log 2>&1; tail -3 $out/gputest.log
head -2 $out/serial_sum.txt; tail -1 $out/bench.json | cut -c1-300; tail -1 $out/bench_one_stream.json | cut -c1-200
