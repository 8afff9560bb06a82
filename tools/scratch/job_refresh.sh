set -x
mkdir -p gpurun_out/r2k
R=$PWD
cd /tmp && export TMPDIR=/tmp && cd $R
timeout 1200 python -m pytest tests -q -m gpu -x > gpurun_out/r2k/gputest_run4.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2k/gputest_run4.log
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/r2k/prof2s -o run --output-format csv -- python3 bench.py --steps 10 --warmup 5 --no-cpu-baseline --dense-steps 0 > gpurun_out/r2k/prof2s.log 2>&1
export WJ_SIDE_STREAM=0
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/r2k/prof1s -o run --output-format csv -- python3 bench.py --steps 10 --warmup 5 --no-cpu-baseline --dense-steps 0 > gpurun_out/r2k/prof1s.log 2>&1
unset WJ_SIDE_STREAM
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/r2k/pmc_fetch -o runc --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --dense-steps 0 > gpurun_out/r2k/pmcf.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/r2k/pmc_write -o runc --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --dense-steps 0 > gpurun_out/r2k/pmcw.log 2>&1
python tools/pmc_traffic.py gpurun_out/r2k/pmc_fetch gpurun_out/r2k/pmc_write gpurun_out/r2k/pmc_traffic.json > gpurun_out/r2k/pmc_traffic.txt
find gpurun_out/r2k -name "*kernel_trace.csv" -delete
find gpurun_out/r2k -name "*counter_collection.csv" -delete
find gpurun_out/r2k -name "*agent_info.csv" -delete
tail -3 gpurun_out/r2k/gputest_run4.log
