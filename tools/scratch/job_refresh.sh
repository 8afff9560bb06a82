set -x
mkdir -p gpurun_out/r2f
R=$PWD
cd /tmp && export TMPDIR=/tmp && cd $R
timeout 1200 python -m pytest tests -q -m gpu -x > gpurun_out/r2f/gputest_run3.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2f/gputest_run3.log
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/r2f/prof2s -o run --output-format csv -- python3 bench.py --steps 10 --warmup 5 --no-cpu-baseline --dense-steps 0 > gpurun_out/r2f/prof2s.log 2>&1
export WJ_SIDE_STREAM=0
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/r2f/prof1s -o run --output-format csv -- python3 bench.py --steps 10 --warmup 5 --no-cpu-baseline --dense-steps 0 > gpurun_out/r2f/prof1s.log 2>&1
unset WJ_SIDE_STREAM
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/r2f/pmc_fetch -o runc --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --dense-steps 0 > gpurun_out/r2f/pmcf.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/r2f/pmc_write -o runc --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --dense-steps 0 > gpurun_out/r2f/pmcw.log 2>&1
python tools/pmc_traffic.py gpurun_out/r2f/pmc_fetch gpurun_out/r2f/pmc_write gpurun_out/r2f/pmc_traffic.json
# keep only the small summaries
find gpurun_out/r2f -name "*kernel_trace.csv" -delete
find gpurun_out/r2f -name "*counter_collection.csv" -delete
find gpurun_out/r2f -name "*agent_info.csv" -delete
ls -R gpurun_out/r2f | head -50
tail -3 gpurun_out/r2f/gputest_run3.log
