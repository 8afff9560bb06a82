mkdir -p gpurun_out/r2h
timeout 900 python -m pytest tests/test_scene_gpu.py -q -x 2>&1 | tail -12
timeout 600 python tools/scene_bench.py --cpu 2>&1 | tail -1 | tee gpurun_out/r2h/scene_bench.json
R=$PWD; cd /tmp && export TMPDIR=/tmp && cd $R
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r2h/prof -o run --output-format csv -- python3 tools/scene_bench.py > /dev/null 2>&1
find gpurun_out/r2h -name "*kernel_trace.csv" -delete; find gpurun_out/r2h -name "*agent_info.csv" -delete
head -4 gpurun_out/r2h/prof/run_kernel_stats.csv | cut -c1-60,150-260
