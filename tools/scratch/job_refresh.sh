set -x
mkdir -p gpurun_out/r2v
R=$PWD
cd /tmp && export TMPDIR=/tmp && cd $R
timeout 900 python bench.py > gpurun_out/r2v/bench.json 2> gpurun_out/r2v/bench.err
cp gpurun_out/bench_kernel_classes.json gpurun_out/bench_gemm_shapes.json gpurun_out/r2v/
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/r2v/prof2s -o run --output-format csv -- python3 bench.py --steps 10 --warmup 5 --no-cpu-baseline --dense-steps 0 > gpurun_out/r2v/prof2s.log 2>&1
export WJ_SIDE_STREAM=0
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/r2v/prof1s -o run --output-format csv -- python3 bench.py --steps 10 --warmup 5 --no-cpu-baseline --dense-steps 0 > gpurun_out/r2v/prof1s.log 2>&1
unset WJ_SIDE_STREAM
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/r2v/pmc_fetch -o runc --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --dense-steps 0 > gpurun_out/r2v/pmcf.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/r2v/pmc_write -o runc --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --dense-steps 0 > gpurun_out/r2v/pmcw.log 2>&1
python tools/pmc_traffic.py gpurun_out/r2v/pmc_fetch gpurun_out/r2v/pmc_write gpurun_out/r2v/pmc_traffic.json > gpurun_out/r2v/pmc_traffic.txt
find gpurun_out/r2v -name "*kernel_trace.csv" -delete; find gpurun_out/r2v -name "*counter_collection.csv" -delete; find gpurun_out/r2v -name "*agent_info.csv" -delete
for w in 4s-fp8 4s-bf16 2s-nat; do timeout 600 python bench.py --workload $w --no-cpu-baseline --dense-steps 0 > gpurun_out/r2v/bench_$w.json 2>/dev/null; done
timeout 300 python tools/blas_reference.py 2>&1 | grep -v amdgpu > gpurun_out/r2v/blas_reference.log
cut -c1-220 gpurun_out/r2v/bench.json
