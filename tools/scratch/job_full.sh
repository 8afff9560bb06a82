mkdir -p gpurun_out/full
for i in 1 2; do timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/full/gputest_$i.log 2>&1; tail -1 gpurun_out/full/gputest_$i.log; done
timeout 600 python bench.py > gpurun_out/full/bench.json 2> gpurun_out/full/bench.err
cp gpurun_out/bench_kernel_classes.json gpurun_out/bench_gemm_shapes.json gpurun_out/full/
cut -c1-220 gpurun_out/full/bench.json
