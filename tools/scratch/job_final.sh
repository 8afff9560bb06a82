mkdir -p gpurun_out/r2n
timeout 900 python bench.py > gpurun_out/r2n/bench.json 2> gpurun_out/r2n/bench.err; tail -2 gpurun_out/r2n/bench.err; cut -c1-1500 gpurun_out/r2n/bench.json
cp gpurun_out/bench_kernel_classes.json gpurun_out/r2n/ 2>/dev/null; cp gpurun_out/bench_gemm_shapes.json gpurun_out/r2n/ 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
