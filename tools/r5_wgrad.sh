mkdir -p gpurun_out/r5/g2
timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -k "wgrad" > gpurun_out/r5/g2/pytest.log 2>&1; tail -3 gpurun_out/r5/g2/pytest.log
for m in 0 1 2; do echo "WJ_WGRAD_384=$m"; WJ_WGRAD_384=$m timeout 200 python3 tools/wgrad_bench.py 2>&1 | grep -v amdgpu.ids; done
