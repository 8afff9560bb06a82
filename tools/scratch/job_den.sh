mkdir -p gpurun_out/r2m
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -E "^[F.]+ |passed|failed|^FAILED|^E  +assert" | head -20 | tee gpurun_out/r2m/gpu_tests.log
timeout 600 python tools/denoiser_bench.py 2>&1 | tail -2 | tee gpurun_out/r2m/denoiser_bench.json
timeout 600 python bench.py --no-cpu-baseline --dense-steps 0 2>/dev/null | cut -c1-200
