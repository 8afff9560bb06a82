mkdir -p gpurun_out/r2j
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | grep -E "^[F.]+ |passed|failed|^FAILED|^E  +assert" | head -20 | tee gpurun_out/r2j/gpu_tests.log
for k in 1 2; do timeout 600 python bench.py --no-cpu-baseline --dense-steps 0 2>/dev/null | cut -c1-200; done | tee gpurun_out/r2j/bench.log
