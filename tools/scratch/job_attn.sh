mkdir -p gpurun_out/r2g
for v in 0 3; do echo "== WJ_ATTN_BWD_FRAG=$v"; WJ_ATTN_BWD_FRAG=$v timeout 300 python tools/attn_bench.py 2>&1 | grep attn; done | tee gpurun_out/r2g/attn_ab3.log
for k in 1 2; do timeout 1200 python -m pytest tests -q -m gpu 2>&1 | grep -E "^[F.]+ |passed|failed|^FAILED|^E  +assert" | head -20; done | tee gpurun_out/r2g/gpu_tests.log
timeout 600 python bench.py --no-cpu-baseline --dense-steps 0 > gpurun_out/r2g/bench.json 2> gpurun_out/r2g/bench.err; cat gpurun_out/r2g/bench.json | cut -c1-400
