#!/bin/bash
# is a failure of the full GPU suite order-dependent?  whole test files, repeated; the names and assertion lines of what failed
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out/r6
for i in $(seq 1 ${1:-2}); do
  timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_jepa_gpu.py -m gpu -q -rf --tb=short 2>&1 | grep -E "passed|failed|^FAILED|^E  " | cut -c1-400
  echo " [run $i]"
done
