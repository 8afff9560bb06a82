#!/bin/bash
# is a failure of the full GPU suite order-dependent?  whole test files, repeated, with and without the round's fused gradient clear
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out/r6
for i in 1 2 3; do timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_jepa_gpu.py -m gpu -q 2>&1 | grep -E "passed|failed|AssertionError: \(" | tr '\n' ' '; echo " [default $i]"; done
for i in 1 2; do WJ_ADAMW_ZERO_GRAD=0 timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_jepa_gpu.py -m gpu -q 2>&1 | grep -E "passed|failed|AssertionError: \(" | tr '\n' ' '; echo " [zero-grad off $i]"; done
