#!/bin/bash
# the GPU suite at HEAD, plain and with a NaN-poisoned arena; then the checkpoint/resume test repeated (its masks come from OS entropy)
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out/r6
timeout 1500 python -m pytest tests -m gpu -q -rf --tb=short 2>&1 | tail -25 > gpurun_out/r6/gputest_final.log; tail -8 gpurun_out/r6/gputest_final.log
WJ_ARENA_FILL=nan timeout 1500 python -m pytest tests -m gpu -q -rf --tb=short 2>&1 | tail -25 > gpurun_out/r6/gputest_nan_arena.log; tail -8 gpurun_out/r6/gputest_nan_arena.log
for i in 1 2 3 4 5 6; do timeout 300 python -m pytest tests/test_jepa_gpu.py -m gpu -q -k checkpoint_resume 2>&1 | grep -E "passed|failed|^E  " | cut -c1-300; done 2>&1 | tee gpurun_out/r6/resume_x6.log
