#!/bin/bash
# one GPU call: row-panel kernel correctness + micro-benchmark, then the interleaved step A/Bs of the round's engine changes, then SQ counters
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$root"; mkdir -p gpurun_out/r6
LAB=$root/wavjepa_amd/lib/libwavjepa_hip_lab.so
timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "row_panel" 2>&1 | tail -12 > gpurun_out/r6/panel_tests.log
cat gpurun_out/r6/panel_tests.log
timeout 200 python tools/panel_bench.py > gpurun_out/r6/panel_bench.log 2>&1; cat gpurun_out/r6/panel_bench.log
OFF="WJ_ADAMW_ZERO_GRAD=0,WJ_SUMSQ_SECTIONS=0,WJ_GEMM_PANEL=0,WAVJEPA_HIP_LIB=$LAB"
if grep -q " passed" gpurun_out/r6/panel_tests.log && ! grep -q "failed" gpurun_out/r6/panel_tests.log; then
  AB_NAME=ab_call4 tools/r6_ab2s.sh 3 base:$OFF zg:WJ_SUMSQ_SECTIONS=0,WJ_GEMM_PANEL=0,WAVJEPA_HIP_LIB=$LAB sumsq:WJ_ADAMW_ZERO_GRAD=0,WJ_GEMM_PANEL=0,WAVJEPA_HIP_LIB=$LAB \
     split:$OFF,WJ_CONV_SPLIT=1 panel:WJ_ADAMW_ZERO_GRAD=0,WJ_SUMSQ_SECTIONS=0,WAVJEPA_HIP_LIB=$LAB all:WJ_CONV_SPLIT=1,WAVJEPA_HIP_LIB=$LAB
  timeout 600 python -m pytest tests/test_jepa_gpu.py -m gpu -x -q -k "test_forward_backward_parity or trajectory or overlapped" 2>&1 | tail -6 > gpurun_out/r6/e2e_call4.log; cat gpurun_out/r6/e2e_call4.log
else
  AB_NAME=ab_call4 tools/r6_ab2s.sh 3 base:$OFF zg:WJ_SUMSQ_SECTIONS=0,WJ_GEMM_PANEL=0,WAVJEPA_HIP_LIB=$LAB sumsq:WJ_ADAMW_ZERO_GRAD=0,WJ_GEMM_PANEL=0,WAVJEPA_HIP_LIB=$LAB split:$OFF,WJ_CONV_SPLIT=1
fi
timeout 900 tools/r6_sq.sh
