#!/bin/bash
# round 4, GEMM screens on the GPU box: the persistent kernel's new paths against the one-tile schedules, then cold / warm timings
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$root"; out=gpurun_out/r4/gemm; mkdir -p $out
timeout 900 python3 tools/gemm_check.py 3 4 3 > $out/check_3_4.log 2>&1; echo "check 3 vs 4: rc=$?"
WJ_PERSIST_HALF=0 timeout 600 python3 tools/gemm_check.py 3 4 2 > $out/check_3_4_nohalf.log 2>&1; echo "check (no half items): rc=$?"
WJ_PERSIST_MIN_TILES=1 WJ_CHECK_ONLY=0 timeout 600 python3 tools/gemm_check.py 3 4 2 > $out/check_3_4_min1.log 2>&1; echo "check (min tiles 1): rc=$?"
timeout 300 python3 tools/gemm_repeat.py > $out/repeat.log 2>&1; echo "repeat: rc=$?"
grep -h "MISMATCH\|COLSUM\|mismatching\|Error\|error" $out/*.log | head -20
grep -h "^M=" $out/check_3_4.log
