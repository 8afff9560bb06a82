#!/bin/bash
# round 6: the mailbox fix of the persistent GEMM (no vmcnt(0) at item boundaries) and the deferred-epilogue kernel (variant 6)
cd ${GRAFT_REPO_ROOT:-.}; out=gpurun_out/r6/pde; mkdir -p $out
timeout 900 python3 tools/gemm_check.py 3 4 4 > $out/check_3_4.log 2>&1; echo "check 3 4 rc=$?"; tail -3 $out/check_3_4.log
timeout 900 python3 tools/gemm_check.py 4 6 4 > $out/check_4_6.log 2>&1; echo "check 4 6 rc=$?"; grep -E "epi=(1|4|5|6)|MISMATCH|mismatching" $out/check_4_6.log | cut -c1-220
AB_NAME=pde/ab tools/r6_ab2s.sh 3 prev:WAVJEPA_HIP_LIB=/root/repo/build_tmp/libwavjepa_hip_prev.so new pde:WAVJEPA_HIP_LIB=/root/repo/wavjepa_amd/lib/libwavjepa_hip_lab.so,WJ_GEMM_PDE=1
