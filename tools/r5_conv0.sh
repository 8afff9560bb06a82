#!/bin/bash
# conv0 forward: round-4 launch shapes against round 5's (statistics chunks sized to one round; apply pass at 4 waves per SIMD)
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$root"; out=gpurun_out/r5/conv0; mkdir -p $out
for cfg in "1024 1" "0 1" "1024 2" "0 2"; do
  set -- $cfg
  echo "WJ_CONV0_STATS_TCS=$1 WJ_CONV0_APPLY_OCC=$2"
  WJ_CONV0_STATS_TCS=$1 WJ_CONV0_APPLY_OCC=$2 WJ_CONV0_DUMP=$out/dump_$1_$2.pt python3 tools/conv0_bench.py $( [ -f $out/dump_1024_1.pt ] && echo $out/dump_1024_1.pt ) 2>&1 | grep -v amdgpu.ids
done
rm -f $out/dump_*.pt
timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -k "conv0" 2>&1 | tail -2
