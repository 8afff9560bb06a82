#!/bin/bash
# LayerNorm launches in isolation under the backward's grid policies (one process per configuration, same box)
out=gpurun_out/r4/ln; mkdir -p $out
for cfg in "WJ_LN_BWD_ONE_PASS_ROWS=0" "WJ_LN_BWD_ONE_PASS_ROWS=16384" "WJ_LN_BWD_ONE_PASS_ROWS=65536"; do
  echo "== $cfg"
  env $cfg timeout 300 python tools/ln_bench.py 2>&1 | grep -v "^$" | tee -a $out/ln_$cfg.log
done
