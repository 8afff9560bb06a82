#!/bin/bash
# SQ counters per kernel of one serialised step (rocprofv3 --pmc, --kernel-trace only; separate passes): what the attention, conv0-backward
# and LayerNorm waves wait for.  tools/r6_sq.sh  ->  gpurun_out/r6/sq/{passA,passB}.txt, counters.txt
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$root"; export TMPDIR=/tmp
out=$root/gpurun_out/r6/sq; mkdir -p $out
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u > $out/counters.txt
common="--steps 2 --warmup 1 --no-profile --no-cpu-baseline --no-calibration --dense-steps 0"
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_WAVES"
B="SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM"
for p in A B; do
  eval "ctrs=\$$p"
  WJ_SIDE_STREAM=0 rocprofv3 --pmc $ctrs --kernel-trace -d $out/pass$p -o run --output-format csv -- python3 bench.py $common > $out/pass$p.log 2>&1
  python3 tools/pmc_sq.py $out/pass$p k=attn_ k=conv0_bwd_rows k=conv0_apply k=ln_fwd k=ln_bwd k=gemm_persist_kernel\<1 k=gemm_persist_kernel\<2 > $out/pass$p.txt 2>&1
  rm -rf $out/pass$p
done
tail -5 $out/passA.log $out/passB.log
