#!/bin/bash
# One GPU: what does the gradient all-reduce's on-GPU footprint (CUs + HBM traffic) cost the step?  bench.py --emulate-allreduce at
# WJ_PERSIST_CUS 32 / 28 / 24, unpaced and paced to an assumed 8-GPU bus bandwidth.  An EMULATION: nothing crosses xGMI.
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$root"
out=$root/gpurun_out/r5/emulate; mkdir -p "$out"
common="--steps 20 --warmup 5 --no-profile --no-cpu-baseline --dense-steps 0"
for cus in 32 28 24; do
  WJ_PERSIST_CUS=$cus python3 bench.py $common > "$out/plain_$cus.json" 2> "$out/plain_$cus.err"
  WJ_PERSIST_CUS=$cus python3 bench.py $common --emulate-allreduce > "$out/emu_unpaced_$cus.json" 2> "$out/emu_unpaced_$cus.err"
  for bw in 150 300; do
    WJ_PERSIST_CUS=$cus WJ_EMULATE_BUSBW_GBPS=$bw python3 bench.py $common --emulate-allreduce > "$out/emu_${bw}_$cus.json" 2> "$out/emu_${bw}_$cus.err"
  done
done
python3 - "$out" <<'PY' | tee "$out/summary.txt"
import json, sys, glob, os
d = sys.argv[1]
print("# bench.py --emulate-allreduce, one MI355X, 256 clips; EMULATION of the all-reduce's CU + HBM footprint (nothing crosses xGMI)")
print(f"{'persist CUs/XCD':>16s} {'mode':>14s} {'ms/step':>9s} {'window ms':>10s} {'exposed ms':>11s}")
for cus in (32, 28, 24):
    for mode in ("plain", "emu_unpaced", "emu_300", "emu_150"):
        try:
            l = json.loads([x for x in open(f"{d}/{mode}_{cus}.json") if x.startswith("{")][-1])
        except Exception as e:
            print(cus, mode, "failed", e); continue
        e = l.get("allreduce_emulated") or {}
        print(f"{cus:16d} {mode:>14s} {l['ms_per_step']:9.2f} {str(e.get('backward_window_ms')):>10s} {str(e.get('exposed_ms')):>11s}")
PY
