#!/bin/bash
# Usage (on the GPU box, from the repo root):  tools/prof_step.sh <tag>[:ENV=VAL[,ENV=VAL...]] [<tag2>[:...]] ...
# For every configuration: the serial kernel-time sum of the training step (the round-4 accept criterion: bench.py with
# WJ_SIDE_STREAM=0 under rocprofv3 --kernel-trace --stats), then the single-stream and the two-stream step times of the same build.
# Configurations of ONE call run on ONE box back to back: compare only those (boxes differ by ~4 % in sustained clock).
# Outputs under gpurun_out/${WJ_ROUND:-r5}/<tag>/: run_kernel_stats.csv, serial_sum.txt, step_1s.json, step_2s.json
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$root"
common="--no-cpu-baseline --dense-steps 0"
first=""
for cfg in "$@"; do
  tag=${cfg%%:*}
  envs=""
  [ "$cfg" != "$tag" ] && envs=${cfg#*:}
  out=$root/gpurun_out/${WJ_ROUND:-r5}/$tag
  mkdir -p "$out"
  (
    IFS=','; for kv in $envs; do [ -n "$kv" ] && export "$kv"; done; unset IFS
    WJ_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof" -o run -- python3 bench.py --steps 10 --warmup 5 --no-profile $common > "$out/prof.json" 2> "$out/prof.err"
    f=$(find "$out/prof" -name "run_kernel_stats.csv" | head -1)
    [ -n "$f" ] && cp "$f" "$out/run_kernel_stats.csv" && python3 tools/serial_sum.py "$out/run_kernel_stats.csv" > "$out/serial_sum.txt"
    rm -rf "$out/prof"                      # the per-launch trace is not needed; keep the merged-back payload small
    WJ_SIDE_STREAM=0 python3 bench.py --steps 20 --warmup 5 --no-profile $common > "$out/step_1s.json" 2> "$out/step_1s.err"
    python3 bench.py --steps 20 --warmup 5 $common > "$out/step_2s.json" 2> "$out/step_2s.err"
    cp gpurun_out/bench_gemm_shapes.json "$out/gemm_shapes.json" 2>/dev/null
    cp gpurun_out/bench_kernel_classes.json "$out/kernel_classes.json" 2>/dev/null
  )
  echo "== $tag ($envs)"
  head -1 "$out/serial_sum.txt"
  python3 - "$out" <<'PY'
import json, sys
for n in ("step_1s", "step_2s"):
    try:
        d = json.loads([l for l in open(f"{sys.argv[1]}/{n}.json") if l.startswith("{")][-1])
        print(" ", n, d["ms_per_step"], "ms/step", d["value"], "clips/s", "loss", d["final_loss"], "frac", (d.get("roofline") or {}).get("frac"))
    except Exception as e:
        print(" ", n, "failed", e)
PY
  if [ -z "$first" ]; then first=$out; else python3 tools/serial_sum.py "$first/run_kernel_stats.csv" "$out/run_kernel_stats.csv" | sed -n 2,28p; fi
done
