#!/bin/bash
# kernel trace + memory-copy trace of a short two-stream bench run: which launches surround the ~50 copyBuffer launches per step?
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$root"; export TMPDIR=/tmp
out=$root/gpurun_out/r5/trace; mkdir -p "$out"
WJ_SIDE_STREAM=0 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$out/prof" -o run -- python3 bench.py --steps 4 --warmup 3 --no-profile --no-cpu-baseline --dense-steps 0 > "$out/bench.json" 2> "$out/bench.err"
k=$(find "$out/prof" -name "run_kernel_trace.csv" | head -1); m=$(find "$out/prof" -name "run_memory_copy_trace.csv" | head -1)
python3 - "$k" "$m" > "$out/copies.txt" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"][:70] for r in rows]
ctx = collections.Counter()
for i, n in enumerate(names):
    if "copyBuffer" in n:
        ctx[(names[i - 1] if i else "-", names[i + 1] if i + 1 < len(names) else "-")] += 1
print("copyBuffer launches:", sum(ctx.values()))
for (a, b), c in ctx.most_common(40):
    print(f"{c:5d}  after {a:72s} before {b}")
try:
    mc = list(csv.DictReader(open(sys.argv[2])))
    print("memory copies:", len(mc))
    c2 = collections.Counter((r.get("Direction", "?"), r.get("Bytes", r.get("Size", "?"))) for r in mc)
    for k, v in c2.most_common(30):
        print(v, k)
except Exception as e:
    print("no memory copy trace:", e)
PY
rm -rf "$out/prof"
cat "$out/copies.txt"
