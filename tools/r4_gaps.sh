#!/bin/bash
# GPU idle gaps of the two-stream step (rocprofv3 kernel trace + tools/trace_gaps.py), then repeated step timings
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$root"; export TMPDIR=/tmp
out=gpurun_out/r4/gaps; mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out/prof -o run -- python3 bench.py --steps 10 --warmup 5 --no-profile --no-cpu-baseline --dense-steps 0 > $out/bench.json 2> $out/bench.err
python3 tools/trace_gaps.py $(find $out/prof -name "run_kernel_trace.csv" | head -1) > $out/trace_gaps.txt 2>&1; rm -rf $out/prof
tail -1 $out/bench.json | cut -c1-140; sed -n 15,30p $out/trace_gaps.txt
for rep in 1 2 3; do
  r=$(python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --dense-steps 0 --no-profile 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'])")
  r1=$(WJ_SIDE_STREAM=0 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --dense-steps 0 --no-profile 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'])")
  echo "rep $rep: two-stream $r ms, one-stream $r1 ms"
done
