#!/bin/bash
# two-stream picture of one step of the current build (tools/trace_streams.py on a rocprofv3 kernel trace)
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$root"; export TMPDIR=/tmp; out=gpurun_out/r5/streams; mkdir -p $out
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/prof -o run -- python3 bench.py --steps 6 --warmup 4 --no-profile --no-cpu-baseline --dense-steps 0 > $out/bench.json 2> $out/err.txt < /dev/null
f=$(find $out/prof -name "run_kernel_trace.csv" | head -1)
[ -n "$f" ] && python3 tools/trace_streams.py "$f" > $out/trace_streams.txt 2>&1
[ -n "$f" ] && python3 tools/r5_window.py "$f" ${WIN_FROM:-3.0} ${WIN_TO:-9.0} > $out/window.txt 2>&1
rm -rf $out/prof
head -3 $out/trace_streams.txt
