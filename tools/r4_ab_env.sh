#!/bin/bash
# unprofiled A/B of one environment switch: tools/r4_ab_env.sh VAR [reps]  -> two-stream and one-stream step times for VAR=1 / VAR=0, interleaved
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$root"; var=$1; reps=${2:-3}
for rep in $(seq 1 $reps); do for v in 1 0; do
  r=$(env $var=$v python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --dense-steps 0 --no-profile 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'])")
  r1=$(env WJ_SIDE_STREAM=0 $var=$v python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --dense-steps 0 --no-profile 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'])")
  echo "rep $rep $var=$v: two-stream $r ms, one-stream $r1 ms"
done; done
