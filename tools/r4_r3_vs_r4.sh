#!/bin/bash
# the round-3 behaviour of this build (every round-4 switch off) against its default, interleaved, unprofiled, one box
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$root"
OFF="WJ_WT_DGRAD=0 WJ_PERSIST_HALF=0 WJ_DEFER_FOLDS=0 WJ_PINNED_UPLOAD=0"
run() { env "$@" python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --dense-steps 0 --no-profile 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'])"; }
for rep in 1 2 3 4; do
  a=$(run $OFF); b=$(run WJ_NOOP=1); a1=$(run WJ_SIDE_STREAM=0 $OFF); b1=$(run WJ_SIDE_STREAM=0)
  echo "rep $rep: round-3 behaviour $a ms (one stream $a1) | round 4 $b ms (one stream $b1)"
done
