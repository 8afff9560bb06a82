#!/bin/bash
# Interleaved two-stream step times of several configurations on ONE box: tools/r6_ab2s.sh <reps> tag[:ENV=VAL,...] ...
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$root"; out=gpurun_out/r6/${AB_NAME:-ab2s}; mkdir -p $out
reps=$1; shift
for r in $(seq 1 $reps); do
  for cfg in "$@"; do
    tag=${cfg%%:*}; envs=""; [ "$cfg" != "$tag" ] && envs=${cfg#*:}
    ( IFS=','; for kv in $envs; do [ -n "$kv" ] && export "$kv"; done; unset IFS
      timeout 200 python3 bench.py --steps 30 --warmup 5 --no-profile --no-cpu-baseline --no-calibration --dense-steps 0 2>/dev/null < /dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('$tag rep $r', d['ms_per_step'])" )
  done
done | tee $out/log.txt
python3 - $out/log.txt <<'PY'
import sys, collections
d = collections.defaultdict(list)
for l in open(sys.argv[1]):
    p = l.split()
    if len(p) == 4: d[p[0]].append(float(p[3]))
for k, v in d.items():
    print(f"{k:12s} mean {sum(v)/len(v):.2f}  min {min(v):.2f}  max {max(v):.2f}  n={len(v)}")
PY
