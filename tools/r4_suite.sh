#!/bin/bash
# round 4: the GPU suite, then serial-sum A/B of the configurations given as arguments (see tools/prof_step.sh)
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$root"; mkdir -p gpurun_out/r4
tag=${SUITE_TAG:-suite}
timeout 1500 python3 -m pytest tests -m gpu -x -q > gpurun_out/r4/$tag.log 2>&1; echo "pytest -m gpu: rc=$?"; tail -5 gpurun_out/r4/$tag.log
[ $# -gt 0 ] && bash tools/prof_step.sh "$@"
