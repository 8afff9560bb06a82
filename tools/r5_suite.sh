#!/bin/bash
# full GPU test suite -> gpurun_out/r5/<tag>/pytest.log
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$root"
tag=${1:-suite}; out=$root/gpurun_out/r5/$tag; mkdir -p "$out"
timeout 1500 python -m pytest tests -m gpu -x -q > "$out/pytest.log" 2>&1; echo "pytest rc $?" >> "$out/pytest.log"
tail -15 "$out/pytest.log"
