#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$root"
for us in 0 8 16 24 32; do
  echo "== stagger $us us"
  WJ_PERSIST_STAGGER_US=$us WJ_COLD_EPI=1,2,6 python3 tools/gemm_cold.py 4 2>&1 | grep "^M="
done
