#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$root"
for ns in 0 1; do
  echo "== nostore $ns"
  WJ_PERSIST_DIAG_NOSTORE=$ns WJ_COLD_EPI=0,1,6 python3 tools/gemm_cold.py 4 2>&1 | grep "^M="
done
