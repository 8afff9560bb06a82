timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "gemm" 2>&1 | tail -3
for v in 0 1; do echo "== WJ_GEMM_TAIL_SPLIT=$v"; WJ_GEMM_TAIL_SPLIT=$v timeout 300 python tools/blas_reference.py 2>&1 | grep -E "teacher|stud"; done
for v in 0 1; do WJ_GEMM_TAIL_SPLIT=$v timeout 600 python bench.py --no-cpu-baseline --dense-steps 0 2>/dev/null | cut -c1-160; done
