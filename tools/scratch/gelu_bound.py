"""Upper bound on what dropping / shrinking the saved gelu'(h) could buy: the same shapes with and without the second tensor."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from wavjepa_amd import ops
dev = torch.device("cuda:0"); bf = torch.bfloat16

def timeit(fn, n=7, reps=3):
    ts = []
    for r in range(n + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        if r: ts.append(e0.elapsed_time(e1) / reps)
    return sorted(ts)[len(ts) // 2] * 1e3

for (M, D, F) in ((87066, 384, 1536), (10152, 768, 3072)):
    X = torch.randn(M, D, device=dev).to(bf); W1 = (torch.randn(F, D, device=dev) * 0.05).to(bf); b1 = torch.randn(F, device=dev)
    H1 = torch.empty(M, F, device=dev, dtype=bf); H2 = torch.empty(M, F, device=dev, dtype=bf)
    kw = dict(M=M, N=F, K=D, lda=D, ldb=D, ldc=F, bias=b1)
    t2 = timeit(lambda: ops.gemm(X, W1, H1, C2=H2, epilogue=ops.EPI_BIAS_GELU2, **kw))
    t1 = timeit(lambda: ops.gemm(X, W1, H1, epilogue=ops.EPI_BIAS_GELU, **kw))
    t0 = timeit(lambda: ops.gemm(X, W1, H1, epilogue=ops.EPI_BF16, **kw))
    print(f"linear1 M={M} N={F} K={D}: GELU2 (two outputs) {t2:.1f} us | GELU (one output) {t1:.1f} us | plain bf16 {t0:.1f} us")
    # dgrad of linear2: dH = (dY . W2) * gelu'(h):  NT, N=F, K=D
    dY = torch.randn(M, D, device=dev).to(bf); W2 = (torch.randn(D, F, device=dev) * 0.05).to(bf)
    dH = torch.empty(M, F, device=dev, dtype=bf)
    kw = dict(M=M, N=F, K=D, lda=D, ldb=F, ldc=F, b_trans=1)
    tg = timeit(lambda: ops.gemm(dY, W2, dH, aux=H1, epilogue=ops.EPI_MUL_GELU_GRAD, **kw))
    tp = timeit(lambda: ops.gemm(dY, W2, dH, epilogue=ops.EPI_BF16, **kw))
    print(f"linear2 dgrad M={M} N={F} K={D}: MUL_GELU_GRAD {tg:.1f} us | plain bf16 {tp:.1f} us")
