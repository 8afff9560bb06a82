import os, sys, torch, numpy as np
sys.path.insert(0, '/root/repo')
from wavjepa_amd import ops
dev = torch.device("cuda:0"); bf = torch.bfloat16
for (M, N, K) in [(51200, 2304, 768), (51200, 768, 3072), (86317, 1536, 384)]:
    A = torch.randn(M, K, device=dev).to(bf); W = (torch.randn(N, K, device=dev) * 0.05).to(bf)
    C = torch.empty(M, N, device=dev, dtype=bf); bias = torch.randn(N, device=dev)
    nwg = ((M + 255) // 256) * ((N + 255) // 256)
    stamps = torch.zeros(nwg, 8, dtype=torch.int64, device=dev)
    ops.gemm_set_variant(3)
    for _ in range(3):
        ops.gemm(A, W, C, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=bias, aux=stamps)
    torch.cuda.synchronize()
    s = stamps.cpu().numpy()
    loop = (s[:, 1] - s[:, 0]); epi = (s[:, 2] - s[:, 1]); real = (s[:, 4] - s[:, 3]) / 100.0   # us (100 MHz)
    start = (s[:, 3] - s[:, 3].min()) / 100.0
    clk = (s[:, 2] - s[:, 0]) / np.maximum(real, 1e-9) / 1e3
    print(f"M={M} N={N} K={K}: WGs {nwg}; cycles prologue+loop median {np.median(loop):.0f}, epilogue median {np.median(epi):.0f}; "
          f"WG lifetime median {np.median(real):.2f} us (p10 {np.percentile(real,10):.2f}, p90 {np.percentile(real,90):.2f}); clock ~{np.median(clk):.2f} GHz; "
          f"kernel span {(s[:,4].max()-s[:,3].min())/100.0:.1f} us")
    order = np.argsort(start)
    # when do WGs start: histogram of start times in rounds
    print("   start-time deciles (us):", np.round(np.percentile(start, [0,10,20,30,40,50,60,70,80,90,100]),1))
