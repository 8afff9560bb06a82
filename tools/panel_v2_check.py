#!/usr/bin/env python3
"""Laboratory: version 2 of the row-panel loop (WJ_PANEL_V2=1) -- bit-identical to the one-tile eight-phase kernel? -- and its time beside
version 1 and the persistent kernel (operands not cache-resident, interleaved)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("WAVJEPA_HIP_LIB", os.path.join(ROOT, "wavjepa_amd", "lib", "libwavjepa_hip_lab.so"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from wavjepa_amd import ops  # noqa: E402

dev, bf = torch.device("cuda:0"), torch.bfloat16
N = 384
ok = True
for M, K, cus, with_bias in [(33100, 384, 32, True), (40000, 1152, 28, True), (87421, 1536, 32, False), (5003, 256, 32, True), (128, 768, 32, False),
                             (20001, 384, 5, True), (87000, 512, 17, True)]:
    g = torch.Generator(device=dev).manual_seed(M + K)
    A = torch.randn(M, K, device=dev, generator=g).to(bf)
    W = (torch.randn(N, K, device=dev, generator=g) * 0.08).to(bf)
    bias = torch.randn(N, device=dev, generator=g) if with_bias else None

    def run(variant, v2, n_cus=None):
        os.environ["WJ_PANEL_V2"] = "1" if v2 else "0"
        C = torch.full((M, N), float("nan"), dtype=bf, device=dev)
        ops.gemm(A, W, C, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=bias, schedule=variant, persist_cus=n_cus)
        torch.cuda.synchronize()
        return C

    C3 = run(3, False)
    same = True
    for n_cus in (cus, 32):
        for rep in range(3):
            C5 = run(5, True, n_cus)
            same = same and torch.equal(C5.view(torch.int16), C3.view(torch.int16))
    nan = bool(torch.isnan(run(5, True, cus).float()).any())
    ok = ok and same and not nan
    print(f"M={M} K={K} cus={cus}: version 2 bit-identical to variant 3: {same}  NaN left: {nan}", flush=True)
print("ALL OK" if ok else "MISMATCH", flush=True)
if not ok:
    sys.exit(1)
M = 87421
junk = torch.empty(768 * 1024 * 1024 // 4, device=dev)
for K in (384, 1152, 1536):
    A = torch.randn(M, K, device=dev).to(bf)
    W = (torch.randn(N, K, device=dev) * 0.05).to(bf)
    bias = torch.randn(N, device=dev)
    C = torch.empty(M, N, device=dev, dtype=bf)
    res = {"persistent": [], "v1": [], "v2": []}
    for r in range(12):
        for name, (v, v2) in (("persistent", (4, False)), ("v1", (5, False)), ("v2", (5, True))):
            os.environ["WJ_PANEL_V2"] = "1" if v2 else "0"
            junk.fill_(float(r))
            e0, e1 = ops.TimingEvent(), ops.TimingEvent()
            s = torch.cuda.current_stream().cuda_stream
            e0.record(s)
            ops.gemm(A, W, C, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=bias, schedule=v)
            e1.record(s)
            torch.cuda.synchronize()
            if r > 1:
                res[name].append(e0.elapsed_time(e1) * 1e3)
    med = {k: sorted(t)[len(t) // 2] for k, t in res.items()}
    fl = 2.0 * M * N * K
    print(f"M={M} K={K}: " + "   ".join(f"{k} {med[k]:7.1f} us ({fl / med[k] / 1e6:5.0f} TFLOP/s)" for k in med), flush=True)
