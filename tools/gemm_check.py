#!/usr/bin/env python3
"""Race screen + A/B for a GEMM schedule variant (run on the GPU box): every listed shape is computed with the reference
variant and with the variant under test, many times, on fresh random operands.  Variants 0-3 must agree BIT for bit (they add the k
index in the same order and the bias last); variant 4 starts its accumulators FROM the bias (fp32 (b + sum) instead of (sum + b)), so
against it a rare last-place difference of the bf16 output is allowed: <= 2 % of the elements, each within 2 bf16 ulps (+ 1e-4
absolute where sum and bias cancel; + 4e-3 for the GELU outputs), and its own repetitions must be bit-identical to each other.  A race shows as a large difference
in a few tiles.  Then interleaved timing rounds of both variants in one process."""
import os
os.environ.setdefault("WAVJEPA_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "wavjepa_amd", "lib", "libwavjepa_hip_lab.so"))  # laboratory build: honours the WJ_* A/B switches, exports the stamp reader
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wavjepa_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
bf = torch.bfloat16
SHAPES = [  # M, N, K, epilogue
    (51200, 2304, 768, ops.EPI_BF16), (51200, 768, 768, ops.EPI_BF16), (51200, 768, 3072, ops.EPI_BF16),
    (51200, 3072, 768, ops.EPI_BIAS_GELU), (9907, 2304, 768, ops.EPI_BF16), (9907, 768, 3072, ops.EPI_BF16), (9907, 768, 768, ops.EPI_BF16),
    (9907, 3072, 768, ops.EPI_BIAS_GELU2), (86317, 1536, 384, ops.EPI_BIAS_GELU2), (86317, 1152, 384, ops.EPI_BF16),
    (86317, 384, 1536, ops.EPI_BF16), (86317, 384, 384, ops.EPI_BF16), (8192, 8192, 8192, ops.EPI_BF16),
    (300, 256, 128, ops.EPI_BF16), (257, 512, 256, ops.EPI_BF16), (1000, 264, 384, ops.EPI_BF16),
    # >= 256 output tiles with ragged edges (the persistent variant's pull / edge paths), short K, conv epilogue
    (65537, 264, 128, ops.EPI_BF16), (33000, 520, 256, ops.EPI_BIAS_GELU2), (70001, 384, 1536, ops.EPI_BF16),
    (102912, 512, 1536, ops.EPI_CONV_GELU), (51456, 512, 1024, ops.EPI_CONV_GELU),
    # N % 256 == 128 (last tile column = a half item), N % 256 in (128, 256) (shifted edge tile), a split tail with GELU2 outputs
    (40000, 640, 256, ops.EPI_BIAS_GELU2), (40000, 448, 384, ops.EPI_BF16), (16640, 1024, 256, ops.EPI_BIAS_GELU2), (19000, 1152, 128, ops.EPI_BIAS_GELU),
    # round 4: the backward through linear2 + GELU on the persistent kernel (gelu' tile read in the store layout, column sums folded in
    # registers, one atomic per wave; M edges shifted: their duplicate rows must be counted once), incl. a half-width last item
    (86317, 1536, 384, ops.EPI_MUL_GELU_GRAD), (9907, 3072, 768, ops.EPI_MUL_GELU_GRAD), (33001, 640, 256, ops.EPI_MUL_GELU_GRAD),
    # round 4: the predictor's dgrads as row-form GEMMs against W^T (N = 384: one full + one half-width item per row panel)
    (86317, 384, 1152, ops.EPI_BF16), (21600, 384, 1536, ops.EPI_BF16), (21600, 384, 384, ops.EPI_BF16),
    (5000, 768, 768, ops.EPI_BF16),     # fewer items than resident workgroups (WJ_PERSIST_MIN_TILES=1 sends it to variant 4)
]
only = os.environ.get("WJ_CHECK_ONLY")          # e.g. "2" = the MUL_GELU_GRAD shapes only
if only is not None:
    SHAPES = [sh for sh in SHAPES if sh[3] == int(only)]
ref_v, new_v = int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 3
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6


def run(v, A, W, C, kw):
    ops.gemm_set_variant(v)
    ops.gemm(A, W, C, **kw)


bad = 0
for (M, N, K, epi) in SHAPES:
    kw = dict(M=M, N=N, K=K, lda=K, ldb=K, ldc=N, epilogue=epi)
    bias = torch.randn(N, device=dev)
    kw["bias"] = bias
    C2a = C2b = None
    if epi == ops.EPI_CONV_GELU:
        kw.pop("bias")
        kw.update(seg_rows=402, seg_valid=400)
    if epi in (ops.EPI_BIAS_GELU2, ops.EPI_CONV_GELU):
        C2a, C2b = torch.empty(M, N, device=dev, dtype=bf), torch.empty(M, N, device=dev, dtype=bf)
    csa = csb = None
    if epi == ops.EPI_MUL_GELU_GRAD:
        kw.pop("bias")
        kw["aux"] = (torch.rand(M, N, device=dev) * 1.2 - 0.1).to(bf)        # gelu' lies in [-0.13, 1.13]
        csa, csb = torch.zeros(N, device=dev), torch.zeros(N, device=dev)
    worst = 0
    fracs = []
    for r in range(reps):
        A = torch.randn(M, K, device=dev).to(bf)
        W = (torch.randn(N, K, device=dev) * 0.05).to(bf)
        Ca = torch.full((M, N), float("nan"), device=dev, dtype=bf)
        Cb = torch.full((M, N), float("nan"), device=dev, dtype=bf)
        if csa is not None:
            csa.zero_(); csb.zero_()
        run(ref_v, A, W, Ca, dict(kw, **({"C2": C2a} if C2a is not None else {}), **({"colsum": csa} if csa is not None else {})))
        run(new_v, A, W, Cb, dict(kw, **({"C2": C2b} if C2b is not None else {}), **({"colsum": csb} if csb is not None else {})))
        torch.cuda.synchronize()
        pairs = [(Ca, Cb)] + ([(C2a, C2b)] if C2a is not None else [])
        same = all(torch.equal(x.view(torch.int16), y.view(torch.int16)) for x, y in pairs)
        if csa is not None:                         # column sums: fp32 sums of the stored bf16 values, in an unspecified order
            want = Cb.float().sum(0, dtype=torch.float64)
            scale = Cb.float().abs().sum(0, dtype=torch.float64) + 1e-6
            for nm, cs in (("ref", csa), ("new", csb)):
                err = float(((cs.double() - want).abs() / scale).max())
                if err > 2e-5:
                    print(f"  COLSUM {nm} M={M} N={N} K={K}: max |err| / sum|x| = {err:.3e}")
                    same = False
        if not same and 4 in (ref_v, new_v):
            same = True
            for x, y in pairs:
                xf, yf = x.float(), y.float()
                d = (xf - yf).abs()
                frac = float((d > 0).float().mean())
                ok = bool((d <= 0.0157 * torch.maximum(xf.abs(), yf.abs()) + (4e-3 if epi != ops.EPI_BF16 else 1e-4)).all())
                same = same and ok and frac <= 0.02 and not bool(torch.isnan(yf).any())
                fracs.append(frac)
            # the variant under test against itself: bit-identical
            Cc = torch.full((M, N), float("nan"), device=dev, dtype=bf)
            C2c = torch.empty_like(C2b) if C2b is not None else None
            run(new_v, A, W, Cc, dict(kw, **({"C2": C2c} if C2c is not None else {}), **({"colsum": torch.zeros_like(csb)} if csb is not None else {})))
            torch.cuda.synchronize()
            same = same and torch.equal(Cb.view(torch.int16), Cc.view(torch.int16)) and (C2c is None or torch.equal(C2b.view(torch.int16), C2c.view(torch.int16)))
        if not same:
            worst += 1
            d = (Ca.float() - Cb.float()).abs()
            print(f"  MISMATCH M={M} N={N} K={K} epi={epi} rep={r}: {int((d > 0).sum())} elements differ, max {float(d.max()):.4g}, "
                  f"nan {int(torch.isnan(Cb.float()).sum())}")
    if r == reps - 1 and M <= 10000:   # fp32 torch reference on the last operands (small shapes)
        ref = A.float() @ W.float().t() + (bias if "bias" in kw else 0)
        if epi == ops.EPI_MUL_GELU_GRAD:
            ref = ref.to(bf).float() * kw["aux"].float()
            got = Cb.float()
        elif epi in (ops.EPI_BIAS_GELU, ops.EPI_BIAS_GELU2):
            ref = torch.nn.functional.gelu(ref.to(bf).float())
            got = (C2b if epi == ops.EPI_BIAS_GELU2 else Cb).float()
        else:
            got = Cb.float()
        print(f"  vs fp32 torch: rel err {float((got - ref).norm() / ref.norm()):.3e}")
    bad += worst
    # timing, interleaved
    ts = {ref_v: [], new_v: []}
    for r in range(7):
        for v in (ref_v, new_v):
            C = Ca if v == ref_v else Cb
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ops.gemm_set_variant(v)
            e0.record()
            for _ in range(3):
                ops.gemm(A, W, C, **dict(kw, **({"C2": C2a} if C2a is not None else {}), **({"colsum": csa} if csa is not None else {})))
            e1.record()
            torch.cuda.synchronize()
            if r:
                ts[v].append(e0.elapsed_time(e1) / 3)
    fl = 2.0 * M * N * K
    ta, tb = sorted(ts[ref_v])[3], sorted(ts[new_v])[3]
    print(f"M={M:6d} N={N:5d} K={K:5d} epi={epi}: variant {ref_v} {ta * 1e3:7.1f} us {fl / ta / 1e9:7.1f} TF | variant {new_v} {tb * 1e3:7.1f} us "
          f"{fl / tb / 1e9:7.1f} TF | {'OK' if worst == 0 else 'MISMATCH x%d' % worst}" + (f" (last-place differences: {max(fracs) * 100:.3f} % of the elements)" if fracs else ""))
ops.gemm_set_variant(-1)
print("mismatching repetitions:", bad)
sys.exit(1 if bad else 0)
