#!/usr/bin/env python3
"""GEMM micro-benchmark over the shapes of the WavJEPA step (run on the GPU box): TFLOP/s per shape/layout/epilogue,
interleaved rounds in one process, HIP-event timing on the launch stream."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wavjepa_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
bf = torch.bfloat16
SHAPES = [
    # tag, a_trans, b_trans, epilogue, M, N, K
    ("enc qkv       NN", 0, 0, ops.EPI_BF16, 51200, 2304, 768),
    ("enc out_proj  NN", 0, 0, ops.EPI_BF16, 51200, 768, 768),
    ("enc linear1   NN", 0, 0, ops.EPI_BIAS_GELU2, 51200, 3072, 768),
    ("enc linear2   NN", 0, 0, ops.EPI_BF16, 51200, 768, 3072),
    ("dec qkv       NN", 0, 0, ops.EPI_BF16, 204800, 1152, 384),
    ("dec linear1   NN", 0, 0, ops.EPI_BIAS_GELU2, 204800, 1536, 384),
    ("dec linear2   NN", 0, 0, ops.EPI_BF16, 204800, 384, 1536),
    ("enc dh        NT", 0, 1, ops.EPI_MUL_GELU_GRAD, 51200, 3072, 768),
    ("enc dx1       NT", 0, 1, ops.EPI_ADD_F32, 51200, 768, 3072),
    ("enc dxin      NT", 0, 1, ops.EPI_ADD_F32, 51200, 768, 2304),
    ("dec dh        NT", 0, 1, ops.EPI_MUL_GELU_GRAD, 204800, 1536, 384),
    ("dec dx1       NT", 0, 1, ops.EPI_ADD_F32, 204800, 384, 1536),
    ("enc dWqkv     TT", 1, 1, ops.EPI_ATOMIC_F32, 2304, 768, 51200),
    ("enc dW1       TT", 1, 1, ops.EPI_ATOMIC_F32, 3072, 768, 51200),
    ("enc dW2       TT", 1, 1, ops.EPI_ATOMIC_F32, 768, 3072, 51200),
    ("dec dW1       TT", 1, 1, ops.EPI_ATOMIC_F32, 1536, 384, 204800),
    # ragged step (AudioSet masker, 256 clips): ~9.9k context rows for the student, ~86k visible rows for the predictor
    ("rag enc qkv   NN", 0, 0, ops.EPI_BF16, 9907, 2304, 768),
    ("rag enc out   NN", 0, 0, ops.EPI_BF16, 9907, 768, 768),
    ("rag enc lin1  NN", 0, 0, ops.EPI_BIAS_GELU2, 9907, 3072, 768),
    ("rag enc lin2  NN", 0, 0, ops.EPI_BF16, 9907, 768, 3072),
    ("rag enc dh    NT", 0, 1, ops.EPI_MUL_GELU_GRAD, 9907, 3072, 768),
    ("rag enc dx1   NT", 0, 1, ops.EPI_ADD_F32, 9907, 768, 3072),
    ("rag enc dxin  NT", 0, 1, ops.EPI_ADD_F32, 9907, 768, 2304),
    ("rag enc do    NT", 0, 1, ops.EPI_BF16, 9907, 768, 768),
    ("rag enc dWqkv TT", 1, 1, ops.EPI_ATOMIC_F32, 2304, 768, 9907),
    ("rag enc dW1   TT", 1, 1, ops.EPI_ATOMIC_F32, 3072, 768, 9907),
    ("rag enc dW2   TT", 1, 1, ops.EPI_ATOMIC_F32, 768, 3072, 9907),
    ("rag enc dWo   TT", 1, 1, ops.EPI_ATOMIC_F32, 768, 768, 9907),
    ("rag dec qkv   NN", 0, 0, ops.EPI_BF16, 86317, 1152, 384),
    ("rag dec out   NN", 0, 0, ops.EPI_BF16, 86317, 384, 384),
    ("rag dec lin1  NN", 0, 0, ops.EPI_BIAS_GELU2, 86317, 1536, 384),
    ("rag dec lin2  NN", 0, 0, ops.EPI_BF16, 86317, 384, 1536),
    ("rag dec dh    NT", 0, 1, ops.EPI_MUL_GELU_GRAD, 86317, 1536, 384),
    ("rag dec dx1   NT", 0, 1, ops.EPI_ADD_F32, 86317, 384, 1536),
    ("rag dec dxin  NT", 0, 1, ops.EPI_ADD_F32, 86317, 384, 1152),
    ("rag dec do    NT", 0, 1, ops.EPI_BF16, 86317, 384, 384),
    ("rag dec dWqkv TT", 1, 1, ops.EPI_ATOMIC_F32, 1152, 384, 86317),
    ("rag dec dW1   TT", 1, 1, ops.EPI_ATOMIC_F32, 1536, 384, 86317),
    ("rag dec dW2   TT", 1, 1, ops.EPI_ATOMIC_F32, 384, 1536, 86317),
    ("rag dec dWo   TT", 1, 1, ops.EPI_ATOMIC_F32, 384, 384, 86317),
    ("square 8192   NN", 0, 0, ops.EPI_BF16, 8192, 8192, 8192),
    ("conv1 fwd     NN", 0, 0, ops.EPI_CONV_GELU, 256 * 3216, 512, 1536),
    ("conv1 wgrad   TT", 1, 1, ops.EPI_ATOMIC_F32, 512, 1536, 256 * 3216),
]


def make(tag, at, bt, epi, M, N, K):
    conv = tag.startswith("conv1 fwd")
    convw = tag.startswith("conv1 wgrad")
    if conv:
        A = torch.randn(2 * M + 16, 512, device=dev).to(bf)
        lda = 1024
    elif at:
        A = torch.randn(K, M, device=dev).to(bf)
        lda = M
    else:
        A = torch.randn(M, K, device=dev).to(bf)
        lda = K
    if convw:
        B = torch.randn(2 * K + 16, 512, device=dev).to(bf)
        ldb = 1024
    elif bt:
        B = (torch.randn(K, N, device=dev) * 0.05).to(bf)
        ldb = N
    else:
        B = (torch.randn(N, K, device=dev) * 0.05).to(bf)
        ldb = K
    f32out = epi in (ops.EPI_ADD_F32, ops.EPI_ATOMIC_F32)
    C = torch.zeros(M, N, device=dev, dtype=torch.float32 if f32out else bf)
    if ALIAS:
        lda = ldb = 0
    kw = dict(M=M, N=N, K=K, lda=lda, ldb=ldb, ldc=N, a_trans=at, b_trans=bt, epilogue=epi)
    keep = [A, B, C]
    if epi in (ops.EPI_BIAS_GELU2, ops.EPI_CONV_GELU):
        C2 = torch.empty_like(C); kw["C2"] = C2; keep.append(C2)
    if epi in (ops.EPI_BF16, ops.EPI_BIAS_GELU2):
        bias = torch.randn(N, device=dev); kw["bias"] = bias; keep.append(bias)
    if epi == ops.EPI_MUL_GELU_GRAD:
        aux = torch.randn(M, N, device=dev).to(bf); kw["aux"] = aux; keep.append(aux)
    if epi == ops.EPI_ADD_F32:
        aux = torch.randn(M, N, device=dev); kw["aux"] = aux; keep.append(aux)
    if epi == ops.EPI_ATOMIC_F32:
        kw["split_k"] = ops.pick_split_k(M, N, K)
    if epi == ops.EPI_CONV_GELU:
        kw["seg_rows"], kw["seg_valid"] = 3216, 3214
    return (A, B, C), kw, keep


ALIAS = "--alias" in sys.argv   # lda = ldb = 0: every operand row aliases one row -> no memory traffic (structure ceiling)


def main():
    sel = [a for a in sys.argv[1:] if not a.startswith("--")] or None
    cases = [s for s in SHAPES if sel is None or any(x in s[0] for x in sel)]
    built = [(s, make(*s)) for s in cases]
    rounds = 5
    times = {s[0]: [] for s in cases}
    for r in range(rounds + 1):
        for s, ((A, B, C), kw, keep) in built:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                ops.gemm(A, B, C, **kw)
            e1.record()
            torch.cuda.synchronize()
            if r:
                times[s[0]].append(e0.elapsed_time(e1) / 3)
    tot_f = tot_t = 0.0
    for s in cases:
        t = sorted(times[s[0]])[len(times[s[0]]) // 2]
        fl = 2.0 * s[4] * s[5] * s[6]
        tot_f += fl; tot_t += t
        print(f"{s[0]:22s} M={s[4]:7d} N={s[5]:5d} K={s[6]:7d}  {t * 1000:8.1f} us  {fl / t / 1e9:7.1f} TFLOP/s")
    print(f"{'ALL':22s} {tot_f / tot_t / 1e9:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
