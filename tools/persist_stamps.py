#!/usr/bin/env python3
"""Diagnostic (GPU box): where the persistent GEMM's workgroups spend their time.  WJ_PERSIST_STAMPS=1 makes every workgroup
record s_memrealtime (100 MHz) at its start, after its prologue and after every output tile; this prints, per shape, the
dispatch skew, the prologue, the first / steady / last tile durations, the tiles pulled per workgroup and the tail."""
import ctypes
import os
os.environ.setdefault("WAVJEPA_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "wavjepa_amd", "lib", "libwavjepa_hip_lab.so"))  # laboratory build: honours the WJ_* A/B switches, exports the stamp reader
import sys

os.environ["WJ_PERSIST_STAMPS"] = "1"
import numpy as np  # noqa: E402
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wavjepa_amd import _abi, ops  # noqa: E402

dev = torch.device("cuda:0")
bf = torch.bfloat16
SHAPES = [(51200, 2304, 768, ops.EPI_BF16), (51200, 768, 768, ops.EPI_BF16), (51200, 768, 3072, ops.EPI_BF16),
          (51200, 3072, 768, ops.EPI_BIAS_GELU), (86317, 1536, 384, ops.EPI_BIAS_GELU2), (86317, 1152, 384, ops.EPI_BF16),
          (8192, 8192, 8192, ops.EPI_BF16)]
lib = _abi.load()
lib.wj_debug_persist_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
N = 64
for (M, Nn, K, epi) in SHAPES:
    A = torch.randn(M, K, device=dev).to(bf)
    W = (torch.randn(Nn, K, device=dev) * 0.05).to(bf)
    C = torch.empty(M, Nn, device=dev, dtype=bf)
    C2 = torch.empty(M, Nn, device=dev, dtype=bf) if epi == ops.EPI_BIAS_GELU2 else None
    bias = torch.randn(Nn, device=dev)
    kw = dict(M=M, N=Nn, K=K, lda=K, ldb=K, ldc=Nn, epilogue=epi, bias=bias)
    if C2 is not None:
        kw["C2"] = C2
    ops.gemm_set_variant(4)
    for _ in range(3):
        ops.gemm(A, W, C, **kw)
    torch.cuda.synchronize()
    # poison, then ONE launch
    buf = np.zeros(2 * 256 * N, dtype=np.uint64)
    ops.gemm(A, W, C, **kw)
    rc = lib.wj_debug_persist_stamps(buf.ctypes.data, 2 * 256 * N)
    assert rc == 0
    ph = buf[256 * N:].reshape(256, N).astype(np.int64)[:, :56].reshape(256, 7, 8)    # tiles 0-6: end of each phase of K tiles 0, 1
    s = buf[:256 * N].reshape(256, N).astype(np.int64)
    act = int(os.environ.get("WJ_PERSIST_ACTIVE", "32"))                    # diagnostic: only the first `act` workgroups of every XCD work
    work = np.nonzero((np.arange(256) >> 3) < act)[0]
    clk = s[:, N - 2:].copy()
    s[:, N - 2:] = 0
    t0 = s[:, 0].min()
    # stamps: [start, prologue end, then per tile: after first K-tile pair, end of K loop, end of epilogue]
    det = s[:, 2:2 + 3 * 6].reshape(256, 6, 3)
    prev_end = np.concatenate([s[:, 1:2], det[:, :-1, 2]], axis=1)          # end of previous epilogue (or prologue)
    pair = (det[:, 1:5, 0] - prev_end[:, 1:5]) / 100.0                      # tiles 1..4: first two K tiles after an epilogue
    rest = (det[:, 1:5, 1] - det[:, 1:5, 0]) / 100.0
    epi_t = (det[:, 1:5, 2] - det[:, 1:5, 1]) / 100.0
    last_rt = np.array([s[w, :N - 2][s[w, :N - 2] > 0].max() for w in range(256)])
    ghz = (clk[:, 1] - clk[:, 0]) / np.maximum(1, (last_rt - s[:, 1])) / 10.0
    ghz, pair, rest, epi_t = ghz[work], pair[work], rest[work], epi_t[work]
    print(f"   shader clock between the prologue and the last stamp: median {np.median(ghz):.3f} GHz (min {ghz.min():.3f}, max {ghz.max():.3f})")
    print(f"   tiles 1-4 (median over WGs): first K-tile pair {np.median(pair):.2f} us, rest of K loop {np.median(rest):.2f} us "
          f"({np.median(rest) / max(1, K // 64 - 2):.3f} us/K-tile), epilogue issue {np.median(epi_t):.2f} us")
    # phases of the first K-tile pair, tiles 1-4: time from the end of the previous epilogue to the end of phase 0 of K tile 0, then
    # phase by phase
    ph_d = np.diff(np.concatenate([prev_end[:, 1:5, None], ph[:, 1:5, :]], axis=2), axis=2) / 100.0
    print("   first K-tile pair phase by phase (us, median): " + " ".join(f"{v:.2f}" for v in np.median(ph_d[work].reshape(-1, 8), axis=0)))
    s = np.concatenate([s[:, :2], s[:, 4::3]], axis=1)                      # keep [start, prologue, tile ends] for the summary below
    ntile = np.array([int(((s[w, 2:] > s[w, 1]) & (s[w, 2:] - t0 < 10_000_000)).sum()) for w in range(256)])
    start = (s[:, 0] - t0) / 100.0
    pro = (s[:, 1] - s[:, 0]) / 100.0
    first = (s[:, 2] - s[:, 1]) / 100.0
    steady = []
    last = []
    ends = []
    for w in range(256):
        k = ntile[w]
        d = np.diff(s[w, 1:2 + k]) / 100.0
        if k > 2:
            steady += list(d[1:-1])
        if k > 1:
            last.append(d[-1])
        ends.append((s[w, 1 + k] - t0) / 100.0)
    ends = np.array(ends)
    nk = K // 64
    print(f"M={M} N={Nn} K={K} epi={epi}: kernel span {ends.max():.1f} us; start skew p50 {np.median(start):.2f} max {start.max():.2f} us; "
          f"prologue p50 {np.median(pro):.2f} us; first tile p50 {np.median(first):.2f}; steady tile p50 "
          f"{(np.median(steady) if steady else float('nan')):.2f} ({(np.median(steady) / nk if steady else float('nan')):.3f} us/K-tile); last p50 "
          f"{(np.median(last) if last else float('nan')):.2f}; tiles/WG min {ntile.min()} max {ntile.max()} sum {ntile.sum()}; "
          f"end p5 {np.percentile(ends, 5):.1f} p50 {np.median(ends):.1f} max {ends.max():.1f}")
ops.gemm_set_variant(-1)
