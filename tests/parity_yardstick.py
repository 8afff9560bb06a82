"""Yardstick form of the bf16 parity bounds (test infrastructure; used by tests/test_jepa_gpu.py and tests/grad_yardstick_sweep.py).

A bf16 pipeline cannot be held to a fixed distance from another bf16 pipeline: how far either sits from the fp32 truth depends on
the batch, the masks and the depth of the stack that produced a tensor.  What CAN be stated is that the HIP path is as close to the
fp32 truth as the oracle's own bf16 flow is on the SAME draw (same weights, clips, masks):

    d(HIP, oracle-fp32)  <  FACTOR x d(oracle-bf16, oracle-fp32) + EPS          per activation / per parameter-gradient group

with d = relative L2.  The oracle-bf16 flow is the reference's autocast dtype flow restated op by op (oracle/jepa_oracle.py), so the
right-hand side is "what bf16 costs the reference itself" -- a bound that moves with the draw instead of failing on some of them.
"""
from typing import Callable, Dict, Iterable

import torch

GRAD_FACTOR = 1.25        # HIP may sit 25 % further from the fp32 gradient than the oracle's bf16 flow does ...
GRAD_EPS = 5e-4           # ... plus the run-to-run noise of fp32 split-K atomics and the last-place differences of tiny groups
# Per-channel conv stacks of the 3-clip ConvChannelFeatureExtractor test: 99 tokens per channel and clip, a third of them context --
# the group's bf16 error is a handful of independent rounding events, so d_hip and d_orc are two DRAWS from one distribution rather
# than two equal numbers.  Measured over 2 x 16 unpinned mask draws (profiles/r05_grad_yardstick.txt): d_orc 1.0-2.2 %, d_hip
# 0.8-2.3 % (the same range; two sweeps of the round), per-draw ratio 0.79-1.36.  Every other group (and every group of the BASE model at 64 clips: ratio
# 0.69-0.81) stays inside 1.25.
GRAD_FACTOR_BY_PREFIX = {"extract_audio.cnns": 1.5}
ACT_FACTOR = 1.1
ACT_EPS = 2e-4
PAIR_FACTOR = 1.5         # activations: two INDEPENDENT bf16 pipelines, each d from the truth, sit ~sqrt(2) d apart (1.5 with slack)


def rel(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def group_errors(got: Dict[str, torch.Tensor], want: Dict[str, torch.Tensor], names: Iterable[str],
                 group_of: Callable[[str], str]) -> Dict[str, float]:
    """Relative L2 distance per parameter group: sqrt(sum |got - want|^2 / sum |want|^2) over the group's tensors."""
    num: Dict[str, float] = {}
    den: Dict[str, float] = {}
    for k in names:
        g = group_of(k)
        a, b = got[k].detach().double(), want[k].detach().double()
        num[g] = num.get(g, 0.0) + float((a - b).pow(2).sum())
        den[g] = den.get(g, 0.0) + float(b.pow(2).sum())
    return {g: (num[g] / max(den[g], 1e-300)) ** 0.5 for g in num}


def oracle_fp32_grads(J, P: Dict[str, torch.Tensor], audio: torch.Tensor, ctx, tgt, vis, names, **oracle_kw):
    """(forward dict, {name: gradient}) of the oracle in fp32 mode on detached copies of P (P's own .grad fields are not touched)."""
    P32 = {k: v.detach().clone() for k, v in P.items()}
    for k in names:
        P32[k].requires_grad_(True)
    ref32 = J.jepa_forward(P32, audio.float(), ctx, tgt, vis, mode="fp32", **oracle_kw)
    ref32["loss"].backward()
    return ref32, {k: P32[k].grad for k in names}


def grad_yardstick(hip: Dict[str, torch.Tensor], orc_bf16: Dict[str, torch.Tensor], orc_fp32: Dict[str, torch.Tensor], names,
                   group_of: Callable[[str], str]) -> Dict[str, Dict[str, float]]:
    """Per group: d_hip = d(HIP, fp32), d_orc = d(oracle-bf16, fp32), pair = d(HIP, oracle-bf16), and the ratio d_hip / d_orc."""
    d_hip = group_errors(hip, orc_fp32, names, group_of)
    d_orc = group_errors(orc_bf16, orc_fp32, names, group_of)
    pair = group_errors(hip, orc_bf16, names, group_of)
    return {g: dict(d_hip=d_hip[g], d_orc=d_orc[g], pair=pair[g], ratio=d_hip[g] / max(d_orc[g], 1e-30)) for g in d_hip}


def grad_factor(group: str, factor: float = GRAD_FACTOR) -> float:
    return max([factor] + [f for p, f in GRAD_FACTOR_BY_PREFIX.items() if group.startswith(p)])


def grad_bound_ok(group: str, r: Dict[str, float], factor: float = GRAD_FACTOR, eps: float = GRAD_EPS) -> bool:
    return r["d_hip"] < grad_factor(group, factor) * r["d_orc"] + eps


def assert_grad_yardstick(table: Dict[str, Dict[str, float]], factor: float = GRAD_FACTOR, eps: float = GRAD_EPS) -> None:
    """(d(HIP, oracle-bf16) needs no bound of its own: the triangle inequality holds it below (1 + factor) x d_orc + eps.)"""
    for g, r in table.items():
        assert grad_bound_ok(g, r, factor, eps), (g, r, grad_factor(g, factor))
