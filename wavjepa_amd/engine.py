"""Forward/backward orchestration of the WavJEPA pre-training step on one MI355X.

This is the host side of the hot path: a fixed sequence of C-ABI kernel launches over a pre-allocated activation
arena (no autograd graph, no allocator traffic in the step, no host synchronisation), mirroring
reference wavjepa/jepa.py:365-419 (forward), :230-270 (teacher targets), :335-362 (loss) and the implicit autograd
backward of those ops.  Activations are token-major bf16/fp32 buffers sized once per batch size; the residual
stream is fp32 and every GEMM operand is bf16, exactly the dtype flow bf16 autocast produces on the reference.

Conv front-end layout: channels-last [clip][row][C] with per-clip row counts P_l chosen so that
P_{l-1} = stride_l * P_l; a strided conv is then ONE GEMM with lda = stride*C and K = k*C over all clips
(rows >= L_l of a clip are padding, kept at zero).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import ops
from .params import FlatParams


def _empty(*shape, dtype, device):
    """Arena buffers are uninitialised by design (every kernel writes what a later kernel reads).  WJ_ARENA_FILL=nan poisons
    them instead, so that a read of never-written memory shows up as NaN in the results (tools/debug_order.py)."""
    import os
    if os.environ.get("WJ_ARENA_FILL", "") == "nan":
        if not dtype.is_floating_point:          # fp8 operand bytes: 0x7f is the e4m3 NaN encoding
            return torch.full(shape, 0x7f, dtype=dtype, device=device)
        return torch.full(shape, float("nan"), dtype=dtype, device=device)
    return torch.empty(*shape, dtype=dtype, device=device)


@dataclass
class EngineConfig:
    conv_spec: Sequence[Tuple[int, int, int]]
    in_channels: int
    n_samples: int          # waveform samples per clip (32159)
    d_enc: int
    h_enc: int
    l_enc: int
    d_dec: int
    h_dec: int
    l_dec: int
    top_k: int
    streams: int = 1        # audio channels run as separate MONO conv stacks (ConvChannelFeatureExtractor); tokens per clip = streams * frames
    conv_prefixes: Sequence[str] = ("extract_audio.cnn.",)   # state_dict prefix of every stream's stack (one entry: shared weights)
    ln_eps: float = 1e-6    # transformer layers (reference types/wavjepa_configs.py:37)
    norm_eps: float = 1e-5  # feature_norms / final norms (nn.LayerNorm default)


@dataclass
class MaskPlan:
    """Device-side masks + the index lists of the boolean-mask gather (reference jepa.py:399,425-428)."""
    ctx_u8: torch.Tensor    # [N, T]     1 = NOT context (key masked for the student)
    tgt_u8: torch.Tensor    # [N, G, T]  1 = target position
    vis_u8: torch.Tensor    # [N*G, T]   1 = key masked for the predictor
    keep: torch.Tensor      # int32 [n_ctx] flat (b*T+t) of context rows, ascending
    inv: torch.Tensor       # int32 [N*T]  position in `keep` or -1
    n_ctx: int
    N: int = 0              # clips, target groups per clip (= target_indices.shape[1], reference jepa.py:402-405), tokens
    G: int = 0
    T: int = 0
    # ragged execution (visible tokens only).  The student's non-context rows are dropped by jepa.py:399 and the
    # predictor's non-target rows carry zero loss weight (jepa.py:356); being key-masked they influence nothing else,
    # so the student runs on the n_ctx context rows and the predictor on the n_dec visible rows, packed per sequence.
    enc_off: Optional[torch.Tensor] = None    # int32 [N+1]     context rows before clip b
    dec_rows: Optional[torch.Tensor] = None   # int32 [n_dec]   dense (b*G+g)*T+t of every visible predictor row, ascending
    dec_off: Optional[torch.Tensor] = None    # int32 [N*G+1]
    dec_map: Optional[torch.Tensor] = None    # int32 [N*G*T]   packed row of a dense position or -1
    n_dec: int = 0
    # predictor rows whose OUTPUT is used (targets): after the last layer's attention only these go on (context rows of the
    # last layer serve as keys / values only)
    tgt_rows: Optional[torch.Tensor] = None   # int32 [n_tgt]  packed predictor row of every target, ascending
    tgt_inv: Optional[torch.Tensor] = None    # int32 [n_dec]  position in tgt_rows or -1
    tgt_dense: Optional[torch.Tensor] = None  # int32 [n_tgt]  dense (b*G+g)*T+t of those rows (loss row list)
    n_tgt: int = 0
    max_enc: int = 0                          # longest context / visible sequence (attention length bound)
    max_dec: int = 0
    ragged_ok: bool = False                   # False when a target position is key-masked: the dense path must be used
    ctx_np: Optional[np.ndarray] = None       # host copy of the context mask (True = NOT context), for the conv row lists


class _PinnedStaging:
    """A small ring of page-locked host buffers for the per-step index uploads.  A copy from PAGEABLE memory is staged by the runtime
    and holds the host until everything queued on the stream before it has run (seen in the rocprofv3 trace as 0.2 ms of idle GPU
    between the crop kernel and the first front-end kernel of every step, and as a 1.7 ms longer one-stream step when the upload was
    issued behind the queued teacher forward); from page-locked memory it is just one more stream operation.  A buffer is reused only
    after the event recorded behind its last copy has completed."""
    RING = 6

    def __init__(self):
        self.slots = []            # [tensor (uint8, pinned), event or None]
        self.next = 0

    def get(self, nbytes: int):
        if len(self.slots) < self.RING:
            self.slots.append([torch.empty(max(nbytes * 3 // 2, 1 << 20), dtype=torch.uint8, pin_memory=True), None])
            return self.slots[-1]
        slot = self.slots[self.next % self.RING]
        self.next += 1
        if slot[1] is not None:
            slot[1].synchronize()
        if slot[0].numel() < nbytes:
            slot[0] = torch.empty(nbytes * 3 // 2, dtype=torch.uint8, pin_memory=True)
        return slot


_STAGING = _PinnedStaging()
_PINNED_UPLOADS = __import__("os").environ.get("WJ_PINNED_UPLOAD", "1") != "0"     # 0: pageable staging (A/B runs)


_UPLOAD_STREAM = __import__("os").environ.get("WJ_UPLOAD_STREAM", "1") != "0"       # 0: uploads on the compute stream (A/B runs)
_UPLOAD_STREAMS: dict = {}


def _upload_stream(device):
    d = torch.device(device)
    key = d.index if d.index is not None else torch.cuda.current_device()
    if key not in _UPLOAD_STREAMS:
        _UPLOAD_STREAMS[key] = torch.cuda.Stream(device=d)
    return _UPLOAD_STREAMS[key]


def pack_upload(arrays: Sequence[np.ndarray], device) -> List[torch.Tensor]:
    """One host -> device copy for a set of small index / mask arrays (uint8 / int32): packed into one byte buffer at 256-byte offsets,
    copied once (from page-locked staging memory when the target is a GPU), returned as typed views of the device buffer (every view
    keeps it alive).  A step builds ~25 such lists (mask plan, sparse-conv row lists); as separate `.to(device)` calls each was its own
    copy kernel on the stream."""
    offs, total = [], 0
    for a in arrays:
        offs.append(total)
        total += (a.nbytes + 255) // 256 * 256
    total = max(total, 256)
    on_gpu = torch.device(device).type == "cuda" and _PINNED_UPLOADS
    if on_gpu:
        slot = _STAGING.get(total)
        host = slot[0].numpy()[:total]
    else:
        host = np.zeros(total, dtype=np.uint8)
    for a, o in zip(arrays, offs):
        host[o:o + a.nbytes] = np.ascontiguousarray(a).view(np.uint8).reshape(-1)
    if on_gpu and _UPLOAD_STREAM:
        # On a stream of its own: queued on the compute stream the copy starts only when that stream gets there, and the kernel behind it
        # waits out the copy engine's latency (~70 us of idle GPU in front of every step's conv0 launches, tools/trace_gaps.py).  The host
        # runs tens of milliseconds ahead of the GPU, so on its own stream the copy is long done when the compute stream reaches the wait.
        main = torch.cuda.current_stream(device)
        up = _upload_stream(device)
        with torch.cuda.stream(up):
            dev_buf = slot[0][:total].to(device, non_blocking=True)
        slot[1] = torch.cuda.Event()
        slot[1].record(up)
        main.wait_event(slot[1])
        dev_buf.record_stream(main)
    elif on_gpu:
        dev_buf = slot[0][:total].to(device, non_blocking=True)
        slot[1] = torch.cuda.Event()
        slot[1].record(torch.cuda.current_stream(device))
    else:
        dev_buf = torch.from_numpy(host).to(device, non_blocking=True)
    out = []
    for a, o in zip(arrays, offs):
        t = dev_buf[o:o + a.nbytes]
        out.append(t.view({1: torch.uint8, 4: torch.int32}[a.dtype.itemsize]).reshape(a.shape) if a.nbytes else
                   torch.zeros(a.shape, dtype={1: torch.uint8, 4: torch.int32}[a.dtype.itemsize], device=device))
    return out


def make_mask_plan(ctx_mask, target_indices, vis_mask, device) -> MaskPlan:
    """Build the plan.  CPU masks (the data-loader case) need no device synchronisation."""
    if isinstance(ctx_mask, torch.Tensor) and ctx_mask.is_cuda:
        # device masks: the index lists need their sizes on the host, like the reference's x[~mask] (one sync)
        ctx_mask, target_indices, vis_mask = ctx_mask.cpu(), target_indices.cpu(), vis_mask.cpu()
    ctx_np = np.ascontiguousarray(np.asarray(ctx_mask, dtype=bool))
    tgt_np = np.ascontiguousarray(np.asarray(target_indices, dtype=bool))
    vis_np = np.ascontiguousarray(np.asarray(vis_mask, dtype=bool))
    if ctx_np.ndim != 2 or tgt_np.ndim != 3 or tgt_np.shape[0] != ctx_np.shape[0] or tgt_np.shape[2] != ctx_np.shape[1]:
        raise ValueError(f"masks: expected ctx [N, T] and target_indices [N, G, T], got {ctx_np.shape} and {tgt_np.shape}")
    n_clips, n_groups, n_tok = tgt_np.shape
    if vis_np.size != tgt_np.size or vis_np.shape[-1] != n_tok:
        raise ValueError(f"masks: ctx_and_target_masks must hold N*G*T = {tgt_np.size} entries with T last, got {vis_np.shape}")
    vis_np = vis_np.reshape(-1, n_tok)
    keep_np = np.flatnonzero(~ctx_np.reshape(-1)).astype(np.int32)
    inv_np = np.full(ctx_np.size, -1, dtype=np.int32)
    inv_np[keep_np] = np.arange(keep_np.size, dtype=np.int32)
    # ragged index lists
    ctx_len = (~ctx_np).sum(axis=-1)
    enc_off = np.concatenate([[0], np.cumsum(ctx_len)]).astype(np.int32)
    seen = ~vis_np                                             # [N*G, T] rows the predictor computes
    dec_rows = np.flatnonzero(seen.reshape(-1)).astype(np.int32)
    dec_len = seen.sum(axis=-1)
    dec_off = np.concatenate([[0], np.cumsum(dec_len)]).astype(np.int32)
    dec_map = np.full(seen.size, -1, dtype=np.int32)
    dec_map[dec_rows] = np.arange(dec_rows.size, dtype=np.int32)
    ragged_ok = (tgt_np.reshape(vis_np.shape).shape == vis_np.shape
                 and not bool(np.any(tgt_np.reshape(vis_np.shape) & vis_np)))   # every target row is also a key
    if ragged_ok:
        is_tgt = tgt_np.reshape(-1)[dec_rows]                 # per packed predictor row
        tgt_rows = np.flatnonzero(is_tgt).astype(np.int32)
        tgt_inv = np.full(dec_rows.size, -1, dtype=np.int32)
        tgt_inv[tgt_rows] = np.arange(tgt_rows.size, dtype=np.int32)
        tgt_dense = dec_rows[tgt_rows]
    else:
        tgt_rows = tgt_inv = tgt_dense = np.zeros(0, np.int32)

    d = pack_upload([ctx_np.astype(np.uint8), tgt_np.astype(np.uint8), vis_np.astype(np.uint8), keep_np.astype(np.int32),
                     inv_np.astype(np.int32), enc_off.astype(np.int32), dec_rows.astype(np.int32), dec_off.astype(np.int32),
                     dec_map.astype(np.int32), tgt_rows.astype(np.int32), tgt_inv.astype(np.int32), tgt_dense.astype(np.int32)], device)
    return MaskPlan(d[0], d[1], d[2], d[3], d[4],
                    int(keep_np.size), N=n_clips, G=n_groups, T=n_tok, enc_off=d[5], dec_rows=d[6],
                    dec_off=d[7], dec_map=d[8], n_dec=int(dec_rows.size),
                    tgt_rows=d[9], tgt_inv=d[10], tgt_dense=d[11],
                    n_tgt=int(tgt_rows.size),
                    max_enc=int(ctx_len.max()) if ctx_len.size else 0, max_dec=int(dec_len.max()) if dec_len.size else 0,
                    ragged_ok=ragged_ok, ctx_np=ctx_np)


def conv_geometry(n_samples: int, spec) -> Tuple[List[int], List[int]]:
    """Valid output lengths L_l and padded per-clip row counts P_l with P_{l-1} = stride_l * P_l and enough zero
    padding rows for the dgrad taps (P_l - L_l >= ceil(k_l / s_l) - 1, at least 1)."""
    L, cur = [], n_samples
    for _, k, s in spec:
        cur = (cur - k) // s + 1
        L.append(cur)
    n = len(spec)
    need = [max(1, -(-spec[l][1] // spec[l][2]) - 1) for l in range(n)]
    p_last = L[-1] + need[-1]
    while True:
        P = [0] * n
        P[-1] = p_last
        for l in range(n - 2, -1, -1):
            P[l] = P[l + 1] * spec[l + 1][2]
        if all(P[l] >= L[l] + need[l] for l in range(n)):
            return L, P
        p_last += 1


def conv_active_rows(keep: np.ndarray, P: Sequence[int], spec) -> Dict[int, Tuple[np.ndarray, np.ndarray]]:
    """Rows of every conv layer's output that can carry a gradient when only `keep` [N, T] (bool) rows of the LAST layer's
    output do (the student sees the context tokens only, so ~80 % of the conv backward would multiply zeros).

    Returns {l: (act, ext)} of int32 GLOBAL row indices (clip * P[l] + row), ascending:
      act[l]: rows of layer l's output gradient that are written this step (consumed by GELU', wgrad, and cleared after);
      ext[l]: act[l] grown by the dgrad halo (ceil(k/s) - 1 rows after every run): the logical rows of the dgrad GEMM, whose
              outputs s*g + rho, rho < s, are exactly act[l-1].
    Layer 0 (no GEMM dgrad below it): (act, per-clip offsets int32 [N+1] into act)."""
    N, T = keep.shape
    edge = np.diff(np.concatenate([np.zeros((N, 1), np.int8), keep.astype(np.int8), np.zeros((N, 1), np.int8)], axis=1), axis=1)
    clip, start = np.nonzero(edge == 1)
    _, end = np.nonzero(edge == -1)          # same (clip, position) order: the i-th end closes the i-th start

    def expand(clip, start, end, rows_per_clip):
        n = end - start
        if n.size == 0:
            return np.zeros(0, np.int32)
        first = np.cumsum(n) - n
        return (np.repeat(clip.astype(np.int64) * rows_per_clip + start - first, n) + np.arange(int(n.sum()))).astype(np.int32)

    out: Dict[int, Tuple[np.ndarray, Optional[np.ndarray]]] = {}
    for l in range(len(spec) - 1, 0, -1):
        _, k, s = spec[l]
        act = expand(clip, start, end, P[l])
        grown = end + (-(-k // s) - 1)
        if start.size:                       # merge runs that now touch or overlap inside a clip
            new = np.ones(start.size, bool)
            new[1:] = (clip[1:] != clip[:-1]) | (start[1:] > grown[:-1])
            head = np.flatnonzero(new)
            clip, start, grown = clip[head], start[head], np.maximum.reduceat(grown, head)
        out[l] = (act, expand(clip, start, grown, P[l]))
        start, end = start * s, grown * s
    per_clip = np.bincount(clip, weights=end - start, minlength=N).astype(np.int64)
    out[0] = (expand(clip, start, end, P[0]), np.concatenate([[0], np.cumsum(per_clip)]).astype(np.int32))
    return out


class _Layer:
    """Raw device pointers of one transformer layer (weights bf16, biases / LN params fp32, gradients fp32)."""
    __slots__ = ("wqkv", "bqkv", "wo", "bo", "w1", "b1", "w2", "b2", "g1", "be1", "g2", "be2",
                 "gwqkv", "gbqkv", "gwo", "gbo", "gw1", "gb1", "gw2", "gb2", "gg1", "gbe1", "gg2", "gbe2",
                 "wqkv_name", "wo_name", "w1_name", "w2_name",      # parameter names of the weights (student stacks)
                 "wqkvT", "woT", "w1T", "w2T")                      # bf16 W^T shadows (row-form dgrads), when the engine keeps them


def _layer_ptrs(flat: FlatParams, prefix: str, teacher: bool) -> _Layer:
    L = _Layer()
    names = dict(wqkv="self_attn.in_proj_weight", bqkv="self_attn.in_proj_bias", wo="self_attn.out_proj.weight",
                 bo="self_attn.out_proj.bias", w1="linear1.weight", b1="linear1.bias", w2="linear2.weight",
                 b2="linear2.bias", g1="norm1.weight", be1="norm1.bias", g2="norm2.weight", be2="norm2.bias")
    for k, n in names.items():
        full = prefix + n
        is_w = k.startswith("w")
        if teacher:
            setattr(L, k, flat.tptr16(full) if is_w else flat.tptr32(full))
            setattr(L, "g" + k, 0)
        else:
            setattr(L, k, flat.ptr16(full) if is_w else flat.ptr32(full))
            setattr(L, "g" + k, flat.gptr(full))
            if is_w:
                setattr(L, k + "_name", full)
    return L


class _Acts:
    """Saved activations of one transformer layer."""
    __slots__ = ("qkv", "o", "lse", "p", "m1", "r1", "x1", "x1b", "h", "g", "f", "m2", "r2", "x2", "x2b")


class JepaEngine:
    def __init__(self, cfg: EngineConfig, flat: FlatParams, pos_enc: torch.Tensor, pos_dec: torch.Tensor):
        ops.require_gpu()
        self.cfg, self.flat = cfg, flat
        self.dev = flat.device
        self.pos_enc = pos_enc.reshape(-1, cfg.d_enc).contiguous().float().to(self.dev)
        self.pos_dec = pos_dec.reshape(-1, cfg.d_dec).contiguous().float().to(self.dev)
        # fp32 tables on this device are used in place (load_state_dict / the data-parallel broadcast write through); a converted copy
        # (other dtype or device) is refreshed from its source in prepare_weights
        self._pos_src = (pos_enc, pos_dec)
        self.L, self.P = conv_geometry(cfg.n_samples, cfg.conv_spec)
        self.S = max(1, int(cfg.streams))            # channel streams through mono conv stacks
        self.Tc = self.L[-1]                         # conv frames per stream
        self.T = self.S * self.Tc                    # tokens per clip: channel-major "B (C S)" flatten
        self.stacks = list(cfg.conv_prefixes)        # distinct conv stacks; stream c uses stacks[min(c, len - 1)]
        assert len(self.stacks) in (1, self.S)
        self.C = cfg.conv_spec[-1][0]
        assert all(c == self.C for c, _, _ in cfg.conv_spec), "all conv layers must have the same width"
        assert cfg.d_enc % cfg.h_enc == 0 and cfg.d_dec % cfg.h_dec == 0
        self.N = 0
        self.G = 0              # target groups per clip of the arena (taken from the mask plan)
        # second HIP stream: work that is off the critical path (teacher forward; all weight-gradient GEMMs of the
        # backward) runs beside the main chain and fills the tails / write bursts of its kernels (WJ_SIDE_STREAM=0: off)
        import os as _os
        self.use_side = _os.environ.get("WJ_SIDE_STREAM", "1") != "0"
        self._opt_ev = None
        # visible-token (ragged) execution of the student and the predictor; WJ_RAGGED=0 keeps the reference's dense
        # key-masked shapes (identical loss and gradients, ~2x the work)
        self.ragged = _os.environ.get("WJ_RAGGED", "1") != "0"
        self.ragged_step = False
        # conv backward over the active rows only (needs a ragged step and k >= stride in every GEMM conv layer)
        self.sparse_conv = _os.environ.get("WJ_SPARSE_CONV", "1") != "0" and all(k >= st for _, k, st in cfg.conv_spec[1:])
        self._conv_grads_dirty = False
        # GELU' of conv layers 1..n-2 in the epilogue of the sparse dgrad above them (WJ_FUSE_CONV_GELU_BWD=0: separate passes)
        self.fuse_conv_gelu_bwd = _os.environ.get("WJ_FUSE_CONV_GELU_BWD", "1") != "0"
        # last predictor layer: after its attention only the target rows go on (WJ_TRIM_TAIL=0: every visible row)
        self.trim_tail = _os.environ.get("WJ_TRIM_TAIL", "1") != "0"
        self.tail = None
        # MX fp8 forward GEMMs (BASELINE config 5; build-defined numerics, WJ_FP8=1 or engine.fp8 = True): every transformer forward
        # linear whose K is a multiple of 256 runs on block-scaled e4m3 operands (wj_gemm_mxfp8, 2x the bf16 MFMA rate); weights are
        # re-quantised from their bf16 shadows once per step, activations by wj_quantize_mxfp8 in front of each GEMM.  The backward
        # is unchanged: it differentiates the bf16 graph (straight-through), with the saved bf16 activations and bf16 weights.
        self.fp8 = _os.environ.get("WJ_FP8", "0") == "1"
        self._w8: Dict[int, Tuple[torch.Tensor, torch.Tensor, int, int]] = {}     # bf16 weight pointer -> (q, scales, N, K)
        self._a8: Dict[str, Tuple[torch.Tensor, torch.Tensor]] = {}               # per stack: activation scratch (q, scales)
        self.side = self._pick_side_stream() if self.use_side else torch.cuda.Stream(device=self.dev)
        self.has_mapper = "post_extraction_mapper.weight" in flat.by_name
        # dgrads dx = dy . W as ROW-form GEMMs against bf16 W^T shadows of the transformer weights (refreshed once per step by one
        # batched transpose, wj_transpose_bf16): the persistent eight-phase kernel instead of the col-form 256 x 128 schedule.
        # WJ_WT_DGRAD=0 keeps the col-form dgrads (A/B runs).
        self.wt_dgrad = _os.environ.get("WJ_WT_DGRAD", "1") != "0"
        # K-split pairs for the student's 117-tile GEMMs (scratch handed to the main-stream launches of that stack).  Round 6: OFF by default
        # (WJ_PAIR_SPLIT=1 enables) -- on the two-stream step they move nothing (46.85 / 46.87 ms in round 5, 46.09 with / 46.02 without in
        # round 6, interleaved on one box) while costing 30 MB of partial sums per launch (traffic 1.31 x algorithmic) and a spin-wait between
        # workgroups; a one-stream host (WJ_SIDE_STREAM=0) gains 0.3 ms with them.  One default for both, so that the serialised profile
        # measures the kernels the timed step runs.
        self.pair_split = _os.environ.get("WJ_PAIR_SPLIT", "0") == "1"
        self.pair_ws = None
        self._conv_w_fresh = False
        self.defer_folds = _os.environ.get("WJ_DEFER_FOLDS", "1") != "0"
        self.fuse_add_pos = _os.environ.get("WJ_FUSE_ADD_POS", "1") != "0"     # 0: mapper GEMM + wj_add_pos as two launches
        self.mapper_wgrad_side = _os.environ.get("WJ_MAPPER_WGRAD_SIDE", "1") != "0"
        self.conv_wgrad_side = _os.environ.get("WJ_CONV_WGRAD_SIDE", "1") != "0"   # 0: the sparse conv weight gradients on the main stream
        # LayerNorm forward in its LEAN form on a capped grid (wj_ln_fwd_args.workgroups; csrc/norm.hip) for the stacks named in WJ_LN_LEAN
        # ("tea", "enc", "dec", joined by + or ,; default: none): a kernel of <= 48 VGPRs shares a CU with the persistent GEMM workgroups of
        # the OTHER stream (teacher beside student / predictor).  Measured (round 6, interleaved, one box): 45.79 ms/step without, 47.05 with
        # all three, 45.91 teacher only, 47.63 student + predictor only, 45.94 with a cap of 1024 -- a capped LayerNorm alone on its stream's
        # critical path loses more than it hides (DESIGN_EXPERIMENTS.md); off.  WJ_LN_LEAN_WGS: the cap (256 = one workgroup per CU).
        lean = _os.environ.get("WJ_LN_LEAN", "")
        self.ln_lean = {t.strip() for t in lean.replace("+", ",").split(",") if t.strip() and t.strip() != "0"} if self.use_side else set()
        self.ln_lean_wgs = int(_os.environ.get("WJ_LN_LEAN_WGS", "256"))
        self._lean_now = False                 # set inside the training forward only: inference runs one stream, nothing to stream under
        self._folds = []
        self._bind_params()
        self._bind_wt()
        self._conv_w: Dict[str, torch.Tensor] = {}
        self._alloc_conv_weights()

    def _pick_side_stream(self) -> torch.cuda.Stream:
        """A second stream that really runs beside the current one.  HIP deals streams onto a few hardware queues round-robin;
        two streams on one queue serialise (measured: with a process group active RCCL's streams shift the deal and the
        side stream landed on the main stream's queue -- the whole two-stream overlap was gone).  Probe: one busy-wait
        wave on each stream, started together; concurrent streams take one wait, serialised ones two."""
        main = torch.cuda.current_stream(self.dev)
        ticks = 20000                                    # s_memtime ticks; calibrated below to a ~0.2 ms wait
        ops.spin(100, stream=main.cuda_stream)           # load the code object before timing

        def pair_ms(cand: torch.cuda.Stream) -> float:   # uses the calibrated `ticks`
            e0, e1, go, done = (torch.cuda.Event(enable_timing=True) for _ in range(4))
            best = 1e9
            for _ in range(2):
                go.record(main)
                cand.wait_event(go)
                e0.record(main)
                ops.spin(ticks, stream=main.cuda_stream)
                ops.spin(ticks, stream=cand.cuda_stream)
                done.record(cand)
                main.wait_event(done)
                e1.record(main)
                e1.synchronize()
                best = min(best, e0.elapsed_time(e1))
            return best

        def single_ms() -> float:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(main)
            ops.spin(ticks, stream=main.cuda_stream)
            e1.record(main)
            e1.synchronize()
            return e0.elapsed_time(e1)

        single = single_ms()
        if single < 0.15:                                # faster tick than assumed: stretch the wait to ~0.2 ms
            ticks = int(ticks * 0.2 / max(single, 1e-3))
            single = single_ms()
        cands = [torch.cuda.Stream(device=self.dev) for _ in range(6)] + [torch.cuda.Stream(device=self.dev, priority=-1)]
        self._stream_probe = []
        for cand in cands:
            t = pair_ms(cand)
            self._stream_probe.append(round(t / max(single, 1e-6), 2))
            if t < 1.5 * single:
                return cand
        return cands[0]                                  # nothing overlapped: keep the semantics, lose the overlap

    # ------------------------------------------------------------------------------------------------ parameters
    def _bind_params(self) -> None:
        f, c = self.flat, self.cfg
        self.enc_layers = [_layer_ptrs(f, f"encoder.layers.{i}.", False) for i in range(c.l_enc)]
        self.dec_layers = [_layer_ptrs(f, f"decoder.layers.{i}.", False) for i in range(c.l_dec)]
        self.tea_layers = [_layer_ptrs(f, f"teacher_encoder.layers.{i}.", True) for i in range(c.l_enc)]

    def _bind_wt(self) -> None:
        """W^T shadows: one more bf16 buffer with the parameter layout, every 2-D transformer weight stored transposed at its own offset."""
        f = self.flat
        self._enc_ids = {id(w) for w in self.enc_layers}       # the student's layers (their main-stream launches carry the K-split scratch)
        self.p16t = None
        self._wt_tables = {}                   # (student weights, predictor weights) that take the row form -> (table, n_mats, n_tiles)
        self._wt_need = ((), ())               # ... of the step being run (set by the forward from its row counts)
        if self.wt_dgrad:
            ok = all(r % 64 == 0 and cc % 64 == 0 for w in self.enc_layers + self.dec_layers for k in ("wqkv", "wo", "w1", "w2")
                     for r, cc in [f.by_name[getattr(w, k + "_name")].shape])
            if ok:
                self.p16t = torch.zeros(f.n, dtype=torch.bfloat16, device=self.dev)
                for w in self.enc_layers + self.dec_layers:
                    for k in ("wqkv", "wo", "w1", "w2"):
                        setattr(w, k + "T", self.p16t.data_ptr() + 2 * f.by_name[getattr(w, k + "_name")].offset)
            else:
                self.wt_dgrad = False          # a width that is not a multiple of 64: keep the col-form dgrads
        self._wt_fresh = None                  # the `_wt_need` the shadows were last refreshed for (None: stale)
        self._wt_live = {}

    @staticmethod
    def _row_form_pays(M: int, N: int) -> bool:
        """A dgrad [M, N] takes the row form (W^T shadow) only where it fills the persistent kernel: >= 256 work items of 256 rows x
        256 / 128 columns.  Measured in the step, the student's 117-tile dgrads (M ~ 10 k context rows, N = 768) run 58 / 47 / 20 us in
        col form on the 256 x 128 schedule (234 workgroups) against 64 / 49 / 24 us in row form on the one-tile eight-phase schedule."""
        return -(-M // 256) * (N // 256 + (1 if N % 256 else 0)) >= 256

    PAIR_TILES = (33, 128)      # output tiles of a problem that runs as K-split pairs (csrc/gemm.hip: pair_shape)

    def _pair_pays(self, M: int, N: int, K: int) -> bool:
        """A row-form GEMM [M, N] over K that the library runs as K-split pairs when it is handed scratch (wj_gemm_args.workspace): the
        ragged student's N = 768 products (117 tiles for 256 CUs).  Measured (tools/gemm_small.py, operands not cache-resident):
        K = 3072 74 -> 60 us, K = 2304 58 -> 48 us, against 75 / 57 us for the col-form dgrad; K = 768 loses (27 -> 30 us)."""
        tiles = -(-M // 256) * (N // 256)
        return self.pair_split and N % 256 == 0 and K % 256 == 0 and K >= 1536 and self.PAIR_TILES[0] <= tiles <= self.PAIR_TILES[1]

    def _wt_keys(self, M: int, D: int, pairs: bool = False):
        """The weights of a stack of width D whose dgrad over M rows takes the row form (output widths: w2 -> 4D, the others -> D;
        contraction lengths: wqkv 3D, wo D, w1 4D, w2 D).  pairs: the stack's launches carry the K-split scratch (the student)."""
        if not self.wt_dgrad:
            return ()
        return tuple(k for k, n_out, k_in in (("wqkv", D, 3 * D), ("wo", D, D), ("w1", D, 4 * D), ("w2", 4 * D, D))
                     if self._row_form_pays(M, n_out) or (pairs and self._pair_pays(M, n_out, k_in)))

    def _alloc_conv_weights(self) -> None:
        C = self.C
        for si in range(len(self.stacks)):
            for l, (_, k, s) in enumerate(self.cfg.conv_spec):
                if l == 0:
                    continue
                self._conv_w[f"{si}:wp{l}"] = _empty(C, k * C, dtype=torch.bfloat16, device=self.dev)
                for rho in range(s):
                    U = len(range(rho, k, s))
                    if U > 0:
                        self._conv_w[f"{si}:wd{l}_{rho}"] = _empty(U * C, C, dtype=torch.bfloat16, device=self.dev)
                self._conv_w[f"{si}:dwp{l}"] = torch.zeros(C, k * C, dtype=torch.float32, device=self.dev)

    def _stack_groups(self):
        """[(stack index, first conv clip, clips)] for the conv GEMMs of a batch of self.N clips: one group over all S*N mono
        clips when the streams share their weights, one group of N clips per stream otherwise (clips are channel-major)."""
        if len(self.stacks) == 1:
            return [(0, 0, self.N * self.S)]
        return [(c, c * self.N, self.N) for c in range(self.S)]

    def _fp8_weights(self) -> None:
        """(Re-)quantise the forward weights of the transformer stacks from their bf16 shadows (once per step in fp8 mode)."""
        c = self.cfg
        if not self._w8:
            for layers, d in ((self.enc_layers, c.d_enc), (self.tea_layers, c.d_enc), (self.dec_layers, c.d_dec)):
                for w in layers:
                    for ptr, n, k in ((w.wqkv, 3 * d, d), (w.wo, d, d), (w.w1, 4 * d, d), (w.w2, d, 4 * d)):
                        if k % 256 == 0:
                            self._w8[ptr] = (_empty(n, k, dtype=torch.uint8, device=self.dev),
                                             torch.zeros(ops.fp8_scale_dwords(n, k), dtype=torch.int32, device=self.dev), n, k)
        for ptr, (q, sc, n, k) in self._w8.items():
            ops.quantize_mxfp8(ptr, q, sc, M=n, K=k, ldx=k, ldq=k, ld_scale=n)

    def _linear_fwd(self, stack: str, x, w_ptr: int, out, *, M: int, N: int, K: int, bias, epilogue: int = ops.EPI_BF16, C2=None) -> None:
        """out[M, N] = x[M, K] . W^T (+ bias, epilogue): the bf16 GEMM, or in fp8 mode (eligible K) quantise x and run the MX fp8 GEMM."""
        w8 = self._w8.get(w_ptr) if self.fp8 else None
        if w8 is None:
            ops.gemm(x, w_ptr, out, C2=C2, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=bias, epilogue=epilogue,
                     workspace=self.pair_ws if stack == "enc" else None)       # K-split pairs: main-stream launches only
            return
        q, sc = self._a8[stack]
        ops.quantize_mxfp8(x, q, sc, M=M, K=K, ldx=K, ldq=K, ld_scale=M)
        ops.gemm_mxfp8(q, w8[0], sc, w8[1], out, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, ld_scale_a=M, ld_scale_b=N, epilogue=epilogue,
                       C2=C2, bias=bias)

    def prepare_weights(self, force_cast: bool = False) -> None:
        """bf16 shadow copies (when stale) + the GEMM layouts of conv layers 1.. from the fp32 masters."""
        f = self.flat
        if force_cast or not f.bf16_fresh or self.fp8:
            self.wait_optimizer()
        for mine, src in zip((self.pos_enc, self.pos_dec), self._pos_src):
            if mine.data_ptr() != src.data_ptr():
                mine.copy_(src.reshape(mine.shape))
        if force_cast or not f.bf16_fresh:
            ops.cast_f32_to_bf16(f.p32, f.p16, f.n)
            if f.tn > 0:
                ops.cast_f32_to_bf16(f.t32, f.t16, f.tn)
            f.bf16_fresh = True
        if self.fp8:
            self._fp8_weights()
        self._wt_fresh = None                       # the shadows follow p16; refreshed by the first backward that needs them
        self._conv_w_fresh = False                  # GEMM layouts of conv layers 1..: rebuilt by the front-end, behind its conv0 launches

    def _conv_weight_layouts(self) -> None:
        """GEMM layouts of conv layers 1.. from the fp32 masters (20 launches of ~5 us).  Issued by the front-end AFTER the conv0 kernels
        are queued: at the start of a step the GPU is idle, and the host needs ~15 us per launch -- behind conv0 (1.1 ms) they cost nothing,
        in front of it they were 0.3 ms of idle GPU per step (rocprofv3 trace, tools/trace_gaps.py)."""
        if self._conv_w_fresh:
            return
        f, C = self.flat, self.C
        for si, pre in enumerate(self.stacks):
            for l, (_, k, s) in enumerate(self.cfg.conv_spec):
                if l == 0:
                    continue
                src = f.ptr32(f"{pre}{l}.0.weight")
                ops.conv_weight_layout(src, self._conv_w[f"{si}:wp{l}"], C_out=C, C_in=C, k=k, mode=0)
                for rho in range(s):
                    U = len(range(rho, k, s))
                    if U > 0:
                        ops.conv_weight_layout(src, self._conv_w[f"{si}:wd{l}_{rho}"], C_out=C, C_in=C, k=k, mode=1, stride=s, rho=rho, U=U)
        self._conv_w_fresh = True

    # ------------------------------------------------------------------------------------------------ arena
    def _rows(self, nrows: int, width: int, dtype, lead: int = 2, tail: int = 8) -> Tuple[torch.Tensor, int]:
        """A zero-initialised [lead + nrows + tail][width] buffer; returns (tensor, pointer to row 0)."""
        t = torch.zeros((lead + nrows + tail) * width, dtype=dtype, device=self.dev)
        return t, t.data_ptr() + lead * width * t.element_size()

    def _alloc_stack(self, M: int, D: int, H: int, B: int, layers: int) -> List[_Acts]:
        bf, f32, dev = torch.bfloat16, torch.float32, self.dev
        out = []
        for _ in range(layers):
            a = _Acts()
            a.qkv = _empty(M, 3 * D, dtype=bf, device=dev)
            a.o = _empty(M, D, dtype=bf, device=dev)
            a.lse = _empty(B * H * self.T, dtype=f32, device=dev)
            a.p = _empty(M, D, dtype=bf, device=dev)
            a.m1 = _empty(M, dtype=f32, device=dev)
            a.r1 = _empty(M, dtype=f32, device=dev)
            a.x1 = _empty(M, D, dtype=f32, device=dev)
            a.x1b = _empty(M, D, dtype=bf, device=dev)
            a.h = _empty(M, 4 * D, dtype=bf, device=dev)        # holds gelu'(linear1 output), see EPI_BIAS_GELU2
            a.g = _empty(M, 4 * D, dtype=bf, device=dev)
            a.f = _empty(M, D, dtype=bf, device=dev)
            a.m2 = _empty(M, dtype=f32, device=dev)
            a.r2 = _empty(M, dtype=f32, device=dev)
            a.x2 = _empty(M, D, dtype=f32, device=dev)
            a.x2b = _empty(M, D, dtype=bf, device=dev)
            out.append(a)
        return out

    def alloc(self, N: int, train: bool = True, G: int = 0, need_enc: int = 0, need_dec: int = 0) -> None:
        """(Re)build the arena for N clips (and, for training, G target groups per clip).

        need_enc / need_dec: rows the student / predictor buffers must hold this step (a ragged step: its context / visible rows;
        a dense step: N*T / N*G*T; 0 = dense).  The student / predictor activations, their backward scratch and the prediction
        buffers are sized for that need + WJ_ARENA_MARGIN (default 15 %), not for the dense worst case: with the AudioSet masker a
        ragged step touches 19 % / 42 % of the dense rows, and those buffers were 57 of the 86 GB of a 256-clip arena.  A later step
        that needs more (a batch with more visible tokens, the dense fall-back, WJ_RAGGED=0) grows them -- a re-allocation, rare by
        construction.  WJ_ARENA_DENSE=1 sizes everything for the dense step up front."""
        import os
        G = G or self.G or 1
        full_enc, full_dec = N * self.T, N * G * self.T
        if os.environ.get("WJ_ARENA_DENSE", "0") == "1" or not train:
            need_enc, need_dec = full_enc, full_dec
        need_enc = min(need_enc or full_enc, full_enc)
        need_dec = min(need_dec or full_dec, full_dec)
        same = N == self.N and (not train or (getattr(self, "_train_alloc", False) and G == self.G))
        if same and (not train or (need_enc <= self.cap_enc and need_dec <= self.cap_dec)):
            return
        margin = float(os.environ.get("WJ_ARENA_MARGIN", "1.15"))

        def cap(need: int, full: int, old: int) -> int:
            if need >= full:
                return full
            return min(full, max((int(need * margin) + 255) // 256 * 256, old if same else 0))

        self.cap_enc = cap(need_enc, full_enc, getattr(self, "cap_enc", 0))
        self.cap_dec = cap(need_dec, full_dec, getattr(self, "cap_dec", 0))
        c, bf, f32, dev = self.cfg, torch.bfloat16, torch.float32, self.dev
        T, C = self.T, self.C
        M, Mp = N * T, N * G * T
        self.N, self.G, self.M, self.Mp = N, G, M, Mp
        self._train_alloc = train
        Me, Md = self.cap_enc, self.cap_dec                   # rows of the student / predictor buffers (<= M / Mp)
        nl = len(c.conv_spec)
        Nc = N * self.S                               # mono conv clips (channel-major: clip index c*N + n)
        # conv activations (post-GELU, and pre-GELU for layers >= 1) + their gradients
        self.post, self.post_ptr, self.pre, self.pre_ptr = [], [], [None], [0]
        self.dpost, self.dpost_ptr, self.dpre, self.dpre_ptr = [], [], [None], [0]
        for l in range(nl):
            t, p = self._rows(Nc * self.P[l], C, bf)
            self.post.append(t); self.post_ptr.append(p)
            if l > 0:
                t, p = self._rows(Nc * self.P[l], C, bf)
                self.pre.append(t); self.pre_ptr.append(p)
            if train:
                t, p = self._rows(Nc * self.P[l], C, bf)
                self.dpost.append(t); self.dpost_ptr.append(p)
                if l > 0:
                    t, p = self._rows(Nc * self.P[l], C, bf)
                    self.dpre.append(t); self.dpre_ptr.append(p)
        if self.pair_split and self.pair_ws is None:
            # zero-filled ONCE; the library keeps its flags at zero between launches.  Sized for the largest problem that splits.
            self.pair_ws = torch.zeros(ops.workspace_bytes("wj_gemm_bf16", M=256 * (self.PAIR_TILES[1] // 2), N=512, K=2048, lda=2048, ldb=2048,
                                                           ldc=512, epilogue=ops.EPI_BF16), dtype=torch.uint8, device=dev)
        self.gn_stats = _empty(2, Nc, C, dtype=f32, device=dev)
        taps = c.in_channels * c.conv_spec[0][1]
        conv0_dims = dict(N=N, C_in=c.in_channels, C=C, k=c.conv_spec[0][1], L_out=self.L[0])      # per stream: N clips a call
        self.gn_ws = _empty(ops.workspace_bytes("wj_conv0_gn_gelu_fwd", **conv0_dims) // 4, dtype=f32, device=dev)
        self.gn_ws_b = _empty(ops.workspace_bytes("wj_conv0_gn_gelu_bwd", max_rows=0, **conv0_dims) // 4, dtype=f32, device=dev) if train else None
        self.gn_yx = _empty(Nc, C, taps, dtype=f32, device=dev) if train else None     # forward sums the backward needs
        self.gn_x1 = _empty(Nc, taps, dtype=f32, device=dev) if train else None
        self.fn_b = _empty(M, C, dtype=bf, device=dev)
        self.fn_mean = _empty(M, dtype=f32, device=dev)
        self.fn_rstd = _empty(M, dtype=f32, device=dev)
        self.map_b = _empty(M, c.d_enc, dtype=bf, device=dev)
        self.lf = _empty(M, c.d_enc, dtype=f32, device=dev)
        self.lf_b = _empty(M, c.d_enc, dtype=bf, device=dev)
        # fp8 mode: one activation scratch (e4m3 bytes + block scales) per stack (the teacher runs beside the student)
        self._a8, self._a8s = {}, {}
        # 'enc' is sized for all M rows even when the student's activation buffers follow the ragged row count: infer() runs the
        # student stack densely (M rows) on whatever arena a training step left behind (N == self.N skips alloc), and the quantiser /
        # GELU q_out epilogue write M x 4D bytes into this scratch -- with cap_enc rows that was an out-of-bounds device write
        for tag, (m, d) in dict(tea=(M, c.d_enc), enc=(M, c.d_enc), dec=(Md if train else 0, c.d_dec)).items():
            if m > 0:
                self._a8[tag] = (_empty(m, 4 * d, dtype=torch.uint8, device=dev),
                                 torch.zeros(ops.fp8_scale_dwords(m, 4 * d), dtype=torch.int32, device=dev))
                self._a8s[tag] = (_empty(m, d, dtype=torch.uint8, device=dev),
                                  torch.zeros(ops.fp8_scale_dwords(m, d), dtype=torch.int32, device=dev))
        # scratch stack (teacher / inference): one layer's worth, reused
        self.scratch = self._alloc_stack(M, c.d_enc, c.h_enc, N, 1)[0]
        self.enc_out = _empty(M, c.d_enc, dtype=f32, device=dev)
        self.enc_out_b = _empty(M, c.d_enc, dtype=bf, device=dev)
        self.enc_fm = _empty(M, dtype=f32, device=dev)
        self.enc_fr = _empty(M, dtype=f32, device=dev)
        if not train:
            return
        self.enc_acts = self._alloc_stack(Me, c.d_enc, c.h_enc, N, c.l_enc)
        self.dec_acts = self._alloc_stack(Md, c.d_dec, c.h_dec, N * G, c.l_dec)
        self.ctx_in = _empty(Me, c.d_enc, dtype=bf, device=dev)      # gathered context rows (<= Me)
        self.enc_in = _empty(Me, c.d_enc, dtype=f32, device=dev)     # ragged: local features of the context rows
        self.enc_in_b = _empty(Me, c.d_enc, dtype=bf, device=dev)
        self.tail_o = _empty(Md, c.d_dec, dtype=bf, device=dev)      # last predictor layer: target rows of o / x_in / do
        self.tail_x = _empty(Md, c.d_dec, dtype=f32, device=dev)
        self.tail_do = _empty(Md, c.d_dec, dtype=bf, device=dev)
        self.cf = _empty(Me, c.d_dec, dtype=bf, device=dev)          # contextual_features
        self.dec_in = _empty(Md, c.d_dec, dtype=f32, device=dev)
        self.dec_in_b = _empty(Md, c.d_dec, dtype=bf, device=dev)
        self.dec_out_b = _empty(Md, c.d_dec, dtype=bf, device=dev)
        self.dec_fm = _empty(Md, dtype=f32, device=dev)
        self.dec_fr = _empty(Md, dtype=f32, device=dev)
        self.preds = _empty(Md, c.d_enc, dtype=bf, device=dev)
        self.targets = _empty(M, c.d_enc, dtype=f32, device=dev)
        # teacher: outputs of the last top_k layers (fp32) and their per-clip (sum, sum of squares)
        nkeep = min(c.top_k, c.l_enc) if 1 < c.top_k <= 8 else 0
        self.tea_keep = [_empty(M, c.d_enc, dtype=f32, device=dev) for _ in range(nkeep)]
        self.tea_stats = _empty(max(nkeep, 1), N, ops.GROUP_STATS_SPLIT, 2, dtype=f32, device=dev)   # written by layernorm_fwd
        self.loss = torch.zeros(2, dtype=f32, device=dev)
        self.mse_ws = _empty(ops.workspace_bytes("wj_masked_mse", B=N, G=G, T=T) // 4, dtype=f32, device=dev)
        # backward scratch, one set per stack width
        self.bw = {}
        for tag, (m, d, nbuf, group) in dict(enc=(Me, c.d_enc, 4, 2), dec=(Md, c.d_dec, 2, 1)).items():
            # The weight gradients of `group` consecutive layers go out as ONE grouped launch on the side stream (wj_wgrad_grouped:
            # one small split-K factor for 4-8 problems instead of a large one per problem).  The buffers it reads (dY of every
            # linear) therefore exist 2 * group times (slot = layer % nbuf), so that the main chain may run a whole group ahead.
            self.bw[tag] = dict(
                dy=_empty(m, d, dtype=f32, device=dev), ds=_empty(m, d, dtype=f32, device=dev),
                dx1=_empty(m, d, dtype=f32, device=dev), do=_empty(m, d, dtype=bf, device=dev),
                dgb=_empty(m, d, dtype=bf, device=dev), dxb=_empty(m, d, dtype=bf, device=dev),   # bf16 grad_input of linear1 / in_proj
                dsb2=[_empty(m, d, dtype=bf, device=dev) for _ in range(nbuf)],
                dsb1=[_empty(m, d, dtype=bf, device=dev) for _ in range(nbuf)],
                dh=[_empty(m, 4 * d, dtype=bf, device=dev) for _ in range(nbuf)],
                dqkv=[_empty(m, 3 * d, dtype=bf, device=dev) for _ in range(nbuf)],
                done=[torch.cuda.Event() for _ in range(nbuf)], used=[False] * nbuf, nbuf=nbuf, group=group, pending=[], pending_slots=[])
        # scratch for two-stage parameter-gradient reductions (LayerNorm: [1536][3][D]; attention in_proj bias: [B][3D])
        red_bytes = max(ops.workspace_bytes("wj_layernorm_bwd", D=max(c.d_enc, c.d_dec, C)),
                        ops.workspace_bytes("wj_attn_bwd", B=N * G, H=c.h_dec, hd=c.d_dec // c.h_dec),
                        ops.workspace_bytes("wj_attn_bwd", B=N, H=c.h_enc, hd=c.d_enc // c.h_enc))
        self.red_ws = _empty(red_bytes // 4, dtype=f32, device=dev)
        # deferred folds (WJ_DEFER_FOLDS=0: every LayerNorm / attention backward folds its own partials at once, 75 launches of ~5 us
        # per step): each pending producer keeps its partial rows in its own slot until one grouped launch folds up to 16 of them
        self._red_bytes = (red_bytes + 255) // 256 * 256
        self.fold_ws = _empty(ops.COLSUM_GROUP_MAX * self._red_bytes // 4, dtype=f32, device=dev)
        self._folds = []
        self.dpreds = _empty(Md, c.d_enc, dtype=bf, device=dev)
        self.d_cf = _empty(Me, c.d_dec, dtype=bf, device=dev)
        self.d_ctx_in = _empty(Me, c.d_enc, dtype=bf, device=dev)
        self.d_lf_b = _empty(M, c.d_enc, dtype=bf, device=dev)
        self.d_fn = _empty(M, C, dtype=f32, device=dev)

    # ------------------------------------------------------------------------------------------------ building blocks
    def _layer_fwd(self, w: _Layer, a: _Acts, x_in: torch.Tensor, xb_in: torch.Tensor, M: int, D: int, H: int, B: int,
                   mask: Optional[torch.Tensor], seq: Optional[Tuple[torch.Tensor, int]] = None, save: bool = True,
                   x2_out: Optional[torch.Tensor] = None, x2_stats: Optional[torch.Tensor] = None,
                   sub: Optional[Tuple[torch.Tensor, torch.Tensor, int]] = None, stack: str = "enc", xq_ready: bool = False) -> bool:
        """Post-norm layer: x1 = LN1(x + out_proj(attn(in_proj(x)))); x2 = LN2(x1 + linear2(gelu(linear1(x1)))).
        `seq` = (offsets int32 [B+1], longest sequence) selects the ragged form: M packed rows, no key mask.
        save=False (teacher / inference): nothing is kept for a backward (no gelu' output, no softmax statistics).
        sub = (rows int32 [Ms], inverse int32 [M], Ms): only these rows continue after the attention (the last predictor layer:
        context rows are keys / values there and nothing reads their outputs).
        xq_ready / return value (fp8 mode): the stack's small fp8 buffer already / now holds this layer's input / output."""
        eps = self.cfg.ln_eps
        # fp8 mode with an eligible width (K = D a multiple of 256): the GEMM inputs are produced directly in the MX fp8 operand
        # format by their producers -- LayerNorm (x1 for linear1, x2 for the next layer's in_proj) and linear1's GELU epilogue
        # (for linear2); only the attention output and the very first layer input go through wj_quantize_mxfp8
        f8 = self.fp8 and w.wqkv in self._w8
        if not f8:
            xq_ready = False
            self._linear_fwd(stack, xb_in, w.wqkv, a.qkv, M=M, N=3 * D, K=D, bias=w.bqkv)
        else:
            qb, sb = self._a8[stack]             # [M][4D] bytes: gelu(h)
            qs, ss = self._a8s[stack]            # [M][D] bytes: x2 -> attention output -> x1 -> x2 (one stream: strictly sequential)
            if not xq_ready:
                ops.quantize_mxfp8(xb_in, qs, ss, M=M, K=D, ldx=D, ldq=D, ld_scale=M)
            self._gemm8(qs, ss, w.wqkv, a.qkv, M=M, N=3 * D, K=D, bias=w.bqkv)
        if seq is not None:
            ops.attn_fwd(a.qkv, a.o, B=B, T=seq[1], H=H, hd=D // H, seq_off=seq[0], lse=a.lse if save else None)
        else:
            ops.attn_fwd(a.qkv, a.o, B=B, T=self.T, H=H, hd=D // H, key_mask=mask, lse=a.lse if save else None)
        o_in = a.o
        if sub is not None:
            rows, _, M = sub                                   # M: the rows that go on
            ops.mask_gather_rows(a.o, rows, self.tail_o, n_rows=M, D=D, elem_bytes=2)
            ops.mask_gather_rows(x_in, rows, self.tail_x, n_rows=M, D=D, elem_bytes=4)
            o_in, x_in = self.tail_o, self.tail_x
        f8_out = dict(y_fp8=qs, y_fp8_scales=ss, ld_fp8_scale=M) if f8 else {}
        if f8:
            ops.quantize_mxfp8(o_in, qs, ss, M=M, K=D, ldx=D, ldq=D, ld_scale=M)
            self._gemm8(qs, ss, w.wo, a.p, M=M, N=D, K=D, bias=w.bo)
        else:
            self._linear_fwd(stack, o_in, w.wo, a.p, M=M, N=D, K=D, bias=w.bo)
        lean = self.ln_lean_wgs if (self._lean_now and stack in self.ln_lean and not f8) else 0
        ops.layernorm_fwd(x_in, w.g1, w.be1, M=M, D=D, eps=eps, r=a.p, y_f32=a.x1, y_bf16=a.x1b, mean=a.m1, rstd=a.r1, workgroups=lean, **f8_out)
        if f8:
            gq = dict(q_out=qb, q_scales=sb, ld_q_scale=M)
            if save:
                self._gemm8(qs, ss, w.w1, a.h, M=M, N=4 * D, K=D, bias=w.b1, epilogue=ops.EPI_BIAS_GELU2, C2=a.g, **gq)
            else:                                              # teacher / inference: gelu(h) exists in fp8 only
                self._gemm8(qs, ss, w.w1, None, M=M, N=4 * D, K=D, bias=w.b1, epilogue=ops.EPI_BIAS_GELU, **gq)
            self._gemm8(qb, sb, w.w2, a.f, M=M, N=D, K=4 * D, bias=w.b2)
        else:
            if save:
                self._linear_fwd(stack, a.x1b, w.w1, a.h, M=M, N=4 * D, K=D, bias=w.b1, epilogue=ops.EPI_BIAS_GELU2, C2=a.g)
            else:
                self._linear_fwd(stack, a.x1b, w.w1, a.g, M=M, N=4 * D, K=D, bias=w.b1, epilogue=ops.EPI_BIAS_GELU)
            self._linear_fwd(stack, a.g, w.w2, a.f, M=M, N=D, K=4 * D, bias=w.b2)
        # x2_out / x2_stats (teacher): the layer output goes to its own buffer and its per-clip (sum, sum of squares) is
        # accumulated on the way, so that the targets are ONE pass over the kept layers (wj_instnorm_mean)
        ops.layernorm_fwd(a.x1, w.g2, w.be2, M=M, D=D, eps=eps, r=a.f, y_f32=a.x2 if x2_out is None else x2_out, y_bf16=a.x2b,
                          mean=a.m2, rstd=a.r2, group_stats=x2_stats, group_rows=self.T if x2_stats is not None else 0,
                          workgroups=lean if x2_stats is None else 0, **f8_out)
        return f8 and sub is None              # the small fp8 buffer now holds x2 for the next layer of this stack

    def _gemm8(self, q, sc, w_ptr: int, out, *, M: int, N: int, K: int, bias, epilogue: int = ops.EPI_BF16, C2=None, **extra) -> None:
        w8 = self._w8[w_ptr]
        ops.gemm_mxfp8(q, w8[0], sc, w8[1], out, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, ld_scale_a=M, ld_scale_b=N, epilogue=epilogue, C2=C2,
                       bias=bias, **extra)

    def set_wt_need(self, rows_enc: int, rows_dec: int) -> None:
        """Which weights' dgrads run in row form this step (decided per stack from its row count)."""
        self._wt_need = (self._wt_keys(rows_enc, self.cfg.d_enc, pairs=True), self._wt_keys(rows_dec, self.cfg.d_dec) if self.dec_layers else ())
        self._wt_live = {id(w): self._wt_need[0] for w in self.enc_layers}
        self._wt_live.update({id(w): self._wt_need[1] for w in self.dec_layers})

    def refresh_wt(self) -> None:
        """bf16 W^T shadows from the current bf16 weights: one batched transpose of the weights this step's dgrads read in row form
        (`_wt_need`: with the AudioSet masker at 256 clips all four of every predictor layer and linear2 of every student layer,
        99 of the 213 MB of transformer weights); once per prepared set of weights."""
        need = self._wt_need
        # fresh only for the set of weights it was refreshed for: a second grad-enabled forward on the same prepared weights with other
        # row counts (direct engine use, gradient accumulation) needs other shadows, which would still hold zeros / older weights
        if not self.wt_dgrad or self._wt_fresh == need:
            return
        if need not in self._wt_tables:
            f, rows, tiles = self.flat, [], 0
            for layers, keys in ((self.enc_layers, need[0]), (self.dec_layers, need[1])):
                for w in layers:
                    for k in keys:
                        sl = f.by_name[getattr(w, k + "_name")]
                        r, cc = sl.shape
                        rows.append((sl.offset, r, cc, tiles))
                        tiles += (r // 64) * (cc // 64)
            self._wt_tables[need] = (torch.tensor(rows, dtype=torch.int64, device=self.dev) if rows else None, len(rows), tiles)
        table, n_mats, n_tiles = self._wt_tables[need]
        if n_mats:
            ops.transpose_bf16(self.flat.p16, self.p16t, table, n_mats, n_tiles)
        self._wt_fresh = need

    def _dgrad(self, dY, w: _Layer, key: str, out, *, M: int, N: int, K: int, **kw) -> None:
        """out[M, N] = dY[M, K] . W   (W = the layer's `key` weight, stored [K][N] as nn.Linear keeps it): against the W^T shadow as a
        row-form GEMM, or (WJ_WT_DGRAD=0, widths that are not multiples of 64) against W itself in col form."""
        if self.wt_dgrad and key in self._wt_live.get(id(w), ()):
            ops.gemm(dY, getattr(w, key + "T"), out, M=M, N=N, K=K, lda=K, ldb=K, ldc=N,
                     workspace=self.pair_ws if id(w) in self._enc_ids else None, **kw)
        else:
            ops.gemm(dY, getattr(w, key), out, M=M, N=N, K=K, lda=K, ldb=N, ldc=N, b_trans=1, **kw)

    # ---- parameter-gradient folds of the LayerNorm / attention backward, deferred and grouped
    def _fold_slot(self) -> int:
        if len(self._folds) >= ops.COLSUM_GROUP_MAX:
            self._flush_folds()
        return self.fold_ws.data_ptr() + len(self._folds) * self._red_bytes

    def _flush_folds(self) -> None:
        """One launch adds the column sums of every pending partial matrix to its gradient slices (stream-ordered behind the
        kernels that wrote them; called before a section of the gradient buffer is declared final).  (On the side stream, with two
        scratch buffers: 46.64 against 46.65 ms/step -- stays on the main stream.)"""
        if self._folds:
            ops.colsum_f32_group(self._folds)
            self._folds = []

    def _ln_bwd(self, dy, x, gamma, mean, rstd, *, M: int, D: int, dgamma=None, dbeta=None, dbias=None, **kw) -> None:
        """wj_layernorm_bwd; its dgamma / dbeta / dbias partials are folded later, together with the neighbours' (see _flush_folds)."""
        if self.defer_folds and 3 * D <= 2304 and (dgamma or dbeta or dbias):
            ws = self._fold_slot()
            ops.layernorm_bwd(dy, x, gamma, mean, rstd, M=M, D=D, workspace=ws, **kw)
            self._folds.append((ws, 3 * D, ops.ln_bwd_partial_rows(M, D), 3 * D, dgamma, dbeta, dbias, D))
        else:
            ops.layernorm_bwd(dy, x, gamma, mean, rstd, M=M, D=D, dgamma=dgamma, dbeta=dbeta, dbias=dbias, workspace=self.red_ws, **kw)

    def _attn_bwd(self, qkv, out, dout, lse, dqkv, *, B: int, H: int, hd: int, dbias, **kw) -> None:
        D3 = 3 * H * hd
        if self.defer_folds and D3 <= 2304 and dbias:
            ws = self._fold_slot()
            ops.attn_bwd(qkv, out, dout, lse, dqkv, B=B, H=H, hd=hd, dbias=dbias, dbias_ws=ws, defer_fold=True, **kw)
            self._folds.append((ws, D3, B, D3, dbias, None, None, D3))
        else:
            ops.attn_bwd(qkv, out, dout, lse, dqkv, B=B, H=H, hd=hd, dbias=dbias, dbias_ws=self.red_ws, **kw)

    def _wgrad(self, dY, X, gW, n_out: int, k_in: int, m_tok: int) -> None:
        """gW[n_out, k_in] += dY[m_tok, n_out]^T @ X[m_tok, k_in]   (the three mapper weights: nothing on the main chain reads the result, so
        the launch goes to the side stream -- WJ_MAPPER_WGRAD_SIDE=0: main stream)"""
        def go():
            ops.gemm(dY, X, gW, M=n_out, N=k_in, K=m_tok, lda=n_out, ldb=k_in, ldc=k_in, a_trans=1, b_trans=1,
                     epilogue=ops.EPI_ATOMIC_F32, split_k=ops.pick_split_k(n_out, k_in, m_tok))
        if self.use_side and self.mapper_wgrad_side:
            self._on_side(go)
        else:
            go()

    def _on_side(self, fn) -> None:
        """Run `fn` (kernel launches) on the side stream after everything enqueued so far on the main stream."""
        if not self.use_side:
            fn()
            return
        main = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(main)
        self.side.wait_event(ev)
        with torch.cuda.stream(self.side):
            fn()

    def optimizer_on_side(self, fn) -> None:
        """The parameter update of everything the front-end does not read, on the side stream behind everything the compute stream has
        queued (gradient norm, the front-end parameters' update); the compute stream waits for it in wait_optimizer()."""
        self._opt_main = torch.cuda.current_stream()       # the compute stream the update was issued from
        self._on_side(fn)
        self._opt_ev = torch.cuda.Event()
        self._opt_ev.record(self.side)

    def wait_optimizer(self) -> None:
        """The calling stream waits for a parameter update still in flight on the side stream (no host wait).  Called by the forward in front
        of its first transformer kernel, by inference, the EMA, weight preparation, state_dict, Module._apply and at the end of Trainer.fit.
        The event is kept until the COMPUTE stream (the one the update was issued from) has waited: a first caller on some other stream
        (a state_dict inside a `torch.cuda.stream(...)` block) must not consume it for the next forward."""
        if self._opt_ev is not None:
            cur = torch.cuda.current_stream()
            cur.wait_event(self._opt_ev)
            if getattr(self, "_opt_main", None) is None or cur == self._opt_main:
                self._opt_ev = None

    def _join_side(self) -> None:
        if self.use_side:
            ev = torch.cuda.Event()
            ev.record(self.side)
            torch.cuda.current_stream().wait_event(ev)

    def _layer_bwd(self, w: _Layer, a: _Acts, x_in: torch.Tensor, xb_in: torch.Tensor, dy: torch.Tensor, dyb: Optional[torch.Tensor],
                   dx_out: torch.Tensor, M: int, D: int, H: int, B: int, mask: Optional[torch.Tensor], bw: dict, parity: int,
                   seq: Optional[Tuple[torch.Tensor, int]] = None,
                   sub: Optional[Tuple[torch.Tensor, torch.Tensor, int]] = None, flush: bool = True, bottom: bool = False):
        """d(x2) = dy (fp32) + dyb (bf16 or None) -> d(x_in), returned as (fp32 part, bf16 part or None, flushed); parameter gradients
        accumulated into the flat gradient buffer.
        The grad_input of linear1 and in_proj is what a bf16 linear returns under autocast -- a bf16 tensor (`dgb` / `dxb`) -- and the
        LayerNorm backward that consumes the sum adds it to the fp32 residual gradient on its way in (`dy2`): 2 + 2 bytes per element
        instead of the 4 + 4 of a read-modify-write fp32 GEMM epilogue.  Only the `bottom` layer of a stack folds the two into one
        fp32 tensor (dx_out), for the consumers below the stack.
        The four weight-gradient GEMMs only need (dY, X) and nothing downstream needs them before the optimiser: they are queued
        and go out as one grouped launch on the side stream (with `flush`; `parity` = this layer's buffer slot) while the main
        stream continues the dgrad chain.  Returns True when the queue was flushed (the queued layers' gradients are then final in
        side-stream order)."""
        ds, dx1, do = bw["ds"], bw["dx1"], bw["do"]
        dsb2, dsb1, dh, dqkv = bw["dsb2"][parity], bw["dsb1"][parity], bw["dh"][parity], bw["dqkv"][parity]
        if self.use_side and bw["used"][parity]:
            torch.cuda.current_stream().wait_event(bw["done"][parity])   # side stream finished reading this parity's buffers
        Mall, x_ln1, o_in = M, x_in, a.o
        if sub is not None:              # dy holds the sub-rows only; everything up to the attention works on them
            M = sub[2]
            x_ln1, o_in = self.tail_x, self.tail_o
        dgb, dxb = bw["dgb"], bw["dxb"]
        self._ln_bwd(dy, a.x1, w.g2, a.m2, a.r2, M=M, D=D, r=a.f, dy2=dyb, dy2_is_bf16=dyb is not None, ds_f32=ds, ds_bf16=dsb2,
                     dgamma=w.gg2, dbeta=w.gbe2, dbias=w.gb2)
        self._dgrad(dsb2, w, "w2", dh, M=M, N=4 * D, K=D, epilogue=ops.EPI_MUL_GELU_GRAD, aux=a.h,
                    colsum=w.gb1)        # linear1.bias gradient = column sums of dh, fused into the producing epilogue

        bw["pending"] += [(dsb2, a.g, w.gw2, D, 4 * D, M), (dh, a.x1b, w.gw1, 4 * D, D, M)]
        self._dgrad(dh, w, "w1", dgb, M=M, N=D, K=4 * D)
        self._ln_bwd(ds, x_ln1, w.g1, a.m1, a.r1, M=M, D=D, r=a.p, dy2=dgb, dy2_is_bf16=True, ds_f32=ds, ds_bf16=dsb1,
                     dgamma=w.gg1, dbeta=w.gbe1, dbias=w.gbo)      # ds updated in place
        ds_all = ds
        if sub is None:
            self._dgrad(dsb1, w, "wo", do, M=M, N=D, K=D)
        else:                             # back to all rows: zero gradient where no output was used
            self._dgrad(dsb1, w, "wo", self.tail_do, M=M, N=D, K=D)
            ops.unmask_rows_f32(self.tail_do, sub[1], do, M=Mall, D=D, dst_is_bf16=True)
            ops.unmask_rows_f32(ds, sub[1], dx1, M=Mall, D=D, src_is_f32=True)       # dx1 is free again: residual gradient
            ds_all = dx1
        if seq is not None:
            self._attn_bwd(a.qkv, a.o, do, a.lse, dqkv, B=B, T=seq[1], H=H, hd=D // H, seq_off=seq[0], dbias=w.gbqkv)
        else:
            self._attn_bwd(a.qkv, a.o, do, a.lse, dqkv, B=B, T=self.T, H=H, hd=D // H, key_mask=mask, dbias=w.gbqkv)

        bw["pending"] += [(dsb1, o_in, w.gwo, D, D, M), (dqkv, xb_in, w.gwqkv, 3 * D, D, Mall)]
        bw["pending_slots"].append(parity)
        if flush:
            probs, slots = bw["pending"], bw["pending_slots"]
            bw["pending"], bw["pending_slots"] = [], []

            def wgrads():
                for i in range(0, len(probs), 8):
                    ops.wgrad_grouped(probs[i:i + 8])
                if self.use_side:
                    for sl in slots:
                        bw["done"][sl].record(self.side)
                        bw["used"][sl] = True
            self._on_side(wgrads)
        if bottom:
            self._dgrad(dqkv, w, "wqkv", dx_out, M=Mall, N=D, K=3 * D, epilogue=ops.EPI_ADD_F32, aux=ds_all)
            return dx_out, None, flush
        self._dgrad(dqkv, w, "wqkv", dxb, M=Mall, N=D, K=3 * D)
        return ds_all, dxb, flush

    # ------------------------------------------------------------------------------------------------ front-end
    def _frontend(self, audio: torch.Tensor) -> None:
        """audio bf16 [N, C_in, L] -> lf (fp32) / lf_b (bf16) [N*T, d_enc]   (reference jepa.py:391-396)"""
        c, f, N, C, S = self.cfg, self.flat, self.N, self.C, self.S
        _, k0, s0 = c.conv_spec[0]
        taps = c.in_channels * k0
        grad = torch.is_grad_enabled() and self.gn_yx is not None      # (an inference arena keeps no backward sums)
        audio_p = audio.data_ptr()
        for ch in range(S):                               # every stream: N mono clips (ConvFeatureExtractor: one stream, C_in channels)
            pre = self.stacks[min(ch, len(self.stacks) - 1)]
            c0 = ch * N                                   # first conv clip of this stream
            ops.conv0_fwd(audio_p + ch * c.n_samples * 2 if S > 1 else audio, f.ptr16(f"{pre}0.0.weight"), f.ptr32(f"{pre}0.2.weight"),
                          f.ptr32(f"{pre}0.2.bias"), self.post_ptr[0] + c0 * self.P[0] * C * 2, self.gn_stats[0, c0:], self.gn_stats[1, c0:],
                          self.gn_ws, N=N, C_in=c.in_channels, L=c.n_samples, C=C, k=k0, stride=s0, L_out=self.L[0], P=self.P[0],
                          yx=self.gn_yx[c0:] if grad else None, x1=self.gn_x1[c0:] if grad else None,
                          audio_clip_stride=S * c.in_channels * c.n_samples if S > 1 else 0)
        self._conv_weight_layouts()

        def conv_layer(l: int, si: int, c0: int, nclips: int) -> None:
            _, k, s = c.conv_spec[l]
            ops.gemm(self.post_ptr[l - 1] + c0 * self.P[l - 1] * C * 2, self._conv_w[f"{si}:wp{l}"], self.pre_ptr[l] + c0 * self.P[l] * C * 2,
                     C2=self.post_ptr[l] + c0 * self.P[l] * C * 2, M=nclips * self.P[l], N=C, K=k * C, lda=s * C, ldb=k * C, ldc=C,
                     epilogue=ops.EPI_CONV_GELU, seg_rows=self.P[l], seg_valid=self.L[l])

        for l in range(1, len(c.conv_spec)):
            for si, c0, nclips in self._stack_groups():
                conv_layer(l, si, c0, nclips)
        M, T = self.M, self.T
        ops.layernorm_fwd(self.post_ptr[-1], f.ptr32("feature_norms.weight"), f.ptr32("feature_norms.bias"), M=M, D=C,
                          eps=c.norm_eps, y_bf16=self.fn_b, mean=self.fn_mean, rstd=self.fn_rstd, x_is_bf16=True,
                          in_seg=self.P[-1], in_valid=self.Tc, in_chan=S if S > 1 else 0)
        if self.has_mapper and self.fuse_add_pos and c.d_enc % 256 == 0 and C % 128 == 0:
            # mapper + positions in one launch (SURVEY K8 + K9): the position add rides in the GEMM's epilogue
            ops.gemm(self.fn_b, f.ptr16("post_extraction_mapper.weight"), self.lf_b, C2=self.lf, M=M, N=c.d_enc, K=C, lda=C, ldb=C,
                     ldc=c.d_enc, bias=f.ptr32("post_extraction_mapper.bias"), epilogue=ops.EPI_BF16_ADD_POS, aux=self.pos_enc, seg_rows=T)
            return
        if self.has_mapper:
            ops.gemm(self.fn_b, f.ptr16("post_extraction_mapper.weight"), self.map_b, M=M, N=c.d_enc, K=C, lda=C, ldb=C,
                     ldc=c.d_enc, bias=f.ptr32("post_extraction_mapper.bias"))
            src = self.map_b
        else:
            src = self.fn_b
        ops.add_pos(src, self.pos_enc, M=M, T=T, D=c.d_enc, y_f32=self.lf, y_bf16=self.lf_b)

    # ------------------------------------------------------------------------------------------------ forward
    def forward(self, audio: torch.Tensor, plan: MaskPlan) -> None:
        """Training forward.  Results: self.loss[0], self.preds, self.targets, self.cf[:n_ctx], self.lf."""
        c, f = self.cfg, self.flat
        N = audio.shape[0]
        if plan.N != N or plan.T != self.T or plan.G < 1:
            raise ValueError(f"mask plan is for {plan.N} clips x {plan.G} groups x {plan.T} tokens; the batch has {N} clips of {self.T} tokens")
        rag = self.ragged and plan.ragged_ok
        self.alloc(N, train=True, G=plan.G, need_enc=plan.n_ctx if rag else 0, need_dec=plan.n_dec if rag else 0)
        self.plan = plan
        self.audio = audio
        M, Mp, T, G = self.M, self.Mp, self.T, self.G
        De, Dd = c.d_enc, c.d_dec
        self._lean_now = self.use_side
        try:
            self._forward_body(audio, plan)
        finally:
            self._lean_now = False

    def _forward_body(self, audio: torch.Tensor, plan: MaskPlan) -> None:
        c, f = self.cfg, self.flat
        N = audio.shape[0]
        M, Mp, T, G = self.M, self.Mp, self.T, self.G
        De, Dd = c.d_enc, c.d_dec
        self._frontend(audio)
        # EMA teacher on the same local features (no mask, no final norm), joint instance-norm, mean of the last k layers:
        # independent of the student / predictor chain below, so it runs beside it on the side stream
        rag_now = self.ragged and plan.ragged_ok
        self.set_wt_need(plan.n_ctx if rag_now else M, plan.n_dec if rag_now else Mp)

        def beside():
            self._teacher_targets()
            if torch.is_grad_enabled():
                self.refresh_wt()       # W^T shadows for the backward's row-form dgrads: off the forward's critical path
        self._on_side(beside)
        self.wait_optimizer()           # (an update overlapped with this step's front-end: the transformer stacks' parameters)
        self.ragged_step = self.ragged and plan.ragged_ok
        if self.ragged_step and self.sparse_conv and torch.is_grad_enabled():
            # host-side list building + upload, hidden behind the forward kernels already queued.  (Building the lists later, behind
            # the student / predictor launches, measured the same outside the profiler -- 48.89 against 48.79 ms over three runs each --
            # and opened a 2-ms gap under rocprofv3, whose per-launch overhead makes the host the slower side.)
            self._conv_rows(plan)
        n_ctx = plan.n_ctx
        if self.ragged_step:
            # student encoder on the context rows only, packed per clip (non-context rows are dropped at jepa.py:399 and,
            # being key-masked, never influence a context row)
            Me, eseq = n_ctx, (plan.enc_off, max(plan.max_enc, 1))
            ops.mask_gather_rows(self.lf, plan.keep, self.enc_in, n_rows=n_ctx, D=De, elem_bytes=4)
            ops.mask_gather_rows(self.lf_b, plan.keep, self.enc_in_b, n_rows=n_ctx, D=De, elem_bytes=2)
            x, xb = self.enc_in, self.enc_in_b
            xq = False
            for w, a in zip(self.enc_layers, self.enc_acts):
                xq = self._layer_fwd(w, a, x, xb, Me, De, c.h_enc, N, None, eseq, xq_ready=xq)
                x, xb = a.x2, a.x2b
            ops.layernorm_fwd(x, f.ptr32("encoder.norm.weight"), f.ptr32("encoder.norm.bias"), M=Me, D=De, eps=c.norm_eps,
                              y_bf16=self.ctx_in, mean=self.enc_fm, rstd=self.enc_fr,
                              workgroups=self.ln_lean_wgs if (self._lean_now and "enc" in self.ln_lean) else 0)
        else:
            # student encoder over every token (keys restricted to the context), then the boolean-mask gather
            x, xb = self.lf, self.lf_b
            xq = False
            for w, a in zip(self.enc_layers, self.enc_acts):
                xq = self._layer_fwd(w, a, x, xb, M, De, c.h_enc, N, plan.ctx_u8, xq_ready=xq)
                x, xb = a.x2, a.x2b
            ops.layernorm_fwd(x, f.ptr32("encoder.norm.weight"), f.ptr32("encoder.norm.bias"), M=M, D=De, eps=c.norm_eps,
                              y_f32=self.enc_out, y_bf16=self.enc_out_b, mean=self.enc_fm, rstd=self.enc_fr)
            ops.mask_gather_rows(self.enc_out_b, plan.keep, self.ctx_in, n_rows=n_ctx, D=De, elem_bytes=2)
        ops.gemm(self.ctx_in, f.ptr16("encoder_to_decoder_mapper.weight"), self.cf, M=n_ctx, N=Dd, K=De, lda=De, ldb=De, ldc=Dd,
                 bias=f.ptr32("encoder_to_decoder_mapper.bias"))
        # predictor over (context U group targets), one sequence per (clip, group)
        if self.ragged_step:
            Md, dseq = plan.n_dec, (plan.dec_off, max(plan.max_dec, 1))
            ops.mask_scatter_fill_pos(self.cf, plan.inv, f.ptr32("mask_token"), self.pos_dec, B=N, T=T, D=Dd, G=G,
                                      out_f32=self.dec_in, out_bf16=self.dec_in_b, rows=plan.dec_rows, n_rows=Md)
        else:
            Md, dseq = Mp, None
            ops.mask_scatter_fill_pos(self.cf, plan.inv, f.ptr32("mask_token"), self.pos_dec, B=N, T=T, D=Dd, G=G,
                                      out_f32=self.dec_in, out_bf16=self.dec_in_b)
        x, xb = self.dec_in, self.dec_in_b
        self.tail = (plan.tgt_rows, plan.tgt_inv, plan.n_tgt) if (self.ragged_step and self.trim_tail and plan.n_tgt > 0) else None
        Mo = Md                          # rows that leave the predictor
        xq = False
        for i, (w, a) in enumerate(zip(self.dec_layers, self.dec_acts)):
            last = i == c.l_dec - 1
            xq = self._layer_fwd(w, a, x, xb, Md, Dd, c.h_dec, N * G, plan.vis_u8, dseq, sub=self.tail if last else None, stack="dec",
                                 xq_ready=xq)
            x, xb = a.x2, a.x2b
        if self.tail is not None:
            Mo = plan.n_tgt
        ops.layernorm_fwd(x, f.ptr32("decoder.norm.weight"), f.ptr32("decoder.norm.bias"), M=Mo, D=Dd, eps=c.norm_eps,
                          y_bf16=self.dec_out_b, mean=self.dec_fm, rstd=self.dec_fr,
                          workgroups=self.ln_lean_wgs if (self._lean_now and "dec" in self.ln_lean) else 0)
        ops.gemm(self.dec_out_b, f.ptr16("decoder_to_encoder_mapper.weight"), self.preds, M=Mo, N=De, K=Dd, lda=Dd, ldb=Dd,
                 ldc=De, bias=f.ptr32("decoder_to_encoder_mapper.bias"))
        self._join_side()               # teacher targets (side stream) are needed by the loss
        self._mse(None, None)

    def _mse(self, dpreds, gscale_ptr) -> None:
        c, plan = self.cfg, self.plan
        if self.ragged_step and self.tail is not None:
            rows = dict(rows=plan.tgt_dense, n_rows=plan.n_tgt)          # preds hold the target rows only
        else:
            rows = dict(rows=plan.dec_rows, n_rows=plan.n_dec) if self.ragged_step else {}
        ops.masked_mse(self.preds, self.targets, plan.tgt_u8, self.loss, self.mse_ws, B=self.N, G=self.G, T=self.T, D=c.d_enc,
                       dpreds=dpreds, gscale_ptr=gscale_ptr, **rows)

    def dense_preds(self) -> torch.Tensor:
        """Predictions as the reference shapes them, bf16 [N*G, T, d_enc].  On a ragged step only the visible rows were
        computed (with the trimmed last layer: only the target rows); the others (zero loss weight on the reference,
        jepa.py:356) read 0."""
        N, G, T, De = self.N, self.G, self.T, self.cfg.d_enc
        if not self.ragged_step:
            return self.preds.view(N * G, T, De)
        out = _empty(N * G * T, De, dtype=torch.bfloat16, device=self.dev)
        if self.tail is not None:        # preds hold the target rows only: dense position -> packed row -> target row
            inv = torch.full((N * G * T,), -1, dtype=torch.int32, device=self.dev)
            inv[self.plan.tgt_dense.long()] = torch.arange(self.plan.n_tgt, dtype=torch.int32, device=self.dev)
        else:
            inv = self.plan.dec_map
        ops.unmask_rows_f32(self.preds, inv, out, M=N * G * T, D=De, src_is_f32=False, dst_is_bf16=True)
        return out.view(N * G, T, De)

    def _teacher_targets(self) -> None:
        c, N, M, De = self.cfg, self.N, self.M, self.cfg.d_enc
        a = self.scratch
        x, xb = self.lf, self.lf_b
        fused = 1 < c.top_k <= 8
        kept = 0
        xq = False
        for i, w in enumerate(self.tea_layers):
            keep = c.l_enc - i <= c.top_k
            if keep and fused:
                xq = self._layer_fwd(w, a, x, xb, M, De, c.h_enc, N, None, save=False, x2_out=self.tea_keep[kept],
                                     x2_stats=self.tea_stats[kept], stack="tea", xq_ready=xq)
                x, xb = self.tea_keep[kept], a.x2b
            else:
                xq = self._layer_fwd(w, a, x, xb, M, De, c.h_enc, N, None, save=False, stack="tea", xq_ready=xq)
                # ping-pong: the next layer reads x2/x2b while writing x1.. of the same scratch set, then x2 again;
                # x2 is only overwritten by the LAST kernel of the layer, after its readers have run (stream order).
                x, xb = a.x2, a.x2b
                if keep and c.top_k > 1:   # mean over the layers actually kept: min(top_k, layers) (reference jepa.py:249-252)
                    ops.instnorm_accumulate(x, self.targets, B=N, TD=self.T * De, accumulate=kept > 0, scale=1.0 / min(c.top_k, c.l_enc))
            kept += int(keep)
        if fused:
            ops.instnorm_mean(self.tea_keep[:kept], self.tea_stats, self.targets, B=N, TD=self.T * De)
        elif c.top_k <= 1:
            self.targets.copy_(x)

    # ------------------------------------------------------------------------------------------------ backward
    def backward(self, gscale_ptr: int = 0, on_grads_ready=None) -> None:
        """Gradients of self.loss[0] w.r.t. every trainable parameter -> flat.g32 (overwritten).

        `on_grads_ready(tag)` is called (host side, stream-ordered) as sections of the flat gradient buffer become
        final: "dec" (predictor + both mappers), "enc:<i>" after encoder layer i, "front" at the end -- the hook the
        data-parallel wrapper uses to launch bucketed RCCL all-reduces that overlap the rest of the backward."""
        if on_grads_ready is None:
            ready = lambda tag: None
        elif self.use_side:
            # a section is final once BOTH streams are past this point: issue the hook (the bucket's all-reduce) from the
            # side stream after it has waited for the main stream, so that the main dgrad chain never stalls on it
            ready = lambda tag: self._on_side(lambda: on_grads_ready(tag))
        else:
            ready = on_grads_ready
        section_final = ready

        def ready(tag):                  # a section is final only once the pending parameter-gradient folds have been queued
            if on_grads_ready is not None:       # (no consumer of sections: the folds go out in full groups and at the end)
                self._flush_folds()
            section_final(tag)
        self._folds = []                 # entries an exception in an earlier backward may have left behind must not be folded into this one
        c, f, plan = self.cfg, self.flat, self.plan
        N, M, Mp, T, G, C = self.N, self.M, self.Mp, self.T, self.G, self.C
        De, Dd = c.d_enc, c.d_dec
        if not getattr(f, "g_clean", False):      # (FusedAdamW.fuse_zero_grad: the last update left the buffer clear)
            f.g32.zero_()
        f.g_clean = False
        self.refresh_wt()               # (already done beside the forward; a no-op then)
        rag = self.ragged_step
        Md, dseq = (plan.n_dec, (plan.dec_off, max(plan.max_dec, 1))) if rag else (Mp, None)
        Me, eseq = (plan.n_ctx, (plan.enc_off, max(plan.max_enc, 1))) if rag else (M, None)
        self._mse(self.dpreds, gscale_ptr if gscale_ptr else None)
        bw = self.bw["dec"]
        # decoder_to_encoder_mapper
        Mo = plan.n_tgt if (rag and self.tail is not None) else Md       # rows that left the predictor
        ops.colsum_bf16(self.dpreds, f.gptr("decoder_to_encoder_mapper.bias"), M=Mo, N=De, ldx=De)
        self._wgrad(self.dpreds, self.dec_out_b, f.gptr("decoder_to_encoder_mapper.weight"), De, Dd, Mo)
        ops.gemm(self.dpreds, f.ptr16("decoder_to_encoder_mapper.weight"), bw["dx1"], M=Mo, N=Dd, K=De, lda=De, ldb=Dd, ldc=Dd,
                 b_trans=1, epilogue=ops.EPI_ADD_F32)
        last = self.dec_acts[-1]
        ops.layernorm_bwd(bw["dx1"], last.x2, f.ptr32("decoder.norm.weight"), self.dec_fm, self.dec_fr, M=Mo, D=Dd, ds_f32=bw["dy"],
                          dgamma=f.gptr("decoder.norm.weight"), dbeta=f.gptr("decoder.norm.bias"), workspace=self.red_ws)
        dy, dyb = bw["dy"], None
        for i in range(c.l_dec - 1, -1, -1):
            x_in, xb_in = (self.dec_in, self.dec_in_b) if i == 0 else (self.dec_acts[i - 1].x2, self.dec_acts[i - 1].x2b)
            dy, dyb, _ = self._layer_bwd(self.dec_layers[i], self.dec_acts[i], x_in, xb_in, dy, dyb, bw["dy"], Md, Dd, c.h_dec, N * G, plan.vis_u8,
                                         bw, i % bw["nbuf"], dseq, sub=self.tail if (rag and i == c.l_dec - 1) else None, bottom=i == 0)
        n_ctx = plan.n_ctx
        # the mask-token gradient leaves the kernel as one partial row per workgroup, folded with the other deferred folds (round 4:
        # 384 global float atomics per workgroup into the same 384 addresses)
        sf_rows = ops.scatter_fill_bwd_partial_rows(N, T)
        if self.defer_folds and Dd <= 2304 and sf_rows * Dd * 4 <= self._red_bytes:
            ws = self._fold_slot()
            ops.mask_scatter_fill_pos_bwd(dy, plan.inv, self.d_cf, f.gptr("mask_token"), B=N, T=T, D=Dd, G=G,
                                          rowmap=plan.dec_map if rag else None, partials=ws)
            self._folds.append((ws, Dd, sf_rows, Dd, f.gptr("mask_token"), None, None, Dd))
        else:
            ops.mask_scatter_fill_pos_bwd(dy, plan.inv, self.d_cf, f.gptr("mask_token"), B=N, T=T, D=Dd, G=G,
                                          rowmap=plan.dec_map if rag else None)
        # encoder_to_decoder_mapper (rows = gathered context tokens)
        ops.colsum_bf16(self.d_cf, f.gptr("encoder_to_decoder_mapper.bias"), M=n_ctx, N=Dd, ldx=Dd)
        self._wgrad(self.d_cf, self.ctx_in, f.gptr("encoder_to_decoder_mapper.weight"), Dd, De, n_ctx)
        ops.gemm(self.d_cf, f.ptr16("encoder_to_decoder_mapper.weight"), self.d_ctx_in, M=n_ctx, N=De, K=Dd, lda=Dd, ldb=De,
                 ldc=De, b_trans=1)
        ready("dec")
        bw = self.bw["enc"]
        if rag:
            ops.unmask_rows_f32(self.d_ctx_in, None, bw["dx1"], M=Me, D=De)      # packed rows: a widening copy
        else:
            ops.unmask_rows_f32(self.d_ctx_in, plan.inv, bw["dx1"], M=M, D=De)
        last = self.enc_acts[-1]
        ops.layernorm_bwd(bw["dx1"], last.x2, f.ptr32("encoder.norm.weight"), self.enc_fm, self.enc_fr, M=Me, D=De, ds_f32=bw["dy"],
                          dgamma=f.gptr("encoder.norm.weight"), dbeta=f.gptr("encoder.norm.bias"), workspace=self.red_ws)
        dy, dyb = bw["dy"], None
        enc_ready = set()
        for i in range(c.l_enc - 1, -1, -1):
            first = (self.enc_in, self.enc_in_b) if rag else (self.lf, self.lf_b)
            x_in, xb_in = first if i == 0 else (self.enc_acts[i - 1].x2, self.enc_acts[i - 1].x2b)
            # the weight gradients of two layers share a grouped launch: a layer's section of the gradient buffer is final (and its
            # all-reduce bucket may go) once the launch that carries it has been queued
            dy, dyb, done = self._layer_bwd(self.enc_layers[i], self.enc_acts[i], x_in, xb_in, dy, dyb, bw["dy"], Me, De, c.h_enc, N, plan.ctx_u8,
                                            bw, i % bw["nbuf"], eseq, flush=(c.l_enc - 1 - i) % bw["group"] == bw["group"] - 1 or i == 0,
                                            bottom=i == 0)
            if done:
                for j in range(min(c.l_enc - 1, i + bw["group"] - 1), i - 1, -1):
                    if j not in enc_ready:
                        enc_ready.add(j)
                        ready(f"enc:{j}")
        self._frontend_bwd(dy, rag, plan)
        self._flush_folds()
        self._join_side()                # all weight gradients are final before the optimiser / last all-reduce
        for tag in ("enc", "dec"):
            self.bw[tag]["used"] = [False] * self.bw[tag]["nbuf"]
        if on_grads_ready is not None:
            on_grads_ready("front")

    def _frontend_bwd(self, dy: torch.Tensor, rag: bool, plan: Optional[MaskPlan]) -> None:
        """dy = d(local_features) fp32 [M, d_enc] (packed context rows on a ragged step) -> gradients of the mapper, feature_norms
        and the conv stack.  (JEPA: zero on non-context rows; the teacher branch is detached, jepa.py:408.)"""
        c, f = self.cfg, self.flat
        N, M, C, De = self.N, self.M, self.C, c.d_enc
        if rag:
            ops.unmask_rows_f32(dy, plan.inv, self.d_lf_b, M=M, D=De, src_is_f32=True, dst_is_bf16=True)
            if not self.has_mapper:
                ops.unmask_rows_f32(dy, plan.inv, self.d_fn, M=M, D=De, src_is_f32=True)
                dy = self.d_fn
        else:
            ops.cast_f32_to_bf16(dy, self.d_lf_b, M * De)
        if self.has_mapper:
            ops.colsum_bf16(self.d_lf_b, f.gptr("post_extraction_mapper.bias"), M=M, N=De, ldx=De)
            self._wgrad(self.d_lf_b, self.fn_b, f.gptr("post_extraction_mapper.weight"), De, C, M)
            ops.gemm(self.d_lf_b, f.ptr16("post_extraction_mapper.weight"), self.d_fn, M=M, N=C, K=De, lda=De, ldb=C, ldc=C,
                     b_trans=1, epilogue=ops.EPI_ADD_F32)
            d_fn = self.d_fn
        else:
            d_fn = dy
        nl = len(c.conv_spec)
        S, Tc = self.S, self.Tc
        ops.layernorm_bwd(d_fn, self.post_ptr[-1], f.ptr32("feature_norms.weight"), self.fn_mean, self.fn_rstd, M=M, D=C,
                          ds_bf16=self.dpost_ptr[-1], dgamma=f.gptr("feature_norms.weight"), dbeta=f.gptr("feature_norms.bias"),
                          x_is_bf16=True, in_seg=self.P[-1], in_valid=Tc, out_seg=self.P[-1], out_valid=Tc, chan=S if S > 1 else 0)
        sparse = rag and self.sparse_conv
        if sparse:
            act_rows = self._conv_rows(plan)
            if self._conv_grads_dirty:       # a dense step left gradients everywhere: restore the all-zero state once
                for t in self.dpost + self.dpre[1:]:
                    t.zero_()
                self._conv_grads_dirty = False
        else:
            self._conv_grads_dirty = True
        groups = self._stack_groups()        # (stack, first conv clip, clips): one group, or one per channel stream
        side_wgrad = sparse and self.use_side and self.conv_wgrad_side
        late_clear = []
        for l in range(nl - 1, 0, -1):
            _, k, s = c.conv_spec[l]
            empty_phase = any(len(range(rho, k, s)) == 0 for rho in range(s))
            if empty_phase:
                self.dpost[l - 1].zero_()
            for gi, (si, c0, nclips) in enumerate(groups):
                rows = nclips * self.P[l]
                r0, r0p = c0 * self.P[l] * C * 2, c0 * self.P[l - 1] * C * 2      # byte offsets of the group's first row (layers l, l-1)
                dwp = self._conv_w[f"{si}:dwp{l}"]
                if not side_wgrad:
                    dwp.zero_()
                if sparse:
                    # Only act[l] rows of this layer's output gradient are non-zero.  Every gradient buffer is all-zero outside
                    # the rows written this step (they are cleared again below), so the dgrad taps may read neighbours freely.
                    # The lists hold rows of the WHOLE buffer; a group takes its contiguous slice of them.
                    act, n_act, ext, n_ext = act_rows[l][gi]
                    if not (self.fuse_conv_gelu_bwd and l < nl - 1):
                        # (layers below the top one: d(pre) was written by the dgrad of the layer above, GELU' fused in its epilogue)
                        ops.gelu_bwd_bf16(self.dpost_ptr[l], self.pre_ptr[l], self.dpre_ptr[l], 0, rows=act, n_rows=n_act, row_elems=C,
                                          clear_dpost=l < nl - 1)
                    if n_act > 0 and side_wgrad:
                        # The layer's weight gradient (+ its scratch clear and the layout fold) on the SIDE stream: at this point of the
                        # backward that stream is idle (every transformer weight gradient is out), and the main chain goes on with this
                        # layer's dgrads, GELU' and the layer-0 pass -- d(pre[l]) is read by both and cleared only behind the join below.
                        def conv_wgrad(l=l, k=k, s=s, dwp=dwp, act=act, n_act=n_act, si=si):
                            dwp.zero_()
                            ops.gemm(self.dpre_ptr[l], self.post_ptr[l - 1], dwp, M=C, N=k * C, K=n_act, lda=C, ldb=s * C, ldc=k * C, a_trans=1,
                                     b_trans=1, epilogue=ops.EPI_ATOMIC_F32, split_k=ops.pick_split_k(C, k * C, n_act), rowmap=act)
                            ops.conv_weight_layout(dwp, f.gptr(f"{self.stacks[si]}{l}.0.weight"), C_out=C, C_in=C, k=k, mode=2)
                        self._on_side(conv_wgrad)
                    elif n_act > 0:
                        ops.gemm(self.dpre_ptr[l], self.post_ptr[l - 1], dwp, M=C, N=k * C, K=n_act, lda=C, ldb=s * C, ldc=k * C, a_trans=1,
                                 b_trans=1, epilogue=ops.EPI_ATOMIC_F32, split_k=ops.pick_split_k(C, k * C, n_act), rowmap=act)
                else:
                    ops.gelu_bwd_bf16(self.dpost_ptr[l] + r0, self.pre_ptr[l] + r0, self.dpre_ptr[l] + r0, rows * C)
                    ops.gemm(self.dpre_ptr[l] + r0, self.post_ptr[l - 1] + r0p, dwp, M=C, N=k * C, K=rows, lda=C, ldb=s * C, ldc=k * C,
                             a_trans=1, b_trans=1, epilogue=ops.EPI_ATOMIC_F32, split_k=ops.pick_split_k(C, k * C, rows))
                if not side_wgrad:
                    ops.conv_weight_layout(dwp, f.gptr(f"{self.stacks[si]}{l}.0.weight"), C_out=C, C_in=C, k=k, mode=2)
                for rho in range(s):
                    U = len(range(rho, k, s))
                    if U == 0:
                        continue
                    if sparse:
                        if n_ext > 0 and self.fuse_conv_gelu_bwd and l - 1 >= 1:
                            # the rows this GEMM writes (s g + rho, g in ext) are exactly act[l - 1]: d(pre[l - 1]) = bf16(d(post)) * gelu'(pre)
                            # straight from its epilogue -- the bits a bf16 d(post) tensor + wj_gelu_bwd_bf16 over act[l - 1] would give
                            ops.gemm(self.dpre_ptr[l] - (U - 1) * C * 2, self._conv_w[f"{si}:wd{l}_{rho}"], self.dpre_ptr[l - 1] + rho * C * 2,
                                     M=n_ext, N=C, K=U * C, lda=C, ldb=C, ldc=s * C, b_trans=1, rowmap=ext,
                                     epilogue=ops.EPI_MUL_GELU_GRAD_Z, aux=self.pre_ptr[l - 1] + rho * C * 2)
                        elif n_ext > 0:
                            ops.gemm(self.dpre_ptr[l] - (U - 1) * C * 2, self._conv_w[f"{si}:wd{l}_{rho}"], self.dpost_ptr[l - 1] + rho * C * 2,
                                     M=n_ext, N=C, K=U * C, lda=C, ldb=C, ldc=s * C, b_trans=1, rowmap=ext)
                    else:
                        ops.gemm(self.dpre_ptr[l] + r0 - (U - 1) * C * 2, self._conv_w[f"{si}:wd{l}_{rho}"],
                                 self.dpost_ptr[l - 1] + r0p + rho * C * 2, M=rows, N=C, K=U * C, lda=C, ldb=C, ldc=s * C, b_trans=1)
                if sparse and side_wgrad:
                    late_clear.append((self.dpre_ptr[l], act, n_act))
                elif sparse:
                    ops.zero_rows(self.dpre_ptr[l], act, n_rows=n_act, row_bytes=C * 2)
        _, k0, s0 = c.conv_spec[0]
        audio_p = self.audio.data_ptr()
        for ch in range(S):                  # layer 0: one call per stream (N mono clips each; ConvFeatureExtractor: one stream)
            pre = self.stacks[min(ch, len(self.stacks) - 1)]
            c0 = ch * N
            lists = {}
            if sparse:
                rows0, n0, off0, max0 = act_rows[0][ch]
                lists = dict(rows=rows0, row_off=off0, max_rows=max0)
            ops.conv0_bwd(audio_p + ch * c.n_samples * 2 if S > 1 else self.audio, f.ptr16(f"{pre}0.0.weight"), f.ptr32(f"{pre}0.2.weight"),
                          f.ptr32(f"{pre}0.2.bias"), self.gn_stats[0, c0:], self.gn_stats[1, c0:], self.dpost_ptr[0] + c0 * self.P[0] * C * 2,
                          f.gptr(f"{pre}0.0.weight"), f.gptr(f"{pre}0.2.weight"), f.gptr(f"{pre}0.2.bias"), self.gn_ws_b,
                          yx=self.gn_yx[c0:], x1=self.gn_x1[c0:], N=N, C_in=c.in_channels, L=c.n_samples, C=C, k=k0, stride=s0,
                          L_out=self.L[0], P=self.P[0], audio_clip_stride=S * c.in_channels * c.n_samples if S > 1 else 0, **lists)
            if sparse:
                ops.zero_rows(self.dpost_ptr[0] + c0 * self.P[0] * C * 2, rows0, n_rows=n0, row_bytes=C * 2)
        if late_clear:
            self._join_side()                # the side stream's conv weight gradients have read d(pre[l]): clear the rows now
            for ptr, act, n_act in late_clear:
                ops.zero_rows(ptr, act, n_rows=n_act, row_bytes=C * 2)

    def _conv_rows(self, plan: MaskPlan):
        """Device copies of conv_active_rows for this plan (cached on the plan: mask sets are reused by the data source).
        {l >= 1: [per stack group (act, n_act, ext, n_ext)]} with rows of the WHOLE layer buffer (a group's rows are a contiguous
        slice of the ascending list: conv clips are channel-major), {0: [per stream (rows, n, row_off, max_rows)]} with rows
        relative to the stream's first clip (one conv0 call per stream)."""
        cached = getattr(plan, "_conv_rows", None)
        if cached is not None and cached[0] == (self.N, self.S, len(self.stacks), tuple(self.P)):
            return cached[1]
        N, S = self.N, self.S
        keep = (plan.ctx_u8.cpu().numpy() == 0) if plan.ctx_np is None else ~plan.ctx_np
        if S > 1:                            # tokens (n, c, t) -> conv clip c*N + n
            keep = np.ascontiguousarray(keep.reshape(N, S, self.Tc).transpose(1, 0, 2)).reshape(S * N, self.Tc)
        lists = conv_active_rows(keep, self.P, self.cfg.conv_spec)
        pad = np.zeros(256, np.int32)        # the k-gather GEMM prefetches indices up to 256 entries past the end
        # every list of the step in ONE host -> device copy
        act0, off0 = lists[0]
        host, keys = [], []
        for l, (act, ext) in lists.items():
            if l == 0:
                continue
            host += [np.concatenate([act, pad]).astype(np.int32), np.concatenate([ext, pad]).astype(np.int32)]
            keys += [("act", l), ("ext", l)]
        per0_host = []
        for ch in range(S):
            lo, hi = int(off0[ch * N]), int(off0[(ch + 1) * N])
            rows = (act0[lo:hi] - ch * N * self.P[0]).astype(np.int32)
            off = (off0[ch * N:(ch + 1) * N + 1] - lo).astype(np.int32)
            per0_host.append((rows, off))
            host += [np.concatenate([rows, pad]).astype(np.int32), np.concatenate([off, pad]).astype(np.int32)]
            keys += [("rows0", ch), ("off0", ch)]
        dev = dict(zip(keys, pack_upload(host, self.dev)))

        groups = self._stack_groups()
        out = {}
        for l, (act, ext) in lists.items():
            if l == 0:
                continue
            d_act, d_ext = dev[("act", l)], dev[("ext", l)]
            per = []
            for _, c0, nclips in groups:
                lo, hi = c0 * self.P[l], (c0 + nclips) * self.P[l]
                a0, a1 = np.searchsorted(act, [lo, hi])
                e0, e1 = np.searchsorted(ext, [lo, hi])
                per.append((d_act.data_ptr() + 4 * int(a0), int(a1 - a0), d_ext.data_ptr() + 4 * int(e0), int(e1 - e0)))
            out[l] = per
            out[("keep", l)] = (d_act, d_ext)        # owners of the pointers above
        per0 = []
        for ch, (rows, off) in enumerate(per0_host):
            per0.append((dev[("rows0", ch)], int(rows.size), dev[("off0", ch)], int(np.diff(off).max()) if off.size > 1 else 0))
        out[0] = per0
        plan._conv_rows = ((self.N, self.S, len(self.stacks), tuple(self.P)), out)
        return out

    # ------------------------------------------------------------------------------------------------ EMA / inference
    def ema_step(self, r: float) -> None:
        f = self.flat
        self.wait_optimizer()
        ops.ema_update(f.p32.data_ptr() + 4 * f.enc_offset, f.t32, f.enc_numel, r, teacher_bf16=f.t16)

    def infer(self, audio: torch.Tensor, key_mask_u8: Optional[torch.Tensor]) -> torch.Tensor:
        """Student-only forward (reference jepa.py:456-467): returns fp32 [N, T, d_enc]."""
        c, f = self.cfg, self.flat
        N = audio.shape[0]
        if N != self.N:
            self.alloc(N, train=False)
        self.wait_optimizer()
        self._frontend(audio)
        a = self.scratch
        x, xb = self.lf, self.lf_b
        xq = False
        for w in self.enc_layers:
            xq = self._layer_fwd(w, a, x, xb, self.M, c.d_enc, c.h_enc, N, key_mask_u8, save=False, xq_ready=xq)
            x, xb = a.x2, a.x2b
        ops.layernorm_fwd(x, f.ptr32("encoder.norm.weight"), f.ptr32("encoder.norm.bias"), M=self.M, D=c.d_enc, eps=c.norm_eps,
                          y_f32=self.enc_out)
        return self.enc_out.view(N, self.T, c.d_enc)
