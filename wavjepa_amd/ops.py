"""Thin Python wrappers over the C ABI of libwavjepa_hip.so (one function per entry point).

Arguments are torch tensors (device memory owned by PyTorch's caching allocator) or raw integer device pointers;
every call is enqueued on PyTorch's *current* HIP stream and returns immediately.  Nothing here computes on the
host and nothing falls back to PyTorch ops: a missing library or a non-zero return code raises.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Union

import torch

from . import _abi
from ._abi import ENUMS, STRUCTS

Ptr = Union[torch.Tensor, int, None]

EPI_BF16 = ENUMS["WJ_EPI_BF16"]
EPI_BIAS_GELU2 = ENUMS["WJ_EPI_BIAS_GELU2"]
EPI_MUL_GELU_GRAD = ENUMS["WJ_EPI_MUL_GELU_GRAD"]
EPI_ADD_F32 = ENUMS["WJ_EPI_ADD_F32"]
EPI_ATOMIC_F32 = ENUMS["WJ_EPI_ATOMIC_F32"]
EPI_CONV_GELU = ENUMS["WJ_EPI_CONV_GELU"]
EPI_BIAS_GELU = ENUMS["WJ_EPI_BIAS_GELU"]
EPI_MUL_GELU_GRAD_Z = ENUMS["WJ_EPI_MUL_GELU_GRAD_Z"]
EPI_BF16_ADD_POS = ENUMS["WJ_EPI_BF16_ADD_POS"]
GROUP_STATS_SPLIT = _abi.DEFINES["WJ_GROUP_STATS_SPLIT"]
COLSUM_GROUP_MAX = _abi.DEFINES["WJ_COLSUM_GROUP_MAX"]


def _p(x: Ptr) -> int:
    if x is None:
        return 0
    if isinstance(x, int):
        return x
    return x.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


# When set to a list, every call is bracketed by HIP events on the launch stream (bench.py's live per-kernel timing):
# entries are (entry point, fields, start_event, end_event).  None (default) = zero overhead.
PROFILE = None


class TimingEvent:
    """A HIP event for TIMING ONLY, created with hipEventDisableSystemFence: a default event performs a system-scope fence when it
    is recorded -- an L2 write-back / invalidate between every two kernels it brackets, which production launches do not have and
    which both costs time inside the bracket and hands the next kernel cold caches (hip_runtime_api.h recommends the flag for exactly
    this use).  torch.cuda.Event cannot pass the flag, so the runtime torch already loaded is called directly."""
    _hip = None
    DISABLE_SYSTEM_FENCE = 0x20000000

    def __init__(self):
        import ctypes
        if TimingEvent._hip is None:
            import os
            TimingEvent._hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
        self._ev = ctypes.c_void_p()
        rc = TimingEvent._hip.hipEventCreateWithFlags(ctypes.byref(self._ev), ctypes.c_uint(TimingEvent.DISABLE_SYSTEM_FENCE))
        if rc != 0:
            raise RuntimeError(f"hipEventCreateWithFlags failed ({rc})")

    def record(self, stream: int) -> None:
        import ctypes
        rc = TimingEvent._hip.hipEventRecord(self._ev, ctypes.c_void_p(stream))
        if rc != 0:
            raise RuntimeError(f"hipEventRecord failed ({rc})")

    def elapsed_time(self, end: "TimingEvent") -> float:
        """milliseconds from this event to `end` (both recorded and complete: synchronise the device first)"""
        import ctypes
        ms = ctypes.c_float()
        rc = TimingEvent._hip.hipEventElapsedTime(ctypes.byref(ms), self._ev, end._ev)
        if rc != 0:
            raise RuntimeError(f"hipEventElapsedTime failed ({rc})")
        return float(ms.value)

    def __del__(self):
        try:
            if self._ev:
                TimingEvent._hip.hipEventDestroy(self._ev)
        except Exception:   # noqa: BLE001  (interpreter shutdown)
            pass


def _run(fn: str, struct_name: str, stream: Optional[int], _lab: bool = False, **fields) -> None:
    a = (_abi.LAB_STRUCTS if _lab else STRUCTS)[struct_name]()
    for k, v in fields.items():
        setattr(a, k, v)
    if PROFILE is None:
        _abi.call(fn, a, _stream() if stream is None else stream, lab=_lab)
        return
    s = _stream() if stream is None else stream
    e0, e1 = TimingEvent(), TimingEvent()
    e0.record(s)
    _abi.call(fn, a, s, lab=_lab)
    e1.record(s)
    PROFILE.append((fn, fields, e0, e1))


def require_gpu() -> None:
    _abi.load()
    if not torch.cuda.is_available():
        raise _abi.WavJepaHipError("wavjepa_amd needs a HIP device (MI355X); there is no CPU fallback")


def workspace_bytes(fn: str, **dims) -> int:
    """Scratch bytes entry point `fn` needs for the given dimensions (fields of its argument struct), from the library."""
    struct_name = {"wj_gemm_bf16": "wj_gemm_args", "wj_mask_scatter_fill_pos_bwd": "wj_scatter_fill_bwd_args", "wj_layernorm_bwd": "wj_ln_bwd_args", "wj_attn_bwd": "wj_attn_bwd_args", "wj_conv0_gn_gelu_fwd": "wj_conv0_fwd_args",
                   "wj_conv0_gn_gelu_bwd": "wj_conv0_bwd_args", "wj_masked_mse": "wj_mse_args", "wj_grad_sumsq": "wj_sumsq_args",
                   "wj_rir_convolve": "wj_rir_conv_args", "wj_snr_mix": "wj_snr_mix_args", "wj_mse_groups": "wj_mse_groups_args"}[fn]
    a = STRUCTS[struct_name]()
    for k, v in dims.items():
        setattr(a, k, v)
    return _abi.workspace_bytes(fn, a)


# ---------------------------------------------------------------------------------------------------------- GEMM
def gemm(A: Ptr, B: Ptr, C: Ptr, *, M: int, N: int, K: int, lda: int, ldb: int, ldc: int, a_trans: int = 0,
         b_trans: int = 0, epilogue: int = EPI_BF16, C2: Ptr = None, bias: Ptr = None, aux: Ptr = None, split_k: int = 1,
         seg_rows: int = 0, seg_valid: int = 0, alpha: float = 1.0, colsum: Ptr = None, rowmap: Ptr = None,
         workspace: Optional[torch.Tensor] = None, schedule: Optional[int] = None, persist_cus: Optional[int] = None,
         stream: Optional[int] = None) -> None:
    """workspace: a zero-initialised byte tensor the library may use for K-split pairs (see include/wavjepa_hip.h: wj_gemm_args).
    schedule: force tile / schedule variant 0..4 for this call (None: this binding's default, gemm_set_variant; -1: the library picks);
    persist_cus: resident persistent-GEMM workgroups per XCD (None: this binding's default, gemm_set_persist_cus)."""
    sched = _GEMM_SCHEDULE if schedule is None else int(schedule)
    _run("wj_gemm_bf16", "wj_gemm_args", stream, A=_p(A), B=_p(B), C=_p(C), C2=_p(C2), bias=_p(bias), aux=_p(aux), colsum=_p(colsum),
         rowmap=_p(rowmap),
         lda=lda, ldb=ldb, ldc=ldc, M=M, N=N, K=K, a_trans=a_trans, b_trans=b_trans, epilogue=epilogue, split_k=split_k,
         seg_rows=seg_rows, seg_valid=seg_valid, alpha=alpha, workspace=_p(workspace),
         workspace_bytes=0 if workspace is None else workspace.numel() * workspace.element_size(),
         schedule=0 if sched < 0 else sched + 1, persist_cus=_PERSIST_CUS if persist_cus is None else int(persist_cus))


def gemm_mxfp8(A8: Ptr, B8: Ptr, scale_a: Ptr, scale_b: Ptr, C: Ptr, *, M: int, N: int, K: int, lda: int, ldb: int, ldc: int,
               ld_scale_a: int, ld_scale_b: int, epilogue: int = EPI_BF16, C2: Ptr = None, bias: Ptr = None, q_out: Ptr = None,
               q_scales: Ptr = None, ld_q_scale: int = 0, stream: Optional[int] = None) -> None:
    """C = A . B^T on block-scaled e4m3 operands (see include/wavjepa_hip.h: wj_gemm_mxfp8).  q_out / q_scales: the GELU epilogues
    also emit gelu(h) as MX fp8 (C may then be None for EPI_BIAS_GELU)."""
    _run("wj_gemm_mxfp8", "wj_gemm_fp8_args", stream, A=_p(A8), B=_p(B8), scale_a=_p(scale_a), scale_b=_p(scale_b), C=_p(C), C2=_p(C2),
         bias=_p(bias), q_out=_p(q_out), q_scales=_p(q_scales), lda=lda, ldb=ldb, ldc=ldc, ld_scale_a=ld_scale_a, ld_scale_b=ld_scale_b,
         ld_q_scale=ld_q_scale, M=M, N=N, K=K, epilogue=epilogue)


def quantize_mxfp8(x: Ptr, q: Ptr, scales: Ptr, *, M: int, K: int, ldx: int, ldq: int, ld_scale: int, stream: Optional[int] = None) -> None:
    """bf16 [M][ldx] -> e4m3 bytes [M][ldq] + E8M0 block scales [K / 128][ld_scale] dwords."""
    _run("wj_quantize_mxfp8", "wj_quantize_fp8_args", stream, x=_p(x), q=_p(q), scales=_p(scales), ldx=ldx, ldq=ldq, ld_scale=ld_scale, M=M, K=K)


def fp8_scale_dwords(rows: int, K: int) -> int:
    """Dwords of a block-scale array for `rows` rows and K columns, including the 256 dwords of readable padding the GEMM asks for."""
    return (K // 128) * rows + 256


def wgrad_grouped(problems, stream: Optional[int] = None) -> None:
    """problems: [(dY, X, gW, n_out, k_in, m_tok)] (<= 8): gW[n_out, k_in] += dY[m_tok, n_out]^T @ X[m_tok, k_in], one launch."""
    a = STRUCTS["wj_wgrad_group_args"]()
    for i, (dY, X, gW, n_out, k_in, m_tok) in enumerate(problems):
        a.A[i], a.B[i], a.C[i] = _p(dY), _p(X), _p(gW)
        a.lda[i], a.ldb[i], a.ldc[i] = n_out, k_in, k_in
        a.M[i], a.N[i], a.K[i] = n_out, k_in, m_tok
    a.n = len(problems)
    if PROFILE is None:
        _abi.call("wj_wgrad_grouped", a, _stream() if stream is None else stream)
        return
    s = _stream() if stream is None else stream
    e0, e1 = TimingEvent(), TimingEvent()
    e0.record(s)
    _abi.call("wj_wgrad_grouped", a, s)
    e1.record(s)
    PROFILE.append(("wj_wgrad_grouped", dict(flops=sum(2.0 * p[3] * p[4] * p[5] for p in problems), n=len(problems),
                                             bytes=sum(2.0 * p[5] * (p[3] + p[4]) + 4.0 * p[3] * p[4] for p in problems)), e0, e1))


# Defaults this BINDING fills into wj_gemm_args.schedule / .persist_cus of every gemm() call that does not name them.  The library itself
# keeps no such state (round 5's wj_gemm_set_variant / wj_gemm_set_persist_cus were process-global setters behind a re-entrant ABI).
_GEMM_SCHEDULE = -1      # -1: the library picks; 0..4: force that variant (tests, tools/gemm_check.py)
_PERSIST_CUS = 0         # 0: the library's default (32, or WJ_PERSIST_CUS); 1..32: resident persistent-GEMM workgroups per XCD


def gemm_set_variant(variant: int) -> int:
    """Force the GEMM tile/schedule variant for the calls of this process's binding (tests, A/B runs); -1 = automatic.  Returns the
    previous setting.  Per call: gemm(..., schedule=v)."""
    global _GEMM_SCHEDULE
    prev, _GEMM_SCHEDULE = _GEMM_SCHEDULE, (-1 if variant < 0 else int(variant))
    return prev


def gemm_release_stream(stream: int) -> None:
    """Hand the persistent GEMM's counter set of a stream that is about to be destroyed back to the library (optional; see the header)."""
    rc = int(_abi.load().wj_gemm_release_stream(ctypes.c_void_p(int(stream))))
    if rc != 0:
        raise _abi.WavJepaHipError(f"wj_gemm_release_stream failed: {rc}")


def transpose_bf16(src: Ptr, dst: Ptr, table: torch.Tensor, n_mats: int, n_tiles: int, stream: Optional[int] = None) -> None:
    """Batched in-layout transposes: for every row (element offset, rows, cols, first tile) of `table` (int64 [n_mats][4], device):
    dst[off + c*rows + r] = src[off + r*cols + c]."""
    _run("wj_transpose_bf16", "wj_transpose_args", stream, src=_p(src), dst=_p(dst), table=_p(table), n_mats=n_mats, n_tiles=n_tiles)


def gemm_set_persist_cus(workgroups_per_xcd: int) -> int:
    """Resident workgroups per XCD of the persistent GEMM for the calls of this binding (1..32; <= 0 only queries).  Returns the value that
    was in effect (32 when neither this nor WJ_PERSIST_CUS has set one)."""
    global _PERSIST_CUS
    import os
    prev = _PERSIST_CUS or max(1, min(32, int(os.environ.get("WJ_PERSIST_CUS", "32") or 32)))
    if workgroups_per_xcd > 0:
        _PERSIST_CUS = min(32, int(workgroups_per_xcd))
    return prev


def pick_split_k(M: int, N: int, K: int) -> int:
    """Split-K factor for the wgrad GEMM: fill the 256 CUs (tile 256 x 256 -> 1 workgroup/CU, 256 x 128 -> 2)."""
    bn = 256 if N % 256 == 0 else 128           # mirrors pick_variant() in csrc/gemm.hip for the ATOMIC (wgrad) epilogue
    tiles = ((M + 255) // 256) * ((N + bn - 1) // bn)
    target = 256 if bn == 256 else 512
    s = max(1, (target + tiles // 2) // max(1, tiles))
    return max(1, min(s, (K + 1023) // 1024))


# ---------------------------------------------------------------------------------------------------------- norms
def layernorm_fwd(x: Ptr, gamma: Ptr, beta: Ptr, *, M: int, D: int, eps: float, r: Ptr = None, y_f32: Ptr = None,
                  y_bf16: Ptr = None, mean: Ptr = None, rstd: Ptr = None, x_is_bf16: bool = False, in_seg: int = 0,
                  in_valid: int = 0, group_stats: Ptr = None, group_rows: int = 0, in_chan: int = 0, y_fp8: Ptr = None,
                  y_fp8_scales: Ptr = None, ld_fp8_scale: int = 0, workgroups: int = 0, stream: Optional[int] = None) -> None:
    _run("wj_layernorm_fwd", "wj_ln_fwd_args", stream, x=_p(x), r=_p(r), gamma=_p(gamma), beta=_p(beta), y_f32=_p(y_f32),
         y_bf16=_p(y_bf16), mean=_p(mean), rstd=_p(rstd), group_stats=_p(group_stats), y_fp8=_p(y_fp8), y_fp8_scales=_p(y_fp8_scales),
         ld_fp8_scale=ld_fp8_scale, M=M, D=D, x_is_bf16=int(x_is_bf16), in_seg=in_seg, in_valid=in_valid, group_rows=group_rows,
         in_chan=in_chan, eps=eps, workgroups=int(workgroups))


def layernorm_bwd(dy: Ptr, x: Ptr, gamma: Ptr, mean: Ptr, rstd: Ptr, *, M: int, D: int, r: Ptr = None, dy2: Ptr = None,
                  ds_f32: Ptr = None, ds_bf16: Ptr = None, dgamma: Ptr = None, dbeta: Ptr = None, dbias: Ptr = None,
                  x_is_bf16: bool = False, in_seg: int = 0, in_valid: int = 0, out_seg: int = 0, out_valid: int = 0,
                  chan: int = 0, workspace: Ptr = None, dy2_is_bf16: bool = False, stream: Optional[int] = None) -> None:
    _run("wj_layernorm_bwd", "wj_ln_bwd_args", stream, dy=_p(dy), dy2=_p(dy2), dy2_is_bf16=int(dy2_is_bf16), x=_p(x), r=_p(r), gamma=_p(gamma),
         mean=_p(mean), rstd=_p(rstd), ds_f32=_p(ds_f32), ds_bf16=_p(ds_bf16), dgamma=_p(dgamma), dbeta=_p(dbeta),
         dbias=_p(dbias), workspace=_p(workspace), M=M, D=D, x_is_bf16=int(x_is_bf16), in_seg=in_seg, in_valid=in_valid, out_seg=out_seg,
         out_valid=out_valid, chan=chan)


def ln_bwd_partial_rows(M: int, D: int) -> int:
    """Rows of partials layernorm_bwd leaves in its workspace ([rows][3][D]) for M token rows of width D."""
    return int(_abi.load().wj_ln_bwd_partial_rows(int(M), int(D)))


def colsum_f32_group(items, stream: Optional[int] = None) -> None:
    """items: [(x, ldx, M, N, o0, o1, o2, n_each)] (<= 16): one launch folds every matrix's column sums into its outputs (+=)."""
    a = STRUCTS["wj_colsum_group_args"]()
    for i, (x, ldx, M, N, o0, o1, o2, n_each) in enumerate(items):
        a.x[i], a.o0[i], a.o1[i], a.o2[i] = _p(x), _p(o0), _p(o1), _p(o2)
        a.ldx[i], a.M[i], a.N[i], a.n_each[i] = ldx, M, N, n_each
    a.n = len(items)
    s = _stream() if stream is None else stream
    if PROFILE is None:
        _abi.call("wj_colsum_f32_group", a, s)
        return
    e0, e1 = TimingEvent(), TimingEvent()
    e0.record(s)
    _abi.call("wj_colsum_f32_group", a, s)
    e1.record(s)
    PROFILE.append(("wj_colsum_f32_group", dict(n=len(items)), e0, e1))


def rccl_unique_id() -> bytes:
    """128 opaque bytes naming a new RCCL communicator (make on ONE rank, hand to every rank's rccl_bucket_allreduce_init)."""
    import ctypes
    buf = ctypes.create_string_buffer(128)
    rc = _abi.load().wj_rccl_unique_id(ctypes.cast(buf, ctypes.c_void_p))
    if rc != 0:
        raise _abi.WavJepaHipError(f"wj_rccl_unique_id failed with {rc}" + (" (no librccl.so in the process or on the library path)" if rc == -3 else ""))
    return buf.raw


def rccl_bucket_allreduce_init(unique_id: bytes, rank: int, world: int) -> None:
    """Collective: returns once every rank of `world` has called it with the same id."""
    import ctypes
    if len(unique_id) != 128:
        raise ValueError("an RCCL unique id is 128 bytes")
    buf = ctypes.create_string_buffer(unique_id, 128)
    a = STRUCTS["wj_rccl_init_args"](unique_id=ctypes.cast(buf, ctypes.c_void_p).value, rank=rank, world=world)
    rc = _abi.load().wj_rccl_bucket_allreduce_init(ctypes.addressof(a))
    if rc != 0:
        raise _abi.WavJepaHipError(f"wj_rccl_bucket_allreduce_init(rank {rank} of {world}) failed with {rc}")


def rccl_bucket_allreduce_launch(buf: Ptr, count: int, *, average: bool = True, stream: Optional[int] = None) -> None:
    _run("wj_rccl_bucket_allreduce_launch", "wj_rccl_launch_args", stream, buf=_p(buf), count=count, average=int(average))


def rccl_bucket_allreduce_wait(on_stream: int, stream: Optional[int] = None) -> None:
    """The stream (default: current) waits for what `on_stream` has queued; the host does not."""
    _run("wj_rccl_bucket_allreduce_wait", "wj_rccl_wait_args", stream, on_stream=on_stream)


def rccl_bucket_allreduce_finalize() -> None:
    _abi.load().wj_rccl_bucket_allreduce_finalize()


def colsum_bf16(x: Ptr, out: Ptr, *, M: int, N: int, ldx: int, stream: Optional[int] = None) -> None:
    _run("wj_colsum_bf16", "wj_colsum_args", stream, x=_p(x), out=_p(out), ldx=ldx, M=M, N=N)


def colsum_f32(x: Ptr, out: Ptr, *, M: int, N: int, ldx: int, stream: Optional[int] = None) -> None:
    _run("wj_colsum_f32", "wj_colsum_args", stream, x=_p(x), out=_p(out), ldx=ldx, M=M, N=N)


# ---------------------------------------------------------------------------------------------------------- attention
def attn_fwd(qkv: Ptr, out: Ptr, *, B: int, T: int, H: int, hd: int, key_mask: Ptr = None, lse: Ptr = None,
             mask_group: int = 1, seq_off: Ptr = None, stream: Optional[int] = None) -> None:
    """seq_off (int32 [B+1]) selects the ragged form: packed sequences, T = length bound, lse [rows][H]."""
    _run("wj_attn_fwd", "wj_attn_fwd_args", stream, qkv=_p(qkv), key_mask=_p(key_mask), seq_off=_p(seq_off), out=_p(out),
         lse=_p(lse), B=B, T=T, H=H, hd=hd, mask_group=mask_group)


def attn_bwd(qkv: Ptr, out: Ptr, dout: Ptr, lse: Ptr, dqkv: Ptr, *, B: int, T: int, H: int, hd: int, key_mask: Ptr = None,
             mask_group: int = 1, dbias: Ptr = None, dbias_ws: Ptr = None, seq_off: Ptr = None, defer_fold: bool = False,
             stream: Optional[int] = None) -> None:
    _run("wj_attn_bwd", "wj_attn_bwd_args", stream, qkv=_p(qkv), key_mask=_p(key_mask), seq_off=_p(seq_off), out=_p(out), dout=_p(dout),
         lse=_p(lse), dqkv=_p(dqkv), dbias=_p(dbias), dbias_ws=_p(dbias_ws), B=B, T=T, H=H, hd=hd, mask_group=mask_group,
         defer_fold=int(defer_fold))


# ---------------------------------------------------------------------------------------------------------- conv front-end
def conv0_fwd(audio: Ptr, w: Ptr, gamma: Ptr, beta: Ptr, act: Ptr, mean: Ptr, rstd: Ptr, workspace: Ptr, *, N: int,
              C_in: int, L: int, C: int, k: int, stride: int, L_out: int, P: int, eps: float = 1e-5, yx: Ptr = None,
              x1: Ptr = None, audio_clip_stride: int = 0, stream: Optional[int] = None) -> None:
    _run("wj_conv0_gn_gelu_fwd", "wj_conv0_fwd_args", stream, audio=_p(audio), w=_p(w), gamma=_p(gamma), beta=_p(beta),
         act=_p(act), mean=_p(mean), rstd=_p(rstd), workspace=_p(workspace), yx=_p(yx), x1=_p(x1), N=N, C_in=C_in, L=L, C=C,
         k=k, stride=stride, L_out=L_out, P=P, eps=eps, audio_clip_stride=audio_clip_stride)


def conv0_bwd(audio: Ptr, w: Ptr, gamma: Ptr, beta: Ptr, mean: Ptr, rstd: Ptr, dact: Ptr, dw: Ptr, dgamma: Ptr, dbeta: Ptr,
              workspace: Ptr, *, yx: Ptr, x1: Ptr, N: int, C_in: int, L: int, C: int, k: int, stride: int, L_out: int, P: int,
              rows: Ptr = None, row_off: Ptr = None, max_rows: int = 0, audio_clip_stride: int = 0,
              stream: Optional[int] = None) -> None:
    """rows / row_off / max_rows: read the output gradient on the listed rows only (see include/wavjepa_hip.h)."""
    _run("wj_conv0_gn_gelu_bwd", "wj_conv0_bwd_args", stream, audio=_p(audio), w=_p(w), gamma=_p(gamma), beta=_p(beta),
         mean=_p(mean), rstd=_p(rstd), dact=_p(dact), yx=_p(yx), x1=_p(x1), rows=_p(rows), row_off=_p(row_off), dw=_p(dw),
         dgamma=_p(dgamma), dbeta=_p(dbeta), workspace=_p(workspace), N=N, C_in=C_in, L=L, C=C, k=k, stride=stride, L_out=L_out,
         P=P, max_rows=max_rows, audio_clip_stride=audio_clip_stride)


def gelu_bwd_bf16(dpost: Ptr, pre: Ptr, dpre: Ptr, n: int, *, rows: Ptr = None, n_rows: int = 0, row_elems: int = 0,
                  clear_dpost: bool = False, stream: Optional[int] = None) -> None:
    """rows (int32 [n_rows]) selects the listed-rows form over [.][row_elems] matrices."""
    _run("wj_gelu_bwd_bf16", "wj_gelu_bwd_args", stream, dpost=_p(dpost), pre=_p(pre), dpre=_p(dpre), rows=_p(rows), n=n,
         n_rows=n_rows, row_elems=row_elems, clear_dpost=int(clear_dpost))


def spin(ticks: int, stream: Optional[int] = None) -> None:
    """One wave busy-waiting for `ticks` s_memtime ticks (stream-concurrency probe)."""
    _run("wj_spin", "wj_spin_args", stream, ticks=int(ticks))


def collective_footprint(buf: Ptr, nbytes: int, *, workgroups: int = 32, passes: int = 2, min_ticks: int = 0,
                         stream: Optional[int] = None) -> None:
    """Measurement aid (bench.py --emulate-allreduce): the on-GPU footprint of an all-reduce of `nbytes` at `buf`; an entry of the LABORATORY
    library (include/wavjepa_hip_lab.h), loaded on first use."""
    _run("wj_collective_footprint", "wj_collective_footprint_args", stream, _lab=True, buf=_p(buf), bytes=int(nbytes), min_ticks=int(min_ticks),
         workgroups=int(workgroups), passes=int(passes))


def zero_rows(buf: Ptr, rows: Ptr, *, n_rows: int, row_bytes: int, stream: Optional[int] = None) -> None:
    _run("wj_zero_rows", "wj_zero_rows_args", stream, buf=_p(buf), rows=_p(rows), n_rows=n_rows, row_bytes=row_bytes)


def conv_weight_layout(src: Ptr, dst: Ptr, *, C_out: int, C_in: int, k: int, mode: int, stride: int = 1, rho: int = 0,
                       U: int = 1, stream: Optional[int] = None) -> None:
    _run("wj_conv_weight_layout", "wj_conv_w_args", stream, src=_p(src), dst=_p(dst), C_out=C_out, C_in=C_in, k=k,
         stride=stride, rho=rho, U=U, mode=mode)


# ---------------------------------------------------------------------------------------------------------- tokens
def add_pos(x: Ptr, pos: Ptr, *, M: int, T: int, D: int, y_f32: Ptr = None, y_bf16: Ptr = None,
            stream: Optional[int] = None) -> None:
    _run("wj_add_pos", "wj_add_pos_args", stream, x=_p(x), pos=_p(pos), y_f32=_p(y_f32), y_bf16=_p(y_bf16), M=M, T=T, D=D)


def mask_gather_rows(x: Ptr, idx: Ptr, out: Ptr, *, n_rows: int, D: int, elem_bytes: int, stream: Optional[int] = None) -> None:
    _run("wj_mask_gather_rows", "wj_gather_args", stream, x=_p(x), idx=_p(idx), out=_p(out), n_rows=n_rows, D=D,
         elem_bytes=elem_bytes)


def mask_scatter_fill_pos(ctx_feats: Ptr, inv: Ptr, mask_token: Ptr, pos: Ptr, *, B: int, T: int, D: int, G: int,
                          out_f32: Ptr = None, out_bf16: Ptr = None, rows: Ptr = None, n_rows: int = 0,
                          stream: Optional[int] = None) -> None:
    """rows (int32 [n_rows] of dense (b*G+g)*T+t) selects the ragged form: only those rows, packed."""
    _run("wj_mask_scatter_fill_pos", "wj_scatter_fill_args", stream, ctx_feats=_p(ctx_feats), inv=_p(inv),
         mask_token=_p(mask_token), pos=_p(pos), rows=_p(rows), out_f32=_p(out_f32), out_bf16=_p(out_bf16), B=B, T=T, D=D, G=G,
         n_rows=n_rows)


def mask_scatter_fill_pos_bwd(d_in: Ptr, inv: Ptr, d_ctx_feats: Ptr, d_mask_token: Ptr, *, B: int, T: int, D: int, G: int,
                              rowmap: Ptr = None, partials: Ptr = None, stream: Optional[int] = None) -> None:
    """partials: scratch for one row of D floats per workgroup (scatter_fill_bwd_partial_rows(B, T) rows); the mask-token gradient is
    then left there as partial rows for a column-sum fold instead of being added with float atomics."""
    _run("wj_mask_scatter_fill_pos_bwd", "wj_scatter_fill_bwd_args", stream, d_in=_p(d_in), inv=_p(inv), rowmap=_p(rowmap),
         d_ctx_feats=_p(d_ctx_feats), d_mask_token=_p(d_mask_token), partials=_p(partials), B=B, T=T, D=D, G=G)


def scatter_fill_bwd_partial_rows(B: int, T: int) -> int:
    return int(_abi.load().wj_scatter_fill_bwd_partial_rows(int(B), int(T)))


def unmask_rows_f32(src: Ptr, inv: Ptr, dst: Ptr, *, M: int, D: int, src_is_f32: bool = False, dst_is_bf16: bool = False,
                    stream: Optional[int] = None) -> None:
    """dst[m] = inv[m] >= 0 ? src[inv[m]] : 0 (bf16 -> f32 unless the flags say otherwise; inv None = identity)."""
    _run("wj_unmask_rows_f32", "wj_unmask_rows_args", stream, src=_p(src), inv=_p(inv), dst=_p(dst), M=M, D=D,
         src_is_f32=int(src_is_f32), dst_is_bf16=int(dst_is_bf16))


# ---------------------------------------------------------------------------------------------------------- targets / loss
def instnorm_accumulate(x: Ptr, targets: Ptr, *, B: int, TD: int, accumulate: bool, scale: float, eps: float = 1e-5,
                        stream: Optional[int] = None) -> None:
    _run("wj_instnorm_accumulate", "wj_instnorm_args", stream, x=_p(x), targets=_p(targets), B=B, TD=TD,
         accumulate=int(accumulate), scale=scale, eps=eps)


def instnorm_mean(xs, stats: Ptr, targets: Ptr, *, B: int, TD: int, eps: float = 1e-5, stream: Optional[int] = None) -> None:
    """targets = mean over the K = len(xs) <= 8 tensors of their per-sample instance norm, from precomputed (sum, sumsq)."""
    kw = {f"x{i}": _p(x) for i, x in enumerate(xs)}
    _run("wj_instnorm_mean", "wj_instnorm_mean_args", stream, stats=_p(stats), targets=_p(targets), B=B, TD=TD, K=len(xs), eps=eps, **kw)


def masked_mse(preds: Ptr, targets: Ptr, tgt: Ptr, loss: Ptr, workspace: Ptr, *, B: int, G: int, T: int, D: int,
               dpreds: Ptr = None, gscale: float = 1.0, gscale_ptr: Ptr = None, rows: Ptr = None, n_rows: int = 0,
               stream: Optional[int] = None) -> None:
    _run("wj_masked_mse", "wj_mse_args", stream, preds=_p(preds), targets=_p(targets), tgt=_p(tgt), rows=_p(rows), loss=_p(loss),
         dpreds=_p(dpreds), workspace=_p(workspace), gscale_ptr=_p(gscale_ptr), B=B, G=G, T=T, D=D, n_rows=n_rows, gscale=gscale)


# ---------------------------------------------------------------------------------------------------------- optimiser side
def ema_update(student: Ptr, teacher: Ptr, n: int, r: float, teacher_bf16: Ptr = None, stream: Optional[int] = None) -> None:
    _run("wj_ema_update", "wj_ema_args", stream, student=_p(student), teacher=_p(teacher), teacher_bf16=_p(teacher_bf16), n=n, r=r)


def grad_sumsq(g: Ptr, out: Ptr, workspace: Ptr, n: int, accumulate: bool = False, workgroups: int = 0, stream: Optional[int] = None) -> None:
    _run("wj_grad_sumsq", "wj_sumsq_args", stream, g=_p(g), out=_p(out), workspace=_p(workspace), n=n, accumulate=int(accumulate),
         workgroups=int(workgroups))


def adamw_step(p: Ptr, g: Ptr, m: Ptr, v: Ptr, n: int, *, lr: float, beta1: float, beta2: float, eps: float,
               weight_decay: float, step: int, max_norm: float = 0.0, sumsq: Ptr = None, p_bf16: Ptr = None,
               grad_scale: float = 1.0, workgroups: int = 0, zero_grad: bool = False, stream: Optional[int] = None) -> None:
    _run("wj_adamw_step", "wj_adamw_args", stream, workgroups=int(workgroups), zero_grad=int(zero_grad), p=_p(p), g=_p(g), m=_p(m), v=_p(v), p_bf16=_p(p_bf16), sumsq=_p(sumsq),
         n=n, lr=lr, beta1=beta1, beta2=beta2, eps=eps, weight_decay=weight_decay, bc1=1.0 - beta1 ** step,
         bc2=1.0 - beta2 ** step, max_norm=max_norm, grad_scale=grad_scale)


def cast_f32_to_bf16(src: Ptr, dst: Ptr, n: int, stream: Optional[int] = None) -> None:
    _run("wj_cast_f32_to_bf16", "wj_cast_args", stream, src=_p(src), dst=_p(dst), n=n)


def crop_normalize_bf16(src: Ptr, starts: Ptr, out: Ptr, *, B: int, S: int, C: int, L_full: int, length: int,
                        perm_inv: Ptr = None, stream: Optional[int] = None) -> None:
    _run("wj_crop_normalize_bf16", "wj_crop_args", stream, src=_p(src), starts=_p(starts), perm_inv=_p(perm_inv), out=_p(out),
         B=B, S=S, C=C, L_full=L_full, length=length)


# ---------------------------------------------------------------------------------------------------------- scene augmentation
def rir_convolve(x: Ptr, h: Ptr, y: Ptr, workspace: Ptr, *, B: int, C: int, T: int, L: int, h_stride_b: int, h_stride_c: int,
                 accumulate: bool = False, fft_size: int = 0, stream: Optional[int] = None) -> None:
    """y[b][c][:T] (+)= (x[b] * h[b][c])[:T]  (generate_scenes_batch.py:12-44); fp32, workspace from workspace_bytes("wj_rir_convolve")."""
    _run("wj_rir_convolve", "wj_rir_conv_args", stream, x=_p(x), h=_p(h), y=_p(y), workspace=_p(workspace), h_stride_b=h_stride_b,
         h_stride_c=h_stride_c, B=B, C=C, T=T, L=L, accumulate=int(accumulate), fft_size=fft_size)


def snr_mix(source: Ptr, noise: Ptr, out: Ptr, snr: Ptr, start: Ptr, length: Ptr, workspace: Ptr, *, B: int, C: int, T: int,
            stream: Optional[int] = None) -> None:
    """out = source + a * noise with the segmental-SNR gain a of generate_scenes_batch.py:108-150."""
    _run("wj_snr_mix", "wj_snr_mix_args", stream, source=_p(source), noise=_p(noise), out=_p(out), snr=_p(snr), start=_p(start),
         length=_p(length), workspace=_p(workspace), B=B, C=C, T=T)


# ---------------------------------------------------------------------------------------------------------- denoiser stage
def resample_fir(x: Ptr, kernel: Ptr, y: Ptr, *, B: int, L_in: int, L_out: int, orig: int, nw: int, width: int,
                 stream: Optional[int] = None) -> None:
    """Polyphase FIR resampling with a [nw][2 * width + orig] kernel table (torchaudio.functional.resample's application step)."""
    _run("wj_resample_fir", "wj_resample_args", stream, x=_p(x), kernel=_p(kernel), y=_p(y), B=B, L_in=L_in, L_out=L_out, orig=orig, nw=nw,
         width=width, taps=2 * width + orig)


def mse_groups(preds: Ptr, targets: Ptr, w: Ptr, loss: Ptr, workspace: Ptr, *, n: int, G: int, dpreds: Ptr = None, gscale: Ptr = None,
               stream: Optional[int] = None) -> None:
    """loss[0] = sum_g w[g] * mean((preds[g] - targets)^2), loss[1 + g] the per-set means; optional gradient (denoiser.py:350-355)."""
    _run("wj_mse_groups", "wj_mse_groups_args", stream, preds=_p(preds), targets=_p(targets), w=_p(w), gscale=_p(gscale), loss=_p(loss),
         dpreds=_p(dpreds), workspace=_p(workspace), n=n, G=G)
