"""Data parallelism for the flat-gradient engine: one process per GPU, RCCL all-reduce (average) of the flat fp32
gradient buffer in a few large contiguous buckets, launched from the backward's section hooks so that the predictor /
encoder buckets travel over xGMI while the rest of the backward (encoder layers, conv stack) still computes.

Replaces Lightning's `strategy="ddp"` (reference train.py:174-179): same semantics -- every rank normalises its loss by
its LOCAL target count and the gradients are averaged with equal rank weights -- without per-tensor hooks
or 25 MB autograd buckets: xGMI is point-to-point (7 links/GPU), so fewer, larger collectives are the better fit.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple

import torch.distributed as dist

from .params import FlatParams


def section_ranges(flat: FlatParams, enc_layers: int, enc_chunk: int = 3) -> Dict[str, List[Tuple[int, int]]]:
    """Map the engine's backward tags to contiguous [lo, hi) ranges of the flat gradient buffer."""
    def span(prefixes) -> Tuple[int, int]:
        sl = [s for s in flat.slots if any(s.name.startswith(p) for p in prefixes)]
        lo = min(s.offset for s in sl)
        hi = max(s.offset + (s.numel + 7) // 8 * 8 for s in sl)
        return lo, hi

    out: Dict[str, List[Tuple[int, int]]] = {}
    out["dec"] = [span(["decoder.", "decoder_to_encoder_mapper.", "encoder_to_decoder_mapper."])]
    # encoder layers are final in descending order; ship them in chunks of `enc_chunk` layers when the lowest is done
    hi_layer = enc_layers - 1
    while hi_layer >= 0:
        lo_layer = max(0, hi_layer - enc_chunk + 1)
        prefixes = [f"encoder.layers.{i}." for i in range(lo_layer, hi_layer + 1)]
        if hi_layer == enc_layers - 1:
            prefixes.append("encoder.norm.")
        out[f"enc:{lo_layer}"] = [span(prefixes)]
        hi_layer = lo_layer - 1
    front = [span(["mask_token", "extract_audio.", "feature_norms."])]
    if any(s.name.startswith("post_extraction_mapper.") for s in flat.slots):
        front.append(span(["post_extraction_mapper."]))
    out["front"] = front
    covered = sorted(r for v in out.values() for r in v)
    pos = 0
    for lo, hi in covered:
        assert lo == pos, "gradient buckets must tile the flat buffer"
        pos = hi
    assert pos == flat.n
    return out


def process_group_backend(pg=None) -> str:
    return str(dist.get_backend(pg)).lower()


class FlatGradAllReducer:
    def __init__(self, module, process_group=None, enc_chunk: int = 3):
        self.module = module
        self.pg = process_group
        self.enc_chunk = enc_chunk
        self.handles: List = []
        self._ranges: Optional[Dict[str, List[Tuple[int, int]]]] = None
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # a process group of size 1 (torch.distributed.run --nproc-per-node 1) still runs every collective: the single-GPU
        # box exercises exactly the code path the 8-GPU launch uses
        self.active = dist.is_initialized()
        # optional timing (bench.py): HIP events on the compute stream at the first bucket's issue, at the end of the backward
        # (= entry of wait()) and behind the last handle's wait -- (done - backward end) is the part of the all-reduce the
        # backward did NOT hide, (backward end - first issue) the window the buckets had to travel in
        self.timing = False
        self.timeline: List[Tuple] = []
        self._t_first = None
        # WJ_RCCL_DIRECT=1: the gradient buckets go through the library's own RCCL binding (wj_rccl_bucket_allreduce_*, the C-ABI
        # family a host without torch.distributed would use) on a communication stream of their own; the process group then only
        # carries the 128-byte communicator id and the start-up broadcasts.  Same averages, same bucket order.
        self.direct = self.active and os.environ.get("WJ_RCCL_DIRECT", "0") == "1"
        self._comm_stream = None
        if self.direct:
            self._init_direct()
        # WJ_EMULATE_ALLREDUCE=1 (bench.py --emulate-allreduce; one GPU, no process group): at every bucket hook a copy kernel confined
        # to WJ_EMULATE_WORKGROUPS resident workgroups (default 32 = the CUs a data-parallel run keeps free) reads and rewrites the
        # bucket twice on a communication stream (ops.collective_footprint), paced to WJ_EMULATE_BUSBW_GBPS (an ASSUMED all-reduce bus
        # bandwidth; 0 = unpaced).  A rehearsal of the collective's CU + HBM share on this GPU -- nothing crosses xGMI, nothing is reduced.
        self.emulate = (not self.active) and os.environ.get("WJ_EMULATE_ALLREDUCE", "0") == "1"
        if self.emulate:
            import torch
            self._comm_stream = torch.cuda.Stream()
            self._emu_wgs = int(os.environ.get("WJ_EMULATE_WORKGROUPS", "32"))
            self._emu_gbps = float(os.environ.get("WJ_EMULATE_BUSBW_GBPS", "0"))
            self._emu_world = int(os.environ.get("WJ_EMULATE_WORLD", "8"))

    def _emulated_launch(self, view) -> None:
        import torch
        from . import ops
        nbytes = view.numel() * 4
        ticks = 0
        if self._emu_gbps > 0:      # ring all-reduce: 2 (W - 1) / W x bytes over the bus bandwidth; ticks of the 100 MHz clock
            ticks = int(2.0 * (self._emu_world - 1) / self._emu_world * nbytes / (self._emu_gbps * 1e9) * 1e8)
        self._comm_stream.wait_stream(torch.cuda.current_stream())
        ops.collective_footprint(view.data_ptr(), nbytes - nbytes % 16, workgroups=self._emu_wgs, passes=2, min_ticks=ticks,
                                 stream=self._comm_stream.cuda_stream)

    def _init_direct(self) -> None:
        import torch
        from . import ops
        if process_group_backend(self.pg) != "nccl":
            raise RuntimeError("WJ_RCCL_DIRECT=1 needs one GPU per rank (backend nccl): RCCL refuses two ranks on one device")
        rank = dist.get_rank(self.pg)
        box = [ops.rccl_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=self.pg)
        ops.rccl_bucket_allreduce_init(box[0], rank, self.world)
        self._comm_stream = torch.cuda.Stream()

    def _direct_launch(self, view) -> None:
        """Bucket `view` (contiguous fp32 slice of the flat gradient buffer) averaged in place on the communication stream, ordered
        behind everything the compute stream has queued so far (the kernels that produced it)."""
        import torch
        from . import ops
        self._comm_stream.wait_stream(torch.cuda.current_stream())
        ops.rccl_bucket_allreduce_launch(view.data_ptr(), view.numel(), average=True, stream=self._comm_stream.cuda_stream)

    def broadcast_parameters(self) -> None:
        """Rank 0's student, teacher and optimiser-visible state to everyone (DDP constructor broadcast)."""
        if not self.active:
            return
        self.module._ensure_engine()
        flat = self.module._flat
        dist.broadcast(flat.p32, 0, group=self.pg)
        if flat.tn > 0:
            dist.broadcast(flat.t32, 0, group=self.pg)
        # module state outside the two flat buffers (the fixed sin-cos position tables: requires_grad=False parameters, and any
        # buffer): DDP's constructor syncs those from rank 0 too.  In place -- the engine reads the tables through these storages (or
        # refreshes its converted copy in prepare_weights).
        owned = set(flat.by_name) | set(flat.tby_name)
        rest = [(n, t) for n, t in list(self.module.named_parameters()) + list(self.module.named_buffers()) if n not in owned]
        for _, t in sorted(rest, key=lambda nt: nt[0]):
            dist.broadcast(t.data, 0, group=self.pg)
        self.module._student_bf16_fresh = False
        self.module._teacher_bf16_fresh = False

    def reduce_all(self) -> None:
        """One average over the whole flat gradient buffer (modules whose backward exposes no section hooks: the denoiser stage)."""
        if self.direct:
            self._direct_launch(self.module._flat.g32)
        elif self.active:
            self.handles.append(dist.all_reduce(self.module._flat.g32, op=dist.ReduceOp.AVG, group=self.pg, async_op=True))

    def hook(self, tag: str) -> None:
        if not (self.active or self.emulate):
            return
        flat = self.module._flat
        if self._ranges is None:
            self._ranges = section_ranges(flat, self.module.encoder.num_layers, self.enc_chunk)
        if self.timing and self._t_first is None:
            import torch
            self._t_first = torch.cuda.Event(enable_timing=True)
            self._t_first.record()
        for lo, hi in self._ranges.get(tag, []):
            if self.emulate:
                self._emulated_launch(flat.g32[lo:hi])
            elif self.direct:
                self._direct_launch(flat.g32[lo:hi])
            else:
                self.handles.append(dist.all_reduce(flat.g32[lo:hi], op=dist.ReduceOp.AVG, group=self.pg, async_op=True))

    def wait(self) -> None:
        t_end = t_done = None
        if self.timing and (self.handles or self.direct or self.emulate):
            import torch
            t_end, t_done = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t_end.record()
        for h in self.handles:
            h.wait()          # stream-level dependency; does not block the host
        if self.direct:
            from . import ops
            ops.rccl_bucket_allreduce_wait(self._comm_stream.cuda_stream)
        if self.emulate:
            import torch
            torch.cuda.current_stream().wait_stream(self._comm_stream)
        if t_done is not None:
            t_done.record()
            self.timeline.append((self._t_first, t_end, t_done))
            self._t_first = None
        self.handles = []

    def timing_summary(self):
        """Median over the recorded steps (call after a device synchronisation): ms between the first bucket's issue and the end
        of the backward, and ms the optimiser then still waited for the collectives."""
        if not self.timeline:
            return None
        win = sorted(a.elapsed_time(b) for a, b, _ in self.timeline if a is not None)
        exp = sorted(b.elapsed_time(c) for _, b, c in self.timeline)
        n_buckets = sum(len(v) for v in (self._ranges or {}).values())
        return dict(buckets=n_buckets, bytes=int(self.module._flat.n) * 4, steps=len(exp),
                    backward_window_ms=round(win[len(win) // 2], 3) if win else None, exposed_ms=round(exp[len(exp) // 2], 3))
