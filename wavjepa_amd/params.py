"""Flat parameter storage for the MI355X step.

All trainable parameters live in ONE fp32 buffer (plus one fp32 gradient buffer, AdamW moments and a bf16 shadow
copy at the same offsets); the EMA teacher lives in a second flat buffer laid out exactly like the student
encoder's slice.  The module's nn.Parameters are re-pointed to views of these buffers, so `state_dict()` keeps the
reference's names/shapes while EMA, grad-norm, AdamW and the RCCL all-reduce each run as a single
kernel / a few large collectives over contiguous memory (sized for 288 GB HBM, not for per-tensor launches).
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Tuple

import torch
from torch import nn

ALIGN = 8  # elements: keeps every tensor 32-B (fp32) / 16-B (bf16) aligned for vector loads


class Slot:
    __slots__ = ("name", "offset", "numel", "shape")

    def __init__(self, name: str, offset: int, numel: int, shape: Tuple[int, ...]):
        self.name, self.offset, self.numel, self.shape = name, offset, numel, shape


def _layout(named: Iterable[Tuple[str, torch.Tensor]]) -> Tuple[List[Slot], int]:
    slots, off = [], 0
    for name, p in named:
        slots.append(Slot(name, off, p.numel(), tuple(p.shape)))
        off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
    return slots, off


class FlatParams:
    """Owns the flat buffers and the name -> slot maps for one JEPA module."""

    def __init__(self, module: nn.Module, device: torch.device):
        self.device = device
        trainable = [(n, p) for n, p in module.named_parameters() if p.requires_grad]
        teacher = [(n, p) for n, p in module.named_parameters() if n.startswith("teacher_encoder.")]
        self.slots, self.n = _layout(trainable)
        self.tslots, self.tn = _layout(teacher)
        self.by_name: Dict[str, Slot] = {s.name: s for s in self.slots}
        self.tby_name: Dict[str, Slot] = {s.name: s for s in self.tslots}

        self.p32 = torch.zeros(self.n, dtype=torch.float32, device=device)
        self.g32 = torch.zeros(self.n, dtype=torch.float32, device=device)
        self.p16 = torch.zeros(self.n, dtype=torch.bfloat16, device=device)
        self.t32 = torch.zeros(self.tn, dtype=torch.float32, device=device)
        self.t16 = torch.zeros(self.tn, dtype=torch.bfloat16, device=device)
        self.adam_m = None
        self.adam_v = None

        params = dict(module.named_parameters())
        self._grad_views: List[Tuple[nn.Parameter, torch.Tensor]] = []
        with torch.no_grad():
            for s in self.slots:
                p = params[s.name]
                view = self.p32[s.offset:s.offset + s.numel].view(s.shape)
                view.copy_(p.data.to(device=device, dtype=torch.float32))
                p.data = view
                gview = self.g32[s.offset:s.offset + s.numel].view(s.shape)
                p.grad = gview
                self._grad_views.append((p, gview))
            for s in self.tslots:
                p = params[s.name]
                view = self.t32[s.offset:s.offset + s.numel].view(s.shape)
                view.copy_(p.data.to(device=device, dtype=torch.float32))
                p.data = view
        # the student-encoder slice must mirror the teacher layout 1:1 (single-kernel EMA over contiguous memory)
        enc = [s for s in self.slots if s.name.startswith("encoder.")]
        self.enc_offset = enc[0].offset
        if self.tslots:                  # (the denoiser stage has no EMA copy inside the module: its teacher is a frozen JEPA)
            assert len(enc) == len(self.tslots), "student encoder / teacher parameter lists differ"
            for s, t in zip(enc, self.tslots):
                assert "teacher_" + s.name == t.name and s.offset - self.enc_offset == t.offset and s.numel == t.numel
        self.enc_numel = self.tn
        self.bf16_fresh = False

    # -- pointers -----------------------------------------------------------------------------------------------
    def front_and_rest_ranges(self):
        """[lo, hi) element ranges of the flat buffers: `front` = what the forward needs before its first transformer kernel (mask token, conv
        extractor, feature norm, post-extraction mapper), `rest` = the transformer stacks and their mappers; both padded to the 8-element
        slot granularity, together exactly [0, n)."""
        if getattr(self, "_fr", None) is None:
            is_front = lambda nm: nm.startswith(("mask_token", "extract_audio.", "feature_norms.", "post_extraction_mapper."))
            runs = []
            for sl in sorted(self.slots, key=lambda t: t.offset):
                lo, hi, f = sl.offset, sl.offset + (sl.numel + 7) // 8 * 8, is_front(sl.name)
                if runs and runs[-1][2] == f and runs[-1][1] == lo:
                    runs[-1][1] = hi
                else:
                    runs.append([lo, hi, f])
            assert runs and runs[0][0] == 0 and runs[-1][1] == self.n and all(a[1] == b[0] for a, b in zip(runs, runs[1:]))
            self._fr = ([(lo, hi) for lo, hi, f in runs if f], [(lo, hi) for lo, hi, f in runs if not f])
        return self._fr

    def ptr32(self, name: str) -> int:
        return self.p32.data_ptr() + 4 * self.by_name[name].offset

    def ptr16(self, name: str) -> int:
        return self.p16.data_ptr() + 2 * self.by_name[name].offset

    def gptr(self, name: str) -> int:
        return self.g32.data_ptr() + 4 * self.by_name[name].offset

    def tptr32(self, name: str) -> int:
        return self.t32.data_ptr() + 4 * self.tby_name[name].offset

    def tptr16(self, name: str) -> int:
        return self.t16.data_ptr() + 2 * self.tby_name[name].offset

    def owns(self, module: nn.Module) -> bool:
        """True while the module's parameters still alias the flat buffers (e.g. not after .to()/.cuda())."""
        # (two direct look-ups: building dict(named_parameters()) walked all 457 tensors, ~1 ms of host time, five times per step)
        s = self.slots[-1]
        if module.get_parameter(s.name).data_ptr() != self.p32.data_ptr() + 4 * s.offset:
            return False
        if not self.tslots:
            return True
        t = self.tslots[-1]
        return module.get_parameter(t.name).data_ptr() == self.t32.data_ptr() + 4 * t.offset

    def attach_grads(self) -> None:
        """(Re-)expose the flat gradient buffer as every parameter's .grad (zero_grad(set_to_none) drops them)."""
        for p, g in self._grad_views:
            p.grad = g

    def ensure_adam_state(self) -> None:
        if self.adam_m is None:
            self.adam_m = torch.zeros_like(self.p32)
            self.adam_v = torch.zeros_like(self.p32)

    def bucket_bounds(self, n_buckets: int) -> List[Tuple[int, int]]:
        """Split [0, n) into ~equal contiguous buckets on slot boundaries (for bucketed all-reduce)."""
        target = self.n / max(1, n_buckets)
        bounds, start = [], 0
        for s in self.slots:
            end = s.offset + (s.numel + ALIGN - 1) // ALIGN * ALIGN
            if end - start >= target and len(bounds) < n_buckets - 1:
                bounds.append((start, end))
                start = end
        if start < self.n:
            bounds.append((start, self.n))
        return bounds
