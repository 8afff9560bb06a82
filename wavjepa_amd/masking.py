"""Mask makers with the reference's call contract: masker(batch_size=, n_times=, in_channels=) ->
(ctx_mask [B,T] True = NOT context, target [B,G,T], visible_mask [B,G,T] = ctx XOR target)
(reference wavjepa/masking.py:7-128 TimeInverseBlockMasker, :131-207 SpeechMasker)."""
from __future__ import annotations

import torch
from torch import nn

from .audio_masking import compute_mask_indices


def _channel_repeat(ctx: torch.Tensor, tgt: torch.Tensor, vis: torch.Tensor, in_channels: int, channel_major: bool = False):
    """Repeat the per-time masks for every channel and flatten.
    Default = the reference's order (masking.py:120-126): time-major interleave "B (S C)".  Note that the reference's own
    ConvChannelFeatureExtractor flattens its tokens channel-major "B (C S)" (audio_channel_feature_extractor.py:176-178), so with
    the default order token j and mask entry j refer to DIFFERENT (channel, time) pairs (a drift in the reference,
    kept bit for bit for drop-in parity).  channel_major=True emits "B (C S)" instead: entry c*S + s masks time s of channel c,
    matching the extractor's token order."""
    if channel_major:
        ctx = ctx[:, None, :].expand(-1, in_channels, -1).reshape(ctx.shape[0], -1)
        tgt = tgt[:, :, None, :].expand(-1, -1, in_channels, -1).reshape(tgt.shape[0], tgt.shape[1], -1)
        vis = vis[:, :, None, :].expand(-1, -1, in_channels, -1).reshape(vis.shape[0], vis.shape[1], -1)
        return ctx, tgt, vis
    ctx = ctx[:, :, None].expand(-1, -1, in_channels).reshape(ctx.shape[0], -1)
    tgt = tgt[:, :, :, None].expand(-1, -1, -1, in_channels).reshape(tgt.shape[0], tgt.shape[1], -1)
    vis = vis[:, :, :, None].expand(-1, -1, -1, in_channels).reshape(vis.shape[0], vis.shape[1], -1)
    return ctx, tgt, vis


class TimeInverseBlockMasker(nn.Module):
    def __init__(self, target_masks_per_context: int = 4, context_mask_prob: float = 0.3, context_mask_length: int = 10,
                 target_prob: float = 0.2, target_length: int = 20, ratio_cutoff: float = 0.05,
                 channel_based_masking: bool = False, channel_major: bool = False, **kwargs):
        super().__init__()
        self.target_masks_per_context = target_masks_per_context
        self.context_mask_prob = context_mask_prob
        self.context_mask_length = context_mask_length
        self.target_prob = target_prob
        self.target_length = target_length
        self.ratio_cutoff = ratio_cutoff
        self.channel_based_masking = channel_based_masking
        self.channel_major = channel_major      # build-side option, see _channel_repeat

    def forward(self, batch_size: int, n_times: int, in_channels: int):
        T = n_times // in_channels
        G = self.target_masks_per_context
        target = torch.zeros(batch_size, G, T, dtype=torch.bool)
        context = torch.zeros(batch_size, T, dtype=torch.bool)
        for b in range(batch_size):
            tg = torch.zeros(G, T, dtype=torch.bool)
            while True:
                ctx = ~compute_mask_indices((1, T), None, self.context_mask_prob, self.context_mask_length)
                for g in range(G):
                    tg[g] = compute_mask_indices((1, T), None, self.target_prob, self.target_length)
                ctx = ctx & ~tg.any(dim=0)
                if ctx.sum() / T >= self.ratio_cutoff:
                    break
            target[b], context[b] = tg, ctx
        ctx_mask = ~context
        vis = torch.logical_xor(ctx_mask.unsqueeze(1), target)
        if self.channel_based_masking:
            ctx_mask, target, vis = _channel_repeat(ctx_mask, target, vis, in_channels, self.channel_major)
        return ctx_mask, target, vis.to(torch.bool)


class SpeechMasker(nn.Module):
    def __init__(self, target_masks_per_context: int = 4, target_prob: float = 0.25, target_length: int = 5,
                 ratio_cutoff: float = 0.3, min_context_len: int = 5, channel_based_masking: bool = False,
                 channel_major: bool = False, **kwargs):
        super().__init__()
        self.target_masks_per_context = target_masks_per_context
        self.target_prob = target_prob
        self.target_length = target_length
        self.ratio_cutoff = ratio_cutoff
        self.min_context_len = min_context_len
        self.channel_based_masking = channel_based_masking
        self.channel_major = channel_major

    def filter_small_clusters(self, mask: torch.Tensor) -> torch.Tensor:
        """Context runs shorter than min_context_len are dropped."""
        m = mask.numpy().copy()
        n, i = len(m), 0
        while i < n:
            j = i
            while j < n and m[j] == m[i]:
                j += 1
            if m[i] and (j - i) < self.min_context_len:
                m[i:j] = False
            i = j
        return torch.from_numpy(m)

    def forward(self, batch_size: int, n_times: int, in_channels: int):
        T = n_times // in_channels
        G = self.target_masks_per_context
        target = torch.zeros(batch_size, G, T, dtype=torch.bool)
        context = torch.ones(batch_size, T, dtype=torch.bool)
        for b in range(batch_size):
            while True:
                tg = torch.zeros(G, T, dtype=torch.bool)
                for g in range(G):
                    tg[g] = compute_mask_indices((1, T), None, self.target_prob, self.target_length)
                ctx = self.filter_small_clusters(~tg.any(dim=0))
                if ctx.sum() / T >= self.ratio_cutoff:
                    break
            target[b], context[b] = tg, ctx
        ctx_mask = ~context
        vis = torch.logical_xor(ctx_mask.unsqueeze(1), target)
        if self.channel_based_masking:
            ctx_mask, target, vis = _channel_repeat(ctx_mask, target, vis, in_channels, self.channel_major)
        return ctx_mask, target, vis.to(torch.bool)
