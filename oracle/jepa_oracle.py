"""CPU oracle: a functional restatement of the WavJEPA pre-training step.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain PyTorch tensor ops on an explicit ``{name: tensor}`` parameter dict
(the reference's state_dict names), the algorithm of the reference hot path:

  * conv waveform encoder ............ reference wavjepa/extractors/audio_feature_extractor.py:54-138
  * feature LayerNorm + mapper + pos . reference wavjepa/jepa.py:391-396
  * post-norm transformer layer ...... torch nn.TransformerEncoderLayer (norm_first=False) as configured
                                       by reference wavjepa/types/wavjepa_configs.py:28-63
  * student encoder + gather ......... reference wavjepa/jepa.py:397-400,444-454
  * predictor ("decoder") ............ reference wavjepa/jepa.py:422-440
  * EMA teacher targets .............. reference wavjepa/jepa.py:230-270
  * masked MSE ....................... reference wavjepa/jepa.py:335-362
  * EMA schedule/update .............. reference wavjepa/jepa.py:186-198
  * crop + normalise ................. reference wavjepa/jepa.py:275-316
  * AdamW / clip / cosine warm-up .... reference wavjepa/jepa.py:215-228, train.py:177-178
                                       (torch.optim.AdamW, clip_grad_norm_, HF get_cosine_schedule_with_warmup)
  * sin-cos positions ................ reference wavjepa/pos_embed.py:75-93

Parity pinning: `tests/test_oracle_golden.py` checks this file against fixtures in `tests/golden/`
that were produced by running the *reference itself* in the build container
(`tests/golden/make_golden.py`, stub-import recipe in `tests/golden/_ref_import.py`).

Two numeric modes:
  ``fp32``  every op in float32 (tight yard-stick against the reference run in fp32);
  ``bf16``  explicit emulation of the dtype flow `torch.autocast("cuda", bfloat16)` produces on the
            reference (conv/linear/attention in bf16 with fp32 accumulation, group_norm / layer_norm /
            mse in fp32, fp32 residual stream).  This is the yard-stick for the HIP path.

The oracle is never on the product path and is never the thing measured except as `cpu_baseline`.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]
ConvSpec = Sequence[Tuple[int, int, int]]

WAVJEPA_CONV_SPEC: ConvSpec = [(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512, 2, 2)]


# --------------------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------------------
def _lin(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], mode: str) -> torch.Tensor:
    """nn.Linear.  bf16 mode: operands rounded to bf16, fp32 accumulate, bf16 result."""
    if mode == "bf16":
        return F.linear(x.to(torch.bfloat16), w.to(torch.bfloat16), None if b is None else b.to(torch.bfloat16))
    return F.linear(x.float(), w.float(), None if b is None else b.float())


def _ln(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float) -> torch.Tensor:
    """LayerNorm, always computed and returned in fp32 (autocast fp32 policy)."""
    return F.layer_norm(x.float(), (x.shape[-1],), w.float(), b.float(), eps)


def conv_token_count(n_samples: int, spec: ConvSpec) -> int:
    """Number of output frames of the un-padded conv chain (reference total_patches :140-145 by a dummy
    forward; here by the closed form floor((L-k)/s)+1 per layer)."""
    length = n_samples
    for _, k, s in spec:
        length = (length - k) // s + 1
    return length


def sincos_positions(dim: int, n_pos: int) -> torch.Tensor:
    """[1, n_pos, dim] fp32 table: sin half then cos half, omega_j = 10000^(-j/(dim/2)), float64 math."""
    assert dim % 2 == 0
    half = dim // 2
    omega = 1.0 / (10000.0 ** (np.arange(half, dtype=np.float64) / (dim / 2.0)))
    ang = np.arange(n_pos, dtype=np.float64)[:, None] * omega[None, :]
    tab = np.concatenate([np.sin(ang), np.cos(ang)], axis=1)
    return torch.from_numpy(tab).float().unsqueeze(0)


# --------------------------------------------------------------------------------------
# conv front-end
# --------------------------------------------------------------------------------------
def _conv_stack(P: Params, x: torch.Tensor, spec: ConvSpec, mode: str, stack: str) -> torch.Tensor:
    """One stacked-Conv1d encoder with parameters `{stack}{l}.0.weight`, `{stack}0.2.{weight,bias}`: [N, C_in, L] -> [N, T, C_out].

    Layer l: Conv1d(no bias) -> [GroupNorm(C, C) on layer 0 only] -> erf-GELU.
    bf16 mode: conv in/out bf16; GroupNorm computed/returned fp32; GELU keeps its input dtype.
    """
    for i, (dim, k, s) in enumerate(spec):
        w = P[f"{stack}{i}.0.weight"]
        if mode == "bf16":
            x = F.conv1d(x.to(torch.bfloat16), w.to(torch.bfloat16), stride=s)
        else:
            x = F.conv1d(x.float(), w.float(), stride=s)
        if i == 0:
            g, b = P[f"{stack}0.2.weight"], P[f"{stack}0.2.bias"]
            x = F.group_norm(x.float(), dim, g.float(), b.float(), 1e-5)
        x = F.gelu(x)
    return x.transpose(1, 2)


def conv_frontend(P: Params, audio: torch.Tensor, spec: ConvSpec, mode: str,
                  prefix: str = "extract_audio.") -> torch.Tensor:
    """audio [N, C_in, L] -> tokens.

    ConvFeatureExtractor (parameters `cnn.*`; reference extractors/audio_feature_extractor.py:124-138): one stack over all
    input channels -> [N, T, C_out].
    ConvChannelFeatureExtractor (parameters `cnns.{c}.*`; reference extractors/audio_channel_feature_extractor.py:154-179): every
    channel x[:, [c]] through its own MONO stack (or the single shared one), stacked and flattened channel-major
    "B (C S)" -> [N, C_in * T, C_out].
    """
    if f"{prefix}cnns.0.0.0.weight" in P:
        n_stacks = 1
        while f"{prefix}cnns.{n_stacks}.0.0.weight" in P:
            n_stacks += 1
        outs = [_conv_stack(P, audio[:, c:c + 1], spec, mode, f"{prefix}cnns.{min(c, n_stacks - 1)}.") for c in range(audio.shape[1])]
        return torch.stack(outs, dim=1).flatten(1, 2)
    return _conv_stack(P, audio, spec, mode, f"{prefix}cnn.")


def binaural_positions(dim: int, time_steps: int) -> torch.Tensor:
    """reference pos_embed.py:122-151 get_binaural_pos_embed: [2 * time_steps, dim] float64 table; first half of the features =
    sin-cos of the time step (dim / 2 wide), second half = 0 for the left channel and the sin-cos code of position 0
    (zeros then ones) for the right channel; left rows first."""
    assert dim % 2 == 0
    half = dim // 2

    def sincos(d, n):
        omega = 1.0 / (10000.0 ** (np.arange(d // 2, dtype=np.float64) / (d / 2.0)))
        ang = np.arange(n, dtype=np.float64)[:, None] * omega[None, :]
        return np.concatenate([np.sin(ang), np.cos(ang)], axis=1)

    time_embed = sincos(half, time_steps)
    left = np.concatenate([time_embed, np.zeros((time_steps, half))], axis=1)
    right = np.concatenate([time_embed, np.tile(sincos(half, 1), (time_steps, 1))], axis=1)
    return torch.from_numpy(np.concatenate([left, right], axis=0))


def local_features(P: Params, audio: torch.Tensor, spec: ConvSpec, mode: str) -> torch.Tensor:
    """conv tokens -> LayerNorm(eps 1e-5) -> Linear -> + fixed positions (fp32 result)."""
    x = conv_frontend(P, audio, spec, mode)
    x = _ln(x, P["feature_norms.weight"], P["feature_norms.bias"], 1e-5)
    if "post_extraction_mapper.weight" in P:
        x = _lin(x, P["post_extraction_mapper.weight"], P["post_extraction_mapper.bias"], mode)
    return x.float() + P["pos_encoding_encoder"].float()


# --------------------------------------------------------------------------------------
# transformer
# --------------------------------------------------------------------------------------
def attention(qkv: torch.Tensor, nhead: int, key_mask: Optional[torch.Tensor], mode: str) -> torch.Tensor:
    """qkv [B, T, 3D] (q | k | v packed) -> [B, T, D].  key_mask [B, T] bool, True = key not attended.

    Scores and softmax in fp32.  bf16 mode: probabilities rounded to bf16 before P@V, output bf16
    (the behaviour of fused flash-style kernels).
    """
    B, T, D3 = qkv.shape
    D = D3 // 3
    hd = D // nhead
    q, k, v = qkv.float().view(B, T, 3, nhead, hd).permute(2, 0, 3, 1, 4)  # each [B, H, T, hd]
    s = torch.matmul(q, k.transpose(-1, -2)) * (1.0 / math.sqrt(hd))
    if key_mask is not None:
        s = s.masked_fill(key_mask[:, None, None, :], float("-inf"))
    p = torch.softmax(s, dim=-1)
    if mode == "bf16":
        p = p.to(torch.bfloat16).float()
    o = torch.matmul(p, v).permute(0, 2, 1, 3).reshape(B, T, D)
    return o.to(torch.bfloat16) if mode == "bf16" else o


def post_norm_layer(P: Params, pre: str, x: torch.Tensor, nhead: int, key_mask: Optional[torch.Tensor],
                    mode: str, eps: float = 1e-6) -> torch.Tensor:
    """x = LN1(x + out_proj(attn(in_proj(x))));  x = LN2(x + linear2(gelu(linear1(x)))).  x is fp32."""
    qkv = _lin(x, P[pre + "self_attn.in_proj_weight"], P[pre + "self_attn.in_proj_bias"], mode)
    a = attention(qkv, nhead, key_mask, mode)
    sa = _lin(a, P[pre + "self_attn.out_proj.weight"], P[pre + "self_attn.out_proj.bias"], mode)
    x = _ln(x.float() + sa.float(), P[pre + "norm1.weight"], P[pre + "norm1.bias"], eps)
    h = _lin(x, P[pre + "linear1.weight"], P[pre + "linear1.bias"], mode)
    h = F.gelu(h)
    ff = _lin(h, P[pre + "linear2.weight"], P[pre + "linear2.bias"], mode)
    x = _ln(x.float() + ff.float(), P[pre + "norm2.weight"], P[pre + "norm2.bias"], eps)
    return x


def n_layers(P: Params, stack: str) -> int:
    i = 0
    while f"{stack}.layers.{i}.norm1.weight" in P:
        i += 1
    return i


def encoder_stack(P: Params, stack: str, x: torch.Tensor, nhead: int, key_mask: Optional[torch.Tensor],
                  mode: str, final_norm: bool = True, eps: float = 1e-6,
                  keep: Optional[List[torch.Tensor]] = None, keep_last: int = 0) -> torch.Tensor:
    L = n_layers(P, stack)
    for i in range(L):
        x = post_norm_layer(P, f"{stack}.layers.{i}.", x, nhead, key_mask, mode, eps)
        if keep is not None and L - i <= keep_last:
            keep.append(x)
    if final_norm:
        x = _ln(x, P[f"{stack}.norm.weight"], P[f"{stack}.norm.bias"], 1e-5)
    return x


def teacher_targets(P: Params, x: torch.Tensor, nhead: int, top_k: int, mode: str) -> torch.Tensor:
    """Teacher layers (no mask, no final norm), last `top_k` layer outputs, each normalised JOINTLY over
    (T, D) per sample (what F.instance_norm does on the reference's 4-D [k, B, D, T] tensor: biased
    variance, eps 1e-5 inside the sqrt), then averaged over layers."""
    kept: List[torch.Tensor] = []
    with torch.no_grad():
        encoder_stack(P, "teacher_encoder", x.detach(), nhead, None, mode, final_norm=False,
                      keep=kept, keep_last=top_k)
        if top_k <= 1:
            return kept[-1]
        acc = torch.zeros_like(kept[0], dtype=torch.float32)
        for y in kept:
            y = y.float()
            mu = y.mean(dim=(1, 2), keepdim=True)
            var = y.var(dim=(1, 2), unbiased=False, keepdim=True)
            acc += (y - mu) / torch.sqrt(var + 1e-5)
        return acc / len(kept)


def predictor(P: Params, ctx_feats: torch.Tensor, ctx_mask: torch.Tensor, vis_mask: torch.Tensor,
              nhead: int, mode: str) -> torch.Tensor:
    """ctx_feats [sum(~ctx_mask), Dd]; ctx_mask [B, T]; vis_mask [B, G, T] -> preds [(B G), T, De]."""
    B, T = ctx_mask.shape
    G = vis_mask.shape[1]
    Dd = ctx_feats.shape[-1]
    tgt = P["mask_token"].to(ctx_feats.dtype).expand(B, T, Dd).clone()
    tgt[~ctx_mask] = ctx_feats.reshape(-1, Dd)
    tgt = tgt.float() + P["pos_encoding_decoder"].float()
    tgt = tgt[:, None].expand(B, G, T, Dd).reshape(B * G, T, Dd)
    km = vis_mask.reshape(B * G, T)
    y = encoder_stack(P, "decoder", tgt, nhead, km, mode, final_norm=True)
    return _lin(y, P["decoder_to_encoder_mapper.weight"], P["decoder_to_encoder_mapper.bias"], mode)


def masked_mse(preds: torch.Tensor, targets: torch.Tensor, target_indices: torch.Tensor) -> torch.Tensor:
    """preds [(B G), T, D], targets [B, T, D], target_indices [B, G, T] bool.
    mean over D of squared error, summed over target positions, divided by (#targets + 1e-8).  fp32."""
    B, G, T = target_indices.shape
    D = preds.shape[-1]
    err = (preds.float().view(B, G, T, D) - targets.float()[:, None]) ** 2
    per_pos = err.mean(dim=-1) * target_indices
    return per_pos.sum() / (target_indices.sum() + 1e-8)


def jepa_forward(P: Params, audio: torch.Tensor, ctx_mask: torch.Tensor, target_indices: torch.Tensor,
                 vis_mask: torch.Tensor, *, spec: ConvSpec = WAVJEPA_CONV_SPEC, enc_heads: int = 12,
                 dec_heads: int = 12, top_k: int = 8, mode: str = "fp32") -> Dict[str, torch.Tensor]:
    """The whole forward of reference JEPA.forward (jepa.py:365-419)."""
    lf = local_features(P, audio, spec, mode)
    ctx = encoder_stack(P, "encoder", lf, enc_heads, ctx_mask, mode, final_norm=True)
    gathered = ctx[~ctx_mask]  # row-major (b, t) order; pure copy
    cf = _lin(gathered, P["encoder_to_decoder_mapper.weight"], P["encoder_to_decoder_mapper.bias"], mode)
    preds = predictor(P, cf, ctx_mask, vis_mask, dec_heads, mode)
    targets = teacher_targets(P, lf, enc_heads, top_k, mode)
    loss = masked_mse(preds, targets, target_indices)
    return dict(local_features=lf, contextual_features=cf, loss=loss, preds=preds, targets=targets)


def audio_representation(P: Params, audio: torch.Tensor, padding_mask: Optional[torch.Tensor], *,
                         spec: ConvSpec = WAVJEPA_CONV_SPEC, enc_heads: int = 12, mode: str = "fp32") -> torch.Tensor:
    """Inference path of reference JEPA.get_audio_representation (jepa.py:456-467): student only."""
    with torch.no_grad():
        lf = local_features(P, audio, spec, mode)
        return encoder_stack(P, "encoder", lf, enc_heads, padding_mask, mode, final_norm=True)


# --------------------------------------------------------------------------------------
# batch preparation (reference on_after_batch_transfer, jepa.py:275-316)
# --------------------------------------------------------------------------------------
def crop_normalize(source: torch.Tensor, starts: torch.Tensor, length: int, perm: Optional[torch.Tensor] = None,
                   to_bf16: bool = True) -> torch.Tensor:
    """source [B, C, L_full] fp32, starts [B, S] -> [B*S, C, length].
    Per crop: (x - mean) / (unbiased_std + 1e-5) over (C, length); cast to bf16; flatten; optional row
    permutation (the reference shuffles the audio rows only)."""
    B, C, _ = source.shape
    S = starts.shape[1]
    out = torch.empty(B, S, C, length, dtype=torch.float32)
    for b in range(B):
        for s in range(S):
            st = int(starts[b, s])
            out[b, s] = source[b, :, st:st + length]
    mean = out.mean(dim=(-2, -1), keepdim=True)
    std = out.std(dim=(-2, -1), keepdim=True)
    out = (out - mean) / (std + 1e-5)
    if to_bf16:
        out = out.to(torch.bfloat16)
    out = out.flatten(0, 1)
    if perm is not None:
        out = out[perm]
    return out


# --------------------------------------------------------------------------------------
# EMA, optimiser, schedule
# --------------------------------------------------------------------------------------
def ema_decay(step: int, start: float = 0.999, end: float = 0.99999, anneal_end_step: int = 100000) -> float:
    if step >= anneal_end_step:
        return end
    return end - (end - start) * (1 - step / anneal_end_step)


def ema_update(P: Params, r: float) -> None:
    """teacher <- r * teacher + (1 - r) * student for every tensor of `encoder.*` (final norm included)."""
    with torch.no_grad():
        for name in list(P.keys()):
            if name.startswith("encoder."):
                t = P["teacher_" + name]
                t.mul_(r).add_((1 - r) * P[name].detach())


def lr_lambda(step: int, warmup: int, total: int) -> float:
    """HF get_cosine_schedule_with_warmup (num_cycles 0.5)."""
    if step < warmup:
        return step / max(1, warmup)
    progress = (step - warmup) / max(1, total - warmup)
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * progress)))


def trainable_names(P: Params) -> List[str]:
    return [n for n in P if not n.startswith("teacher_encoder.") and not n.startswith("pos_encoding_")]


def clip_grad_norm(grads: Dict[str, torch.Tensor], max_norm: float) -> float:
    total = math.sqrt(sum(float(g.float().pow(2).sum()) for g in grads.values()))
    coef = min(1.0, max_norm / (total + 1e-6))
    for g in grads.values():
        g.mul_(coef)
    return total


def adamw_update(P: Params, grads: Dict[str, torch.Tensor], state: Dict[str, Dict[str, torch.Tensor]], step_no: int,
                 lr: float, betas=(0.9, 0.98), eps: float = 1e-6, weight_decay: float = 0.04) -> None:
    """torch.optim.AdamW (decoupled decay, bias-corrected).  step_no is 1-based."""
    b1, b2 = betas
    bc1 = 1 - b1 ** step_no
    bc2 = 1 - b2 ** step_no
    with torch.no_grad():
        for n, g in grads.items():
            st = state.setdefault(n, dict(m=torch.zeros_like(P[n]), v=torch.zeros_like(P[n])))
            p = P[n]
            p.mul_(1 - lr * weight_decay)
            st["m"].mul_(b1).add_(g, alpha=1 - b1)
            st["v"].mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = (st["v"].sqrt() / math.sqrt(bc2)).add_(eps)
            p.addcdiv_(st["m"], denom, value=-lr / bc1)


def train_step(P: Params, opt_state: Dict[str, Dict[str, torch.Tensor]], step: int, batch, *, lr: float = 4e-4,
               warmup: int = 100000, total_steps: int = 375000, betas=(0.9, 0.98), eps: float = 1e-6,
               weight_decay: float = 0.04, clip: float = 5.0, ema=(0.999, 0.99999, 100000), mode: str = "fp32",
               **fw) -> Dict[str, float]:
    """One optimisation step in the reference's order:
    forward -> EMA with the pre-update student -> backward -> clip(5) -> AdamW at lr*lambda(step) ."""
    audio, ctx, tgt, vis = batch
    names = trainable_names(P)
    for n in names:
        P[n].requires_grad_(True)
        P[n].grad = None
    out = jepa_forward(P, audio, ctx, tgt, vis, mode=mode, **fw)
    r = ema_decay(step, *ema)
    ema_update(P, r)
    out["loss"].backward()
    grads = {n: P[n].grad.detach().clone() for n in names if P[n].grad is not None}
    for n in names:
        P[n].requires_grad_(False)
        P[n].grad = None
    gnorm = clip_grad_norm(grads, clip)
    adamw_update(P, grads, opt_state, step + 1, lr * lr_lambda(step, warmup, total_steps), betas, eps, weight_decay)
    return dict(loss=float(out["loss"].detach()), grad_norm=gnorm, ema=r)
