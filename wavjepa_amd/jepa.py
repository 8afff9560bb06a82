"""JEPA module for MI355X: the reference's LightningModule surface over the HIP engine.

Keeps the constructor, hooks, attributes, `ForwardReturn` keys and state_dict names/shapes of
reference wavjepa/jepa.py:24-467 (so reference checkpoints load and callers such as train.py / hear_api need no
change), but every tensor op of the step runs in `libwavjepa_hip.so` through `wavjepa_amd.engine`:
the nn.Modules below only hold parameters.  There is no PyTorch-op fallback; without the HIP library or a GPU the
compute entry points raise.
"""
from __future__ import annotations

import copy
import math
import os
from typing import Any, Dict, Optional

import numpy as np
import torch
from torch import nn

from . import ops
from .engine import EngineConfig, JepaEngine, MaskPlan, make_mask_plan, pack_upload
from .extractors.audio_extractor import Extractor
from .functions import trunc_normal_
from .params import FlatParams
from .pos_embed import get_1d_sincos_pos_embed_from_grid
from .types import ForwardReturn, TransformerEncoderCFG, TransformerLayerCFG


def collate_fn(batch: torch.Tensor) -> torch.Tensor:
    return batch.flatten(start_dim=0, end_dim=1)


class _AttrDict(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


# ---------------------------------------------------------------------------------------------------------------------
# Parameter containers with nn.TransformerEncoder's state_dict names (…layers.{i}.self_attn.in_proj_weight, …)
# ---------------------------------------------------------------------------------------------------------------------
class _SelfAttention(nn.Module):
    def __init__(self, d: int, bias: bool = True):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d, d))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d))
        self.out_proj = nn.Linear(d, d, bias=bias)
        nn.init.xavier_uniform_(self.in_proj_weight)          # nn.MultiheadAttention._reset_parameters
        nn.init.constant_(self.out_proj.bias, 0.0)


class _PostNormLayer(nn.Module):
    def __init__(self, d_model: int, nhead: int, dim_feedforward: int, layer_norm_eps: float, **unused):
        super().__init__()
        self.self_attn = _SelfAttention(d_model)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model, eps=layer_norm_eps)
        self.norm2 = nn.LayerNorm(d_model, eps=layer_norm_eps)
        self.nhead = nhead


class TransformerStack(nn.Module):
    """`num_layers` deep copies of one post-norm layer + final LayerNorm (as nn.TransformerEncoder(layer, norm=...))."""

    def __init__(self, layer_cfg: Dict[str, Any], num_layers: int):
        super().__init__()
        if layer_cfg.get("norm_first", False):
            raise NotImplementedError("the WavJEPA path is post-norm (norm_first=False)")
        if layer_cfg.get("dropout", 0.0) != 0.0:
            raise NotImplementedError("dropout is 0 on the WavJEPA path")
        if int(layer_cfg["dim_feedforward"]) != 4 * int(layer_cfg["d_model"]):
            raise NotImplementedError("feed-forward width must be 4 * d_model")
        first = _PostNormLayer(**layer_cfg)
        self.layers = nn.ModuleList([first] + [copy.deepcopy(first) for _ in range(num_layers - 1)])
        self.norm = nn.LayerNorm(layer_cfg["d_model"])
        self.d_model, self.nhead, self.num_layers = int(layer_cfg["d_model"]), int(layer_cfg["nhead"]), num_layers
        self.layer_norm_eps = float(layer_cfg["layer_norm_eps"])


# ---------------------------------------------------------------------------------------------------------------------
# Autograd bridge: loss.backward() runs the engine's hand-written backward into the flat gradient buffer
# ---------------------------------------------------------------------------------------------------------------------
class _StepOutputs(dict):
    """The reference's ForwardReturn (a plain dict).  `preds` in the reference's dense [N*G, T, D] shape is materialised
    on first access: the training loop never reads it, and on a ragged step the engine holds only the visible rows
    (the others, which carry zero loss weight at reference jepa.py:356, read 0)."""

    def __init__(self, engine, **items):
        super().__init__(**items)
        self._engine, self._plan = engine, engine.plan

    def __missing__(self, key):
        if key != "preds":
            raise KeyError(key)
        if self._engine.plan is not self._plan:
            raise RuntimeError("preds belongs to an earlier step: read it before the next forward overwrites the arena")
        value = self._engine.dense_preds()
        self[key] = value
        return value

    def __contains__(self, key):
        return key == "preds" or super().__contains__(key)

    def get(self, key, default=None):
        return self[key] if key in self else default


class _EngineLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor: torch.Tensor, module: "JEPA") -> torch.Tensor:
        ctx.module = module
        return module._engine.loss[0].clone()

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor):
        m = ctx.module
        g = grad_out.contiguous().float()
        m._engine.backward(gscale_ptr=g.data_ptr(), on_grads_ready=m._grads_ready_hook)
        m._flat.attach_grads()
        return torch.zeros_like(m._anchor), None


_ADAMW_SIDE = os.environ.get("WJ_ADAMW_SIDE", "1") != "0"
# workgroups of the overlapped update: one per CU -- it should take the HBM the front-end kernels leave idle, not their CU slots
# (a full grid of 8192: 45.68 ms/step, 1024: 45.64, 512: 45.48, 256: 45.40, interleaved on one box; 128 / 192 / 384 within 0.1 of 256)
_ADAMW_SIDE_WGS = int(os.environ.get("WJ_ADAMW_SIDE_WGS", "256"))
_ADAMW_ZERO_GRAD = os.environ.get("WJ_ADAMW_ZERO_GRAD", "1") != "0"


class FusedAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW semantics (reference jepa.py:215-222) as ONE kernel over the flat parameter buffer, with the
    global-norm clip of Lightning's gradient_clip_val (train.py:177-178) fused in (`max_grad_norm`)."""

    def __init__(self, module: "JEPA", lr: float, betas=(0.9, 0.98), eps: float = 1e-6, weight_decay: float = 0.01,
                 max_grad_norm: float = 0.0):
        params = [p for p in module.parameters() if p.requires_grad]
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._module = module
        self.max_grad_norm = max_grad_norm
        self._t = 0
        self._sumsq = None
        self._ws = None
        # overlap_next_forward (set by the training loop, trainer.StepRunner; WJ_ADAMW_SIDE=0 keeps it off): only the conv extractor, the
        # feature norm and the post-extraction mapper are updated on the compute stream; the transformer stacks (99 % of the parameters,
        # ~0.65 ms of streaming) go to the engine's side stream, where they run beside the NEXT step's crop / conv0 / conv stack -- matrix-
        # and VALU-bound kernels that leave the HBM idle, on a side stream that has nothing to do until the conv features exist.  The
        # engine's forward waits for the event before its first transformer kernel (JepaEngine.wait_optimizer); so do state_dict, the EMA
        # and inference.  Off by default: a caller that reads parameters right behind step() on its own stream would race the update.
        # With it on, readers that do NOT go through the engine -- a submodule's state_dict (model.encoder.state_dict()), copy.deepcopy,
        # direct p.data reads -- must call model._engine.wait_optimizer() first; JEPA.state_dict / load_state_dict / _apply (.cpu(), .to())
        # and Trainer.fit's return do so themselves.
        self.overlap_next_forward = False
        # fuse_zero_grad (training loop only; WJ_ADAMW_ZERO_GRAD=0 keeps it off): the update clears the gradient it has just read
        # (wj_adamw_args.zero_grad), so the next backward starts from a clean flat buffer without its own 444-MB fill on the critical path.
        # p.grad then reads 0 after step() -- what zero_grad() leaves; a caller that inspects gradients after step() keeps it off.
        self.fuse_zero_grad = False

    def _ensure_scratch(self, flat) -> None:
        if self._sumsq is None:
            self._sumsq = torch.zeros(1, dtype=torch.float32, device=flat.device)
            self._ws = torch.empty(1024, dtype=torch.float32, device=flat.device)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:          # Lightning's automatic optimisation hands training_step + backward over as a closure
            with torch.enable_grad():
                loss = closure()
        m = self._module
        m._ensure_engine()
        flat = m._flat
        flat.ensure_adam_state()
        self._ensure_scratch(flat)
        g = self.param_groups[0]
        self._t += 1
        if self.max_grad_norm > 0:
            ops.grad_sumsq(flat.g32, self._sumsq, self._ws, flat.n)
        zg = bool(self.fuse_zero_grad) and _ADAMW_ZERO_GRAD
        kw = dict(lr=float(g["lr"]), beta1=g["betas"][0], beta2=g["betas"][1], eps=g["eps"], weight_decay=g["weight_decay"], step=self._t,
                  max_norm=self.max_grad_norm, sumsq=self._sumsq if self.max_grad_norm > 0 else None)

        def update(lo: int, hi: int, workgroups: int = 0) -> None:
            ops.adamw_step(flat.p32.data_ptr() + 4 * lo, flat.g32.data_ptr() + 4 * lo, flat.adam_m.data_ptr() + 4 * lo,
                           flat.adam_v.data_ptr() + 4 * lo, hi - lo, p_bf16=flat.p16.data_ptr() + 2 * lo, workgroups=workgroups, zero_grad=zg, **kw)
        eng = m._engine
        front, rest = flat.front_and_rest_ranges()
        if self.overlap_next_forward and eng.use_side and rest and _ADAMW_SIDE:
            for lo, hi in front:
                update(lo, hi)
            eng.optimizer_on_side(lambda: [update(lo, hi, _ADAMW_SIDE_WGS) for lo, hi in rest])
        else:
            update(0, flat.n)
        flat.g_clean = zg            # engine.backward skips its own clear of the flat gradient buffer
        m._student_bf16_fresh = True
        return loss

    def grad_norm(self) -> torch.Tensor:
        """Global L2 norm of the last step's gradients (device scalar; no sync)."""
        return self._sumsq.sqrt() if self._sumsq is not None else torch.zeros(1)

    def zero_grad(self, set_to_none: bool = True) -> None:  # the engine overwrites the flat gradient buffer each backward
        return None

    def state_dict(self):
        flat = self._module._flat
        if self._module._engine is not None:
            self._module._engine.wait_optimizer()
        return dict(step=self._t, lr=self.param_groups[0]["lr"], m=None if flat is None or flat.adam_m is None else flat.adam_m.cpu(),
                    v=None if flat is None or flat.adam_v is None else flat.adam_v.cpu())

    def load_state_dict(self, sd):
        self._t = int(sd["step"])
        if sd.get("lr") is not None:     # LambdaLR.load_state_dict does not write the group lr back: without this the first
            self.param_groups[0]["lr"] = float(sd["lr"])   # resumed step would run at the constructor's lr
        self._module._ensure_engine().wait_optimizer()
        flat = self._module._flat
        flat.ensure_adam_state()
        if sd.get("m") is not None:
            flat.adam_m.copy_(sd["m"])
            flat.adam_v.copy_(sd["v"])


def cosine_schedule_with_warmup(optimizer, num_warmup_steps: int, num_training_steps: int):
    """HF get_cosine_schedule_with_warmup (num_cycles=0.5), which the reference calls at jepa.py:224-225."""
    def lr_lambda(step: int) -> float:
        if step < num_warmup_steps:
            return float(step) / float(max(1, num_warmup_steps))
        progress = float(step - num_warmup_steps) / float(max(1, num_training_steps - num_warmup_steps))
        return max(0.0, 0.5 * (1.0 + math.cos(math.pi * progress)))

    return torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda)


class _NullTrainer:
    max_steps = 375000


# The reference's JEPA is a pytorch_lightning.LightningModule (reference jepa.py:24).  When Lightning is importable this class
# derives from it too, so pl.Trainer.fit(model, datamodule) accepts it (hooks: on_after_batch_transfer, training_step,
# configure_optimizers; `hparams` through save_hyperparameters; `global_step` / `trainer` / `device` / `log_dict` are Lightning's
# own).  Without Lightning (this image ships none) it is a plain nn.Module that provides those attributes itself and
# wavjepa_amd.trainer.Trainer drives it.
try:                                         # pragma: no cover - depends on the environment
    import pytorch_lightning as _pl
except Exception:                            # noqa: BLE001
    try:
        import lightning.pytorch as _pl
    except Exception:                        # noqa: BLE001
        _pl = None
HAS_LIGHTNING = _pl is not None
_ModuleBase = _pl.LightningModule if HAS_LIGHTNING else nn.Module


class JEPA(_ModuleBase):
    """Joint-Embedding Predictive Architecture for waveforms (student encoder + predictor vs EMA teacher)."""

    def __init__(self, feature_extractor: Extractor, transformer_encoder_layers_cfg: TransformerLayerCFG,
                 transformer_encoder_cfg: TransformerEncoderCFG, transformer_decoder_layers_cfg: TransformerLayerCFG,
                 transformer_decoder_cfg: TransformerEncoderCFG, decoder_embedding_dim: int = 512, loss_fn: Optional[nn.Module] = None,
                 lr: float = 0.0002, adam_betas=(0.9, 0.98), adam_eps: float = 1e-06, adam_weight_decay: float = 0.01,
                 ema_decay: float = 0.999, ema_end_decay: float = 0.99999, ema_anneal_end_step: int = 100000,
                 average_top_k_layers: int = 12, resample_sr: int = 16000, process_audio_seconds: float = 2.00,
                 nr_samples_per_audio: int = 16, use_gradient_checkpointing: bool = False, compile_modules: bool = False,
                 size: str = "base", warmup_steps: int = 100000, **kwargs: Any):
        super().__init__()
        self.sr = resample_sr
        self.nr_samples_per_audio = nr_samples_per_audio
        self.ema_end_step = ema_anneal_end_step
        self.target_length = int(resample_sr * process_audio_seconds)
        self.total_patches = feature_extractor.total_patches(self.target_length)
        self.use_compiled_forward = False            # there is no tracing compiler on this path: kernels are hand-written
        self.use_gradient_checkpointing = False      # the reference's flag is off by default and buggy (drops the mask)
        if HAS_LIGHTNING:        # as the reference does (jepa.py:105-107): hyper-parameters into the checkpoint
            self.save_hyperparameters(ignore=["feature_encoder", "feature_extractor", "loss_fn"])
            self.hparams["adam_betas"] = tuple(adam_betas)
        else:
            self.hparams = _AttrDict(lr=lr, adam_betas=tuple(adam_betas), adam_eps=adam_eps, adam_weight_decay=adam_weight_decay,
                                     ema_decay=ema_decay, ema_end_decay=ema_end_decay, ema_anneal_end_step=ema_anneal_end_step,
                                     average_top_k_layers=average_top_k_layers, resample_sr=resample_sr,
                                     process_audio_seconds=process_audio_seconds, nr_samples_per_audio=nr_samples_per_audio,
                                     compile_modules=compile_modules, size=size, warmup_steps=warmup_steps,
                                     decoder_embedding_dim=decoder_embedding_dim)
            self.global_step = 0
            self.trainer = _NullTrainer()
        self.extract_audio = feature_extractor
        self.feature_norms = nn.LayerNorm(self.extract_audio.embedding_dim)
        self.loss_fn = loss_fn

        enc_l, enc_c = dict(transformer_encoder_layers_cfg), dict(transformer_encoder_cfg)
        dec_l, dec_c = dict(transformer_decoder_layers_cfg), dict(transformer_decoder_cfg)
        if size == "large":    # ViT-Large student (reference jepa.py:114-118)
            enc_l.update(nhead=16, d_model=1024, dim_feedforward=4096)
            enc_c.update(num_layers=24)
        elif size == "tiny":
            # BASELINE config 1 (a build-defined size, the reference knows base / large only): 2-layer student d = 128,
            # 2-layer predictor d = 64 with 4 heads (head dim 16: wj_attn_* run it in their 32-wide geometry).
            enc_l.update(nhead=4, d_model=128, dim_feedforward=512)
            enc_c.update(num_layers=2)
            dec_l.update(nhead=4, d_model=64, dim_feedforward=256)
            dec_c.update(num_layers=2)
            self.hparams["average_top_k_layers"] = min(int(average_top_k_layers), 2)
        self.n_encoder_heads = enc_l["nhead"]
        self.encoder_embedding_dim = enc_l["d_model"]
        self.n_decoder_heads = dec_l["nhead"]
        self.decoder_embedding_dim = dec_l["d_model"]          # the ctor's decoder_embedding_dim is ignored upstream too

        self.encoder = TransformerStack(enc_l, enc_c["num_layers"])
        c_feat = feature_extractor.embedding_dim
        self.post_extraction_mapper = nn.Linear(c_feat, self.encoder_embedding_dim) if c_feat != self.encoder_embedding_dim else None
        self.decoder = TransformerStack(dec_l, dec_c["num_layers"])
        self.decoder_to_encoder_mapper = nn.Linear(self.decoder_embedding_dim, self.encoder_embedding_dim, bias=True)
        self.encoder_to_decoder_mapper = nn.Linear(self.encoder_embedding_dim, self.decoder_embedding_dim)
        self.mask_token = nn.Parameter(torch.zeros(1, 1, self.decoder_embedding_dim))
        nn.init.normal_(self.mask_token, std=0.02)
        self.pos_encoding_encoder = self._get_pos_embed_params(self.encoder_embedding_dim)
        self.pos_encoding_decoder = self._get_pos_embed_params(self.decoder_embedding_dim)
        self.apply(self._init_weights)
        self._init_teacher()
        self.collate_fn = collate_fn

        self._flat: Optional[FlatParams] = None
        self._engine: Optional[JepaEngine] = None
        self._anchor: Optional[torch.Tensor] = None
        self._student_bf16_fresh = False
        self._teacher_bf16_fresh = False
        self._grads_ready_hook = None
        self._adam_carry = None
        self._logged: Dict[str, Any] = {}

    # ------------------------------------------------------------------------------------------------ construction
    def _init_weights(self, m: nn.Module) -> None:
        if isinstance(m, nn.Linear):
            trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def _get_pos_embed_params(self, embedding_dim: int) -> nn.Parameter:
        tab = get_1d_sincos_pos_embed_from_grid(embedding_dim, np.arange(self.total_patches, dtype=np.float64))
        return nn.Parameter(torch.from_numpy(tab).float().unsqueeze(0), requires_grad=False)

    def _init_teacher(self) -> None:
        self.teacher_encoder = copy.deepcopy(self.encoder)
        self.teacher_encoder.requires_grad_(False)

    # ------------------------------------------------------------------------------------------------ device / engine
    if not HAS_LIGHTNING:        # Lightning provides `device`, `log_dict`, `global_step` and `trainer` itself
        @property
        def device(self) -> torch.device:
            return self.mask_token.device

        def log_dict(self, data: Dict[str, Any], **kw) -> None:
            self._logged = dict(data)
            log = getattr(self.trainer, "log_dict", None)
            if log is not None:
                log(data, **kw)

    def _max_steps(self) -> int:
        try:
            steps = int(self.trainer.max_steps)
        except (RuntimeError, AttributeError, TypeError):      # Lightning: not attached to a Trainer yet
            steps = -1
        return steps if steps > 0 else _NullTrainer.max_steps

    def _apply(self, fn, *a, **k):
        if getattr(self, "_engine", None) is not None:
            # .cpu() / .to() / .half() read every parameter (and the moments carried below): an update overlapped with the next forward
            # (FusedAdamW.overlap_next_forward) may still be writing them on the engine's side stream
            self._engine.wait_optimizer()
        out = super()._apply(fn, *a, **k)
        if self._flat is not None and self._flat.adam_m is not None:
            self._adam_carry = (self._flat.adam_m, self._flat.adam_v, self._flat.n)   # optimiser moments survive .to() / .cuda()
        self._flat = None          # parameters were re-created: re-flatten lazily
        self._engine = None
        return out

    def state_dict(self, *args, **kw):
        if getattr(self, "_engine", None) is not None:
            self._engine.wait_optimizer()     # an update overlapped with the next forward (FusedAdamW.overlap_next_forward) writes these tensors
        return super().state_dict(*args, **kw)

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        if getattr(self, "_engine", None) is not None:
            self._engine.wait_optimizer()
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        self._student_bf16_fresh = False
        self._teacher_bf16_fresh = False
        return out

    def _ensure_engine(self) -> JepaEngine:
        if self._engine is not None and self._flat is not None and self._flat.owns(self):
            return self._engine
        if self._engine is not None:
            self._engine.wait_optimizer()      # a rebuilt engine must not drop an update still in flight on the old one's side stream
        ops.require_gpu()
        if self.device.type != "cuda":
            raise RuntimeError("wavjepa_amd.JEPA computes only on an MI355X: move the module with .cuda() first "
                               "(there is no CPU fallback on the product path)")
        ext = self.extract_audio
        spec = ext.conv_layers_spec
        if hasattr(ext, "cnns"):             # ConvChannelFeatureExtractor: every audio channel through a mono conv stack
            streams, conv_in = ext.in_channels, 1
            prefixes = ["extract_audio.cnns.0."] if ext.weight_sharing else [f"extract_audio.cnns.{c}." for c in range(streams)]
        else:
            streams, conv_in, prefixes = 1, ext.in_channels, ["extract_audio.cnn."]
        self._audio_channels = ext.in_channels
        self._flat = FlatParams(self, self.device)
        carry = getattr(self, "_adam_carry", None)
        if carry is not None:
            if carry[2] != self._flat.n:
                raise RuntimeError("optimiser state exists for a different parameter layout; rebuild the optimiser")
            self._flat.adam_m, self._flat.adam_v = carry[0].to(self.device), carry[1].to(self.device)
            self._adam_carry = None
        cfg = EngineConfig(conv_spec=spec, in_channels=conv_in, streams=streams, conv_prefixes=prefixes, n_samples=self.target_length,
                           d_enc=self.encoder_embedding_dim, h_enc=self.n_encoder_heads, l_enc=self.encoder.num_layers,
                           d_dec=self.decoder_embedding_dim, h_dec=self.n_decoder_heads, l_dec=self.decoder.num_layers,
                           top_k=int(self.hparams.average_top_k_layers), ln_eps=self.encoder.layer_norm_eps)
        self._engine = JepaEngine(cfg, self._flat, self.pos_encoding_encoder.data, self.pos_encoding_decoder.data)
        self._anchor = torch.zeros(1, device=self.device, requires_grad=True)
        self._student_bf16_fresh = False
        self._teacher_bf16_fresh = False
        return self._engine

    def _prepare_weights(self) -> None:
        eng = self._ensure_engine()
        fresh = self._student_bf16_fresh and self._teacher_bf16_fresh
        self._flat.bf16_fresh = fresh
        eng.prepare_weights()
        # only the fused optimiser / EMA kernels keep the shadows in sync; anything else (a stock torch optimiser,
        # manual edits) changes fp32 masters behind our back, so stay conservative unless they ran.
        self._student_bf16_fresh = False
        self._teacher_bf16_fresh = True

    # ------------------------------------------------------------------------------------------------ schedule / EMA
    def _get_ema_decay(self) -> float:
        if self.global_step >= self.ema_end_step:
            return self.hparams.ema_end_decay
        r = self.hparams.ema_end_decay - self.hparams.ema_decay
        return self.hparams.ema_end_decay - r * (1 - self.global_step / self.ema_end_step)

    @torch.no_grad()
    def _step_teacher(self) -> None:
        self._ensure_engine().ema_step(float(self._get_ema_decay()))
        self._teacher_bf16_fresh = True

    def configure_optimizers(self):
        optimizer = FusedAdamW(self, lr=self.hparams.lr, betas=self.hparams.adam_betas, eps=self.hparams.adam_eps,
                               weight_decay=self.hparams.adam_weight_decay)
        sched = cosine_schedule_with_warmup(optimizer, num_warmup_steps=int(self.hparams.warmup_steps),
                                            num_training_steps=self._max_steps())
        return {"optimizer": optimizer, "lr_scheduler": {"scheduler": sched, "interval": "step"}}

    # ------------------------------------------------------------------------------------------------ batch preparation
    def on_after_batch_transfer(self, batch, dataloader_idx: int = 0):
        """Random 2.01 s crops, per-crop normalisation, bf16 cast, flatten, shuffle of the audio rows
        (reference jepa.py:275-316).  RNG calls mirror the reference's (torch.randint on the device, torch.randperm on CPU)."""
        audio_batch, ctx_masks, target_indices, ctx_and_target_masks = batch
        if audio_batch.ndim != 3:
            audio_batch = audio_batch.unsqueeze(1)
        self._ensure_engine()
        audio_batch = audio_batch.to(self.device, dtype=torch.float32).contiguous()
        B, C, L_full = audio_batch.shape
        S = self.nr_samples_per_audio
        starts = torch.randint(0, L_full - self.target_length + 1, (B, S), device=self.device)
        idx = torch.randperm(B * S)
        perm_inv = torch.empty_like(idx)
        perm_inv[idx] = torch.arange(B * S)
        out = torch.empty(B * S, C, self.target_length, dtype=torch.bfloat16, device=self.device)
        # (through the page-locked staging ring: a copy from pageable memory would hold the host until the previous step's optimiser
        # kernels have run)
        perm_dev = pack_upload([perm_inv.to(torch.int32).numpy()], self.device)[0]
        ops.crop_normalize_bf16(audio_batch, starts.to(torch.int32), out, B=B, S=S, C=C, L_full=L_full, length=self.target_length,
                                perm_inv=perm_dev)
        return out, self.collate_fn(ctx_masks), self.collate_fn(target_indices), self.collate_fn(ctx_and_target_masks)

    # ------------------------------------------------------------------------------------------------ step
    def training_step(self, batch, batch_idx: int) -> ForwardReturn:
        audio_input, ctx_masks, target_indices, ctx_and_target_masks = batch
        out = self(audio_input, ctx_masks, target_indices, ctx_and_target_masks)
        self.log_dict({"train/loss": out["loss"], "ema": self._get_ema_decay()}, prog_bar=True, sync_dist=True)
        self._step_teacher()           # EMA with the pre-update student, before backward (reference jepa.py:330-331)
        return out

    def forward(self, audio: torch.Tensor, ctx_masks, target_indices, ctx_and_target_masks) -> ForwardReturn:
        eng = self._ensure_engine()
        if audio.ndim != 3:
            raise ValueError("audio must be [batch, channels, samples]")
        audio = audio.to(self.device, dtype=torch.bfloat16).contiguous()
        if audio.shape[-1] != self.target_length:
            raise ValueError(f"expected {self.target_length} samples per clip, got {audio.shape[-1]}")
        if audio.shape[1] != self.extract_audio.in_channels:
            raise ValueError(f"expected {self.extract_audio.in_channels} audio channel(s), got {audio.shape[1]}")
        plan = ctx_masks if isinstance(ctx_masks, MaskPlan) else make_mask_plan(ctx_masks, target_indices, ctx_and_target_masks, self.device)
        self._prepare_weights()
        eng.forward(audio, plan)
        N, T = audio.shape[0], eng.T
        if torch.is_grad_enabled():
            loss = _EngineLoss.apply(self._anchor, self)
        else:
            loss = eng.loss[0].clone()
        return _StepOutputs(eng, local_features=eng.lf.view(N, T, -1), contextual_features=eng.cf[:plan.n_ctx], loss=loss,
                            targets=eng.targets.view(N, T, -1))

    @torch.no_grad()
    def get_audio_representation(self, audio: torch.Tensor, padding_mask: Optional[torch.Tensor]) -> torch.Tensor:
        """Student-only inference (reference jepa.py:456-467): [B, C, L] -> [B, T, d_enc] fp32."""
        self.eval()
        eng = self._ensure_engine()
        audio = audio.to(self.device, dtype=torch.bfloat16).contiguous()
        self._prepare_weights()
        mask = None if padding_mask is None else padding_mask.to(self.device).to(torch.uint8).contiguous()
        return eng.infer(audio, mask).clone()
