"""Denoiser stage for MI355X: the reference's `wavjepa/denoiser.py` surface over the HIP engine.

`Denoiser` keeps the reference's constructor, hooks, returned keys and state_dict names (extract_audio.*, feature_norms.*,
post_extraction_mapper.*, encoder.*, pos_encoding_encoder): a student encoder (conv stack + post-norm ViT, no predictor) is trained so
that its contextual features of BOTH the clean clip and the generated scene (RIR + noise) match the features a frozen, pre-trained
JEPA produces for the clean clip (denoiser.py:312-357):

    loss = alpha * mse(encoder(clean), teacher(clean)) + (1 - alpha) * mse(encoder(generated), teacher(clean))

Every tensor op runs in `libwavjepa_hip.so`: the same front-end / transformer kernels as the JEPA step (dense, unmasked), the scene
augmentation and resampling kernels in the batch hook, `wj_mse_groups` for the loss.  No PyTorch-op fallback.
"""
from __future__ import annotations

from typing import Any, Dict, Optional

import numpy as np
import torch
from torch import nn

from . import ops, scene
from .engine import EngineConfig, JepaEngine, _layer_ptrs
from .extractors.audio_extractor import Extractor
from .jepa import HAS_LIGHTNING, FusedAdamW, TransformerStack, _AttrDict, _ModuleBase, _NullTrainer, collate_fn, cosine_schedule_with_warmup
from .params import FlatParams
from .pos_embed import get_1d_sincos_pos_embed_from_grid
from .resample import resample
from .types import ForwardReturn, TransformerEncoderCFG, TransformerLayerCFG

ORIGINAL_SR = 32000


class DenoiserEngine(JepaEngine):
    """Dense student encoder over 2N clips (N clean, then N generated) against given targets; reuses the JEPA engine's front-end,
    layer forward / backward and front-end backward, without masks, predictor or EMA teacher."""

    def _bind_params(self) -> None:
        f, c = self.flat, self.cfg
        self.enc_layers = [_layer_ptrs(f, f"encoder.layers.{i}.", False) for i in range(c.l_enc)]
        self.dec_layers, self.tea_layers = [], []

    def forward_pair(self, audio: torch.Tensor, targets: torch.Tensor, alpha: float) -> None:
        """audio bf16 [2N, C, L] (clean clips first), targets fp32 [N, T, d_enc] -> self.dn_loss = (loss, loss_clean, loss_generated)."""
        c, f = self.cfg, self.flat
        N2 = audio.shape[0]
        if N2 % 2 or targets.numel() != (N2 // 2) * self.T * c.d_enc:
            raise ValueError("denoiser step: 2N clips (N clean + N generated) and N x T x d_enc targets expected")
        self.alloc(N2, train=True, G=1)
        if not hasattr(self, "dn_loss") or self.dn_w.device != self.dev:
            self.dn_loss = torch.zeros(3, dtype=torch.float32, device=self.dev)
            self.dn_w = torch.zeros(2, dtype=torch.float32, device=self.dev)
            self.dn_ws = torch.empty(ops.workspace_bytes("wj_mse_groups", G=2, n=1) // 4, dtype=torch.float32, device=self.dev)
        self.dn_w.copy_(torch.tensor([alpha, 1.0 - alpha], dtype=torch.float32))
        self.audio, self.plan, self.ragged_step = audio, None, False
        self.set_wt_need(self.M, 0)
        self.dn_targets = targets.reshape(-1, c.d_enc).float().contiguous()
        self._frontend(audio)
        x, xb = self.lf, self.lf_b
        for w, a in zip(self.enc_layers, self.enc_acts):
            self._layer_fwd(w, a, x, xb, self.M, c.d_enc, c.h_enc, N2, None)
            x, xb = a.x2, a.x2b
        ops.layernorm_fwd(x, f.ptr32("encoder.norm.weight"), f.ptr32("encoder.norm.bias"), M=self.M, D=c.d_enc, eps=c.norm_eps,
                          y_f32=self.enc_out, mean=self.enc_fm, rstd=self.enc_fr)
        self.dn_n = (N2 // 2) * self.T * c.d_enc
        ops.mse_groups(self.enc_out, self.dn_targets, self.dn_w, self.dn_loss, self.dn_ws, n=self.dn_n, G=2)

    def backward_pair(self, gscale_ptr: int = 0) -> None:
        c, f = self.cfg, self.flat
        De, M, N2 = c.d_enc, self.M, self.N
        f.g32.zero_()
        self.refresh_wt()
        bw = self.bw["enc"]
        ops.mse_groups(self.enc_out, self.dn_targets, self.dn_w, self.dn_loss, self.dn_ws, n=self.dn_n, G=2, dpreds=bw["dx1"],
                       gscale=gscale_ptr if gscale_ptr else None)
        last = self.enc_acts[-1]
        ops.layernorm_bwd(bw["dx1"], last.x2, f.ptr32("encoder.norm.weight"), self.enc_fm, self.enc_fr, M=M, D=De, ds_f32=bw["dy"],
                          dgamma=f.gptr("encoder.norm.weight"), dbeta=f.gptr("encoder.norm.bias"), workspace=self.red_ws)
        dy, dyb = bw["dy"], None
        for i in range(c.l_enc - 1, -1, -1):
            x_in, xb_in = (self.lf, self.lf_b) if i == 0 else (self.enc_acts[i - 1].x2, self.enc_acts[i - 1].x2b)
            dy, dyb, _ = self._layer_bwd(self.enc_layers[i], self.enc_acts[i], x_in, xb_in, dy, dyb, bw["dy"], M, De, c.h_enc, N2, None, bw,
                                         i % bw["nbuf"], None, flush=(c.l_enc - 1 - i) % bw["group"] == bw["group"] - 1 or i == 0,
                                         bottom=i == 0)
        self._frontend_bwd(dy, False, None)
        self._flush_folds()
        self._join_side()
        bw["used"] = [False] * bw["nbuf"]


class _DenoiserLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor: torch.Tensor, module: "Denoiser") -> torch.Tensor:
        ctx.module = module
        return module._engine.dn_loss[0].clone()

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor):
        m = ctx.module
        g = grad_out.contiguous().float()
        m._engine.backward_pair(gscale_ptr=g.data_ptr())
        m._flat.attach_grads()
        return torch.zeros_like(m._anchor), None


class Denoiser(_ModuleBase):
    TARGET_SECONDS: int = 10

    def __init__(self, feature_extractor: Extractor, transformer_encoder_layers_cfg: TransformerLayerCFG,
                 transformer_encoder_cfg: TransformerEncoderCFG, lr: float = 0.0001, adam_betas=(0.9, 0.98), adam_eps: float = 1e-06,
                 adam_weight_decay: float = 0.0, resample_sr: int = 16000, process_audio_seconds: float = 2.01, nr_samples_per_audio=16,
                 size: str = "base", alpha: float = 0.0, **kwargs: Any):
        super().__init__()
        self.alpha = alpha
        self.sr = resample_sr
        self.target_audio_length = self.TARGET_SECONDS * self.sr
        self.process_audio_seconds = process_audio_seconds
        self.nr_samples_per_audio = nr_samples_per_audio
        self.target_length = int(resample_sr * process_audio_seconds)
        self.total_patches = feature_extractor.total_patches(self.target_length)
        if HAS_LIGHTNING:
            self.save_hyperparameters(ignore=["feature_encoder", "feature_extractor", "loss_fn"])
        else:
            self.hparams = _AttrDict(lr=lr, adam_betas=tuple(adam_betas), adam_eps=adam_eps, adam_weight_decay=adam_weight_decay,
                                     resample_sr=resample_sr, process_audio_seconds=process_audio_seconds,
                                     nr_samples_per_audio=nr_samples_per_audio, size=size, alpha=alpha)
            self.global_step = 0
            self.trainer = _NullTrainer()
        self.extract_audio = feature_extractor
        self.feature_norms = nn.LayerNorm(self.extract_audio.embedding_dim)
        enc_l, enc_c = dict(transformer_encoder_layers_cfg), dict(transformer_encoder_cfg)
        if size == "large":                      # reference denoiser.py:127-131
            enc_l.update(nhead=16, d_model=1024, dim_feedforward=4096)
            enc_c.update(num_layers=24)
        self.n_encoder_heads = enc_l["nhead"]
        self.encoder_embedding_dim = enc_l["d_model"]
        self.encoder = TransformerStack(enc_l, enc_c["num_layers"])
        c_feat = feature_extractor.embedding_dim
        self.post_extraction_mapper = nn.Linear(c_feat, self.encoder_embedding_dim) if c_feat != self.encoder_embedding_dim else None
        tab = get_1d_sincos_pos_embed_from_grid(self.encoder_embedding_dim, np.arange(self.total_patches, dtype=np.float64))
        self.pos_encoding_encoder = nn.Parameter(torch.from_numpy(tab).float().unsqueeze(0), requires_grad=False)
        self.collate_fn = collate_fn
        self.teacher = None
        self._flat: Optional[FlatParams] = None
        self._engine: Optional[DenoiserEngine] = None
        self._anchor: Optional[torch.Tensor] = None
        self._student_bf16_fresh = False
        self._logged: Dict[str, Any] = {}

    # ------------------------------------------------------------------------------------------------ teacher
    def _set_teacher(self, weights_ckpt) -> None:
        """reference denoiser.py:146-181: a frozen WavJEPA-base from a Lightning checkpoint (`state_dict` key, `._orig_mod`
        infixes stripped).  Also accepts an already built `wavjepa_amd.jepa.JEPA`."""
        from .extractors import ConvFeatureExtractor
        from .jepa import JEPA
        if isinstance(weights_ckpt, nn.Module):
            model = weights_ckpt
        else:
            weights = torch.load(weights_ckpt, weights_only=False)
            sd = {k.replace("._orig_mod", ""): v for k, v in weights["state_dict"].items()}
            extractor = ConvFeatureExtractor(conv_layers_spec=[(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512, 2, 2)], in_channels=1)
            model = JEPA(feature_extractor=extractor, transformer_encoder_cfg=TransformerEncoderCFG.create(),
                         transformer_encoder_layers_cfg=TransformerLayerCFG.create(), transformer_decoder_cfg=TransformerEncoderCFG.create(),
                         transformer_decoder_layers_cfg=TransformerLayerCFG.create(d_model=384), resample_sr=self.sr, size="base",
                         process_audio_seconds=self.process_audio_seconds)
            model.load_state_dict(sd, strict=False)
        # Frozen by construction: the teacher is NOT a sub-module (its weights are in neither parameters() nor the optimiser; its
        # state_dict entries are added by hand under `teacher.*`, as the reference's checkpoints have them) and only its inference entry is ever called.  (The reference also flips requires_grad off, denoiser.py:177-178;
        # here that flag selects what lives in the flat parameter buffer the kernels read, so it stays as it is.)
        model.eval()
        object.__setattr__(self, "teacher", model)

    # ------------------------------------------------------------------------------------------------ device / engine
    if not HAS_LIGHTNING:
        @property
        def device(self) -> torch.device:
            return self.feature_norms.weight.device

        def log_dict(self, data: Dict[str, Any], **kw) -> None:
            self._logged = dict(data)

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._flat = None
        self._engine = None
        if self.teacher is not None:
            self.teacher._apply(fn, *a, **k)
        return out

    def state_dict(self, *args, destination=None, prefix: str = "", keep_vars: bool = False):
        """The reference's Denoiser holds its frozen teacher as a sub-module (denoiser.py:146-181), so its Lightning checkpoints carry
        `teacher.*` entries.  Here the teacher is deliberately NOT a sub-module (it must stay out of parameters(), the optimiser and
        the flat buffer), so its entries are added under the same names: a checkpoint written here strict-loads in the reference."""
        out = super().state_dict(*args, destination=destination, prefix=prefix, keep_vars=keep_vars)
        if self.teacher is not None:
            self.teacher.state_dict(destination=out, prefix=prefix + "teacher.", keep_vars=keep_vars)
        return out

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        """`teacher.*` entries (a reference Denoiser checkpoint, or one written here) are routed to the frozen teacher when there is
        one and dropped otherwise (`_set_teacher` builds it from its own checkpoint); everything else loads as usual."""
        own = {k: v for k, v in state_dict.items() if not k.startswith("teacher.")}
        tea = {k[len("teacher."):].replace("._orig_mod", ""): v for k, v in state_dict.items() if k.startswith("teacher.")}
        out = super().load_state_dict(own, strict=strict, **kw)
        if tea and self.teacher is not None:
            self.teacher.load_state_dict(tea, strict=strict)
        self._student_bf16_fresh = False
        return out

    def _ensure_engine(self) -> DenoiserEngine:
        if self._engine is not None and self._flat is not None and self._flat.owns(self):
            return self._engine
        ops.require_gpu()
        if self.device.type != "cuda":
            raise RuntimeError("wavjepa_amd.Denoiser computes only on an MI355X: move the module with .cuda() first "
                               "(there is no CPU fallback on the product path)")
        ext = self.extract_audio
        if hasattr(ext, "cnns"):
            raise NotImplementedError("the denoiser stage is mono (reference denoiser.py:268)")
        self._flat = FlatParams(self, self.device)
        cfg = EngineConfig(conv_spec=ext.conv_layers_spec, in_channels=ext.in_channels, streams=1, conv_prefixes=["extract_audio.cnn."],
                           n_samples=self.target_length, d_enc=self.encoder_embedding_dim, h_enc=self.n_encoder_heads,
                           l_enc=self.encoder.num_layers, d_dec=64, h_dec=1, l_dec=0, top_k=1, ln_eps=self.encoder.layer_norm_eps)
        self._engine = DenoiserEngine(cfg, self._flat, self.pos_encoding_encoder.data,
                                      torch.zeros(self.total_patches, 64, device=self.device))
        self._anchor = torch.zeros(1, device=self.device, requires_grad=True)
        self._student_bf16_fresh = False
        return self._engine

    def _prepare_weights(self) -> None:
        eng = self._ensure_engine()
        self._flat.bf16_fresh = self._student_bf16_fresh
        eng.prepare_weights()
        self._student_bf16_fresh = False

    def configure_optimizers(self):
        """reference denoiser.py:200-213: AdamW + cosine schedule with 5000 warm-up steps."""
        optimizer = FusedAdamW(self, lr=self.hparams.lr, betas=self.hparams.adam_betas, eps=self.hparams.adam_eps,
                               weight_decay=self.hparams.adam_weight_decay)
        try:
            steps = int(self.trainer.max_steps)
        except (RuntimeError, AttributeError, TypeError):
            steps = -1
        sched = cosine_schedule_with_warmup(optimizer, num_warmup_steps=5000, num_training_steps=steps if steps > 0 else _NullTrainer.max_steps)
        return {"optimizer": optimizer, "lr_scheduler": {"scheduler": sched, "interval": "step"}}

    # ------------------------------------------------------------------------------------------------ batch preparation
    def on_after_batch_transfer(self, batch, dataloader_idx: int = 0):
        """reference denoiser.py:217-309: scene generation (source RIR + noise RIRs + SNR mix, receiver channel 0), 32 kHz -> `sr`
        kaiser-sinc resampling of the scene and of the clean source, the SAME random crops of both, per-crop normalisation, bf16,
        flatten, one shared shuffle.  Returns (generated, clean), each bf16 [B * S, 1, target_length]."""
        self._ensure_engine()
        dev = self.device
        # (Lightning calls this hook with the batch already on the device; the plain trainer hands over the loader's CPU batch)
        audio, source_rir, noise, noise_length, noise_start_idx, noise_rirs, snr = (
            t.to(dev, non_blocking=True) if isinstance(t, torch.Tensor) else t for t in batch)
        audio = audio.to(dev, dtype=torch.float32)
        generated = scene.generate_scene(source_rir=source_rir, source=audio, noise=noise, real_noise_length=noise_length,
                                         noise_start_idx=noise_start_idx, noise_rirs=noise_rirs, snr=snr)
        if audio.ndim != 3:
            audio = audio.unsqueeze(1)
        if generated.ndim != 3:
            generated = generated.unsqueeze(1)
        assert generated.ndim == audio.ndim
        clean = audio
        if self.sr != ORIGINAL_SR:
            generated = resample(generated, resample_sr=self.sr, original_sr=ORIGINAL_SR)
            clean = resample(clean, resample_sr=self.sr, original_sr=ORIGINAL_SR)
        assert generated.shape[1] == 1, f"Generated scene has more channels than in channels, {generated.shape}, 1"
        B, C, L_full = generated.shape
        S = self.nr_samples_per_audio
        starts = torch.randint(0, L_full - self.target_length + 1, (B, S), device=dev)
        idx = torch.randperm(B * S)
        perm_inv = torch.empty_like(idx)
        perm_inv[idx] = torch.arange(B * S)
        perm_dev = perm_inv.to(torch.int32).to(dev, non_blocking=True)
        starts32 = starts.to(torch.int32)
        outs = []
        for src in (generated, clean):
            out = torch.empty(B * S, C, self.target_length, dtype=torch.bfloat16, device=dev)
            ops.crop_normalize_bf16(src.contiguous(), starts32, out, B=B, S=S, C=C, L_full=L_full, length=self.target_length, perm_inv=perm_dev)
            outs.append(out)
        return outs[0], outs[1]

    # ------------------------------------------------------------------------------------------------ step
    def training_step(self, batch, batch_idx: int) -> ForwardReturn:
        generated_scene, clean_scene = batch
        out = self(generated_scene, clean_scene)
        self.log_dict({"train/loss": out["loss"], "loss_clean": out["loss_clean"], "loss_denoise_dereverb": out["loss_denoise_dereverb"]},
                      prog_bar=True, sync_dist=True)
        return out

    def forward(self, generated_scene: torch.Tensor, clean_scene: torch.Tensor) -> ForwardReturn:
        if self.teacher is None:
            raise RuntimeError("Denoiser needs a teacher: call _set_teacher(checkpoint or JEPA) first (reference denoiser.py:146)")
        eng = self._ensure_engine()
        for t in (generated_scene, clean_scene):
            if t.ndim != 3 or t.shape[-1] != self.target_length:
                raise ValueError(f"expected [batch, 1, {self.target_length}] clips, got {tuple(t.shape)}")
        clean = clean_scene.to(self.device, dtype=torch.bfloat16).contiguous()
        gen = generated_scene.to(self.device, dtype=torch.bfloat16).contiguous()
        with torch.no_grad():
            targets = self.teacher.get_audio_representation(clean, padding_mask=None)
        self._prepare_weights()
        eng.forward_pair(torch.cat([clean, gen], dim=0), targets, float(self.alpha))
        if torch.is_grad_enabled():
            loss = _DenoiserLoss.apply(self._anchor, self)
        else:
            loss = eng.dn_loss[0].clone()
        return ForwardReturn(loss=loss, loss_clean=eng.dn_loss[1].clone(), loss_denoise_dereverb=eng.dn_loss[2].clone())

    @torch.no_grad()
    def encoder_forward(self, audio: torch.Tensor) -> torch.Tensor:
        """Contextual features of the student for [B, 1, L] clips (inference; fp32 [B, T, d_enc])."""
        eng = self._ensure_engine()
        self._prepare_weights()
        return eng.infer(audio.to(self.device, dtype=torch.bfloat16).contiguous(), None).clone()
