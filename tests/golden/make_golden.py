#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by running the REFERENCE (labhamlet/wavjepa at
/root/reference, imported with the stub recipe of `_ref_import.py`).

Runs only in the build container (the reference does not exist on the GPU box).  Re-run with
    python tests/golden/make_golden.py
The fixtures are data only (inputs + the reference's outputs); no reference source is stored.

Fixtures written:
  tiny_model.npz      tiny JEPA (conv 32ch, d=64/2 layers, predictor d=32/2 layers): the reference's own
                      random-init state_dict, masks, audio, fp32 forward outputs, all parameter gradients'
                      norms + two full gradients, bf16-autocast(CPU) loss.
  tiny_traj.npz       12 optimisation steps of the tiny model in the reference's order
                      (training_step incl. EMA -> backward -> clip 5 -> AdamW -> scheduler) with a 3-step
                      warm-up, and 4 steps with the stock 100 000-step warm-up: losses, grad norms, final
                      parameter checksums.
  masks.npz           maskers under a pinned numpy Generator sequence (AudioSet + LibriSpeech settings).
  crops.npz           on_after_batch_transfer inputs/outputs for a pinned torch seed.
  base_forward.npz    base model (d=768/12 layers, predictor 384/12, conv 512ch) with hash-synthesised
                      weights (tests/golden/synth.py), N=2: loss, per-group grad norms, tensor slices.
  misc.npz            sin-cos tables slices/checksums, EMA decay schedule, LR schedule samples.
  channel_frontend.npz  ConvChannelFeatureExtractor (own / shared stacks) weights + input + output, get_binaural_pos_embed
                      slices, channel-based masks under a pinned numpy Generator sequence.
  large_forward.npz   (python tests/golden/make_golden.py large) the reference's size="large" model (student d=1024 x 24, 16 heads), synthesised
                      weights, N=2: state_dict layout, fp32 loss, slices, per-group gradient norms.
  base_traj.npz       (python tests/golden/make_golden.py base_traj) 100 optimisation steps of the BASE model, N=4, through
                      the reference's training_step / EMA / clip / AdamW / schedule, fp32 and bf16-autocast: per-step loss,
                      grad norm, lr, EMA decay, final parameter checksums and slices (the north-star trajectory).
"""
import os
import inspect
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from _ref_import import import_reference  # noqa: E402
import _ref_import as RI  # noqa: E402
import synth  # noqa: E402

R = import_reference()
import transformers  # noqa: E402

TINY_SPEC = [(32, 10, 5)] + [(32, 3, 2)] * 4 + [(32, 2, 2)]
BASE_SPEC = [(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512, 2, 2)]


def build_ref(spec, d_enc, nh_enc, L_enc, d_dec, nh_dec, L_dec, top_k, seconds=2.01, samples=2, size="base"):
    ext = R.ConvFeatureExtractor(conv_layers_spec=list(spec), in_channels=1)
    return R.jepa.JEPA(
        feature_extractor=ext,
        transformer_encoder_cfg=R.TransformerEncoderCFG.create(num_layers=L_enc),
        transformer_encoder_layers_cfg=R.TransformerLayerCFG.create(d_model=d_enc, nhead=nh_enc),
        transformer_decoder_cfg=R.TransformerEncoderCFG.create(num_layers=L_dec),
        transformer_decoder_layers_cfg=R.TransformerLayerCFG.create(d_model=d_dec, nhead=nh_dec),
        lr=4e-4, adam_betas=(0.9, 0.98), adam_weight_decay=0.04,
        average_top_k_layers=top_k, process_audio_seconds=seconds, nr_samples_per_audio=samples, size=size)


class PinnedRng:
    """Replaces np.random.default_rng so that the k-th call returns default_rng(base + k)."""

    def __init__(self, base):
        self.base, self.k, self._orig = base, 0, np.random.default_rng

    def __call__(self, seed=None):
        g = self._orig(self.base + self.k)
        self.k += 1
        return g

    def __enter__(self):
        np.random.default_rng = self
        return self

    def __exit__(self, *a):
        np.random.default_rng = self._orig


def sd_numpy(model):
    return {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}


def gen_masks():
    out = {}
    with PinnedRng(1000):
        mk = R.masking.TimeInverseBlockMasker(target_masks_per_context=4, context_mask_prob=0.65, context_mask_length=10,
                                              target_prob=0.25, target_length=10, ratio_cutoff=0.1)
        c, t, v = mk(batch_size=8, n_times=200, in_channels=1)
    out.update(as_ctx=c.numpy(), as_tgt=t.numpy(), as_vis=v.numpy(), as_base=1000)
    with PinnedRng(5000):
        mk = R.masking.SpeechMasker(target_masks_per_context=4, target_prob=0.1, target_length=10, ratio_cutoff=0.5,
                                    min_context_len=5)
        c, t, v = mk(batch_size=8, n_times=200, in_channels=1)
    out.update(ls_ctx=c.numpy(), ls_tgt=t.numpy(), ls_vis=v.numpy(), ls_base=5000)
    with PinnedRng(9000):
        mk = R.masking.TimeInverseBlockMasker(target_masks_per_context=4, context_mask_prob=0.65, context_mask_length=10,
                                              target_prob=0.25, target_length=10, ratio_cutoff=0.1)
        c, t, v = mk(batch_size=4, n_times=400, in_channels=1)
    out.update(as400_ctx=c.numpy(), as400_tgt=t.numpy(), as400_vis=v.numpy(), as400_base=9000)
    np.savez_compressed(os.path.join(HERE, "masks.npz"), **out)
    return out


def gen_tiny(masks):
    torch.manual_seed(1234)
    m = build_ref(TINY_SPEC, 64, 4, 2, 32, 4, 2, top_k=2)
    # make biases / norm params non-trivial so that every term is exercised
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith("bias"):
                p.add_(0.02 * torch.randn_like(p))
            elif "norm" in n and n.endswith("weight") or n.endswith("cnn.0.2.weight"):
                p.add_(0.1 * torch.randn_like(p))
        for s, t in zip(m.encoder.parameters(), m.teacher_encoder.parameters()):
            t.copy_(s)
        # perturb the teacher so that student != teacher in the fixture
        for t in m.teacher_encoder.parameters():
            t.add_(0.01 * torch.randn_like(t))
    sd = sd_numpy(m)
    N = 2
    audio = torch.randn(N, 1, 32159)
    ctx = torch.from_numpy(masks["as_ctx"][:N])
    tgt = torch.from_numpy(masks["as_tgt"][:N])
    vis = torch.from_numpy(masks["as_vis"][:N])
    m.train()
    out = m(audio, ctx, tgt, vis)
    out["loss"].backward()
    fx = {f"sd::{k}": v for k, v in sd.items()}
    fx.update(audio=audio.numpy(), ctx=ctx.numpy(), tgt=tgt.numpy(), vis=vis.numpy())
    for k in ("local_features", "contextual_features", "preds", "targets", "loss"):
        fx[f"out::{k}"] = out[k].detach().float().numpy()
    gn = {}
    for n, p in m.named_parameters():
        if p.grad is not None:
            gn[n] = float(p.grad.norm())
    fx["grad_names"] = np.array(list(gn.keys()))
    fx["grad_norms"] = np.array(list(gn.values()), dtype=np.float64)
    fx["grad::extract_audio.cnn.0.0.weight"] = m.extract_audio.cnn[0][0].weight.grad.numpy()
    fx["grad::encoder.layers.0.self_attn.in_proj_weight"] = m.encoder.layers[0].self_attn.in_proj_weight.grad.numpy()
    fx["grad::mask_token"] = m.mask_token.grad.numpy()
    # the stub LightningModule is a plain nn.Module: conv0 input grad check is not needed.
    with torch.autocast("cpu", dtype=torch.bfloat16):
        ob = m(audio.to(torch.bfloat16), ctx, tgt, vis)
    fx["out_bf16cpu::loss"] = ob["loss"].detach().float().numpy()
    # inference path
    pad = torch.zeros(N, 200, dtype=torch.bool)
    pad[:, 150:] = True
    rep = m.get_audio_representation(audio, pad)
    m.train()
    fx["out::audio_representation"] = rep.detach().numpy()
    fx["pad_mask"] = pad.numpy()
    np.savez_compressed(os.path.join(HERE, "tiny_model.npz"), **fx)
    return m, sd, (audio, ctx, tgt, vis)


def run_traj(m, batches, steps, warmup, total, autocast_bf16=False):
    trainables = [p for p in m.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(trainables, lr=m.hparams.lr, betas=m.hparams.adam_betas, eps=m.hparams.adam_eps,
                            weight_decay=m.hparams.adam_weight_decay)
    sch = transformers.get_cosine_schedule_with_warmup(opt, num_warmup_steps=warmup, num_training_steps=total)
    losses, gnorms, emas, lrs = [], [], [], []
    m.global_step = 0
    m.train()
    for i in range(steps):
        b = batches[i % len(batches)]
        opt.zero_grad()
        emas.append(m._get_ema_decay())
        lrs.append(opt.param_groups[0]["lr"])
        if autocast_bf16:                    # Lightning's precision="bf16-mixed": autocast around the step's forward only
            with torch.autocast("cpu", dtype=torch.bfloat16):   # (one context per step: its weight-cast cache must not outlive
                out = m.training_step(b, i)                      # the optimiser update)
        else:
            out = m.training_step(b, i)      # forward + EMA (reference jepa.py:318-333)
        out["loss"].backward()
        gn = torch.nn.utils.clip_grad_norm_(m.parameters(), 5.0)   # Lightning gradient_clip_val=5, "norm"
        opt.step()
        sch.step()
        m.global_step += 1
        losses.append(float(out["loss"]))
        gnorms.append(float(gn))
    return np.array(losses), np.array(gnorms), np.array(emas), np.array(lrs)


def checksums(m):
    names, sums, abss = [], [], []
    for k, v in m.state_dict().items():
        names.append(k)
        sums.append(float(v.double().sum()))
        abss.append(float(v.double().abs().sum()))
    return np.array(names), np.array(sums), np.array(abss)


def gen_traj(masks):
    fx = {}
    torch.manual_seed(77)
    batches = []
    for j in range(3):
        a = torch.randn(2, 1, 32159)
        sl = slice(2 * j, 2 * j + 2)
        batches.append((a, torch.from_numpy(masks["as_ctx"][sl]), torch.from_numpy(masks["as_tgt"][sl]),
                        torch.from_numpy(masks["as_vis"][sl])))
    fx["audio"] = np.stack([b[0].numpy() for b in batches])
    for tag, (warm, total, steps) in dict(short=(3, 20, 12), stock=(100000, 375000, 4)).items():
        torch.manual_seed(1234)
        m = build_ref(TINY_SPEC, 64, 4, 2, 32, 4, 2, top_k=2)
        m.ema_end_step = 100000
        if tag == "short":
            fx.update({f"sd0::{k}": v for k, v in sd_numpy(m).items()})
            # faster EMA so that the teacher visibly moves in 12 steps
            m.hparams["ema_decay"] = 0.9
            m.hparams["ema_end_decay"] = 0.99
            m.ema_end_step = 10
        l, g, e, lr = run_traj(m, batches, steps, warm, total)
        n, s, a = checksums(m)
        fx.update({f"{tag}::loss": l, f"{tag}::gnorm": g, f"{tag}::ema": e, f"{tag}::lr": lr,
                   f"{tag}::names": n, f"{tag}::sum": s, f"{tag}::abs": a})
        if tag == "short":
            fx["short::final::encoder.layers.1.linear1.weight"] = m.encoder.layers[1].linear1.weight.detach().numpy().copy()
            fx["short::final::teacher_encoder.layers.1.linear1.weight"] = m.teacher_encoder.layers[1].linear1.weight.detach().numpy().copy()
    np.savez_compressed(os.path.join(HERE, "tiny_traj.npz"), **fx)


def gen_crops():
    torch.manual_seed(4321)
    m = build_ref(TINY_SPEC, 64, 4, 2, 32, 4, 2, top_k=2, seconds=0.25, samples=3)
    T = m.total_patches
    B = 2
    src = torch.randn(B, 1, 9000) * 0.3 + 0.05
    ctx = torch.zeros(B, 3, T, dtype=torch.bool)
    tg = torch.zeros(B, 3, 4, T, dtype=torch.bool)
    torch.manual_seed(99)
    a, c, t, v = m.on_after_batch_transfer((src, ctx, tg, tg.clone()), 0)
    # recover the internal random draws by replaying the same generator sequence
    torch.manual_seed(99)
    starts = torch.randint(0, 9000 - m.target_length + 1, (B, 3))
    perm = torch.randperm(B * 3)
    np.savez_compressed(os.path.join(HERE, "crops.npz"), src=src.numpy(), starts=starts.numpy(), perm=perm.numpy(),
                        target_length=m.target_length, total_patches=T,
                        out_bits=a.view(torch.int16).numpy(), out_shape=np.array(a.shape),
                        ctx_shape=np.array(c.shape), tgt_shape=np.array(t.shape))


def gen_base(masks):
    t0 = time.time()
    torch.manual_seed(0)
    m = build_ref(BASE_SPEC, 768, 12, 12, 384, 12, 12, top_k=8)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    ref_shapes = synth.jepa_shapes(conv_spec=BASE_SPEC, in_channels=1, d_enc=768, enc_layers=12, d_dec=384,
                                   dec_layers=12, n_tokens=200)
    assert shapes == ref_shapes, "synth.jepa_shapes drifted from the reference state_dict"
    sd = synth.synth_state_dict(shapes, seed=7)
    # teacher slightly different from student
    for k in list(sd):
        if k.startswith("teacher_encoder.") and k.endswith("weight") and sd[k].ndim == 2:
            sd[k] = (sd[k] * np.float32(0.97)).astype(np.float32)
    own = m.state_dict()
    for k, v in sd.items():
        own[k].copy_(torch.from_numpy(v))
    N = 2
    audio = torch.from_numpy(synth.synth_audio(N, 1, 32159, seed=3))
    ctx = torch.from_numpy(masks["as_ctx"][:N])
    tgt = torch.from_numpy(masks["as_tgt"][:N])
    vis = torch.from_numpy(masks["as_vis"][:N])
    m.train()
    out = m(audio, ctx, tgt, vis)
    out["loss"].backward()
    fx = dict(loss=out["loss"].detach().numpy(), n=N, weight_seed=7, audio_seed=3, teacher_scale=0.97)
    fx["local_features_slice"] = out["local_features"][:, ::25, ::64].detach().numpy()
    fx["contextual_features_slice"] = out["contextual_features"][::7, ::16].detach().numpy()
    fx["preds_slice"] = out["preds"][:, ::25, ::64].detach().numpy()
    fx["targets_slice"] = out["targets"][:, ::25, ::64].detach().numpy()
    for k in ("local_features", "contextual_features", "preds", "targets"):
        fx[f"{k}_abs_sum"] = float(out[k].detach().double().abs().sum())
    groups = {"conv": "extract_audio.", "feature_norms": "feature_norms.", "mapper": "post_extraction_mapper.",
              "encoder": "encoder.", "enc2dec": "encoder_to_decoder_mapper.", "decoder": "decoder.",
              "dec2enc": "decoder_to_encoder_mapper.", "mask_token": "mask_token"}
    gn = {}
    for g, pre in groups.items():
        tot = 0.0
        for n, p in m.named_parameters():
            if n.startswith(pre) and p.grad is not None:
                tot += float(p.grad.double().pow(2).sum())
        gn[g] = tot ** 0.5
    fx["grad_group_names"] = np.array(list(gn.keys()))
    fx["grad_group_norms"] = np.array(list(gn.values()))
    fx["grad_slice::encoder.layers.0.linear1.weight"] = m.encoder.layers[0].linear1.weight.grad[::128, ::64].numpy()
    fx["grad_slice::extract_audio.cnn.3.0.weight"] = m.extract_audio.cnn[3][0].weight.grad[::64, ::64, :].numpy()
    fx["grad::extract_audio.cnn.0.0.weight"] = m.extract_audio.cnn[0][0].weight.grad.numpy()
    fx["n_params_total"] = sum(v.numel() for v in m.state_dict().values())
    fx["n_params_trainable"] = sum(p.numel() for p in m.parameters() if p.requires_grad)
    fx["seconds"] = time.time() - t0
    np.savez_compressed(os.path.join(HERE, "base_forward.npz"), **fx)
    return m


def gen_large(masks):
    """The reference's `size="large"` branch (jepa.py:114-118): ViT-Large student (d = 1024, 16 heads, 24 layers, feed-forward 4096) built from the
    BASE layer configs exactly as train.py would with trainer.size=large; predictor unchanged (d = 384 x 12).  Hash-synthesised weights, N = 2,
    fp32 forward + backward on the CPU: names / shapes of the state_dict, loss, tensor slices, per-group gradient norms."""
    t0 = time.time()
    torch.manual_seed(0)
    m = build_ref(BASE_SPEC, 768, 12, 12, 384, 12, 12, top_k=8, size="large")
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    ref_shapes = synth.jepa_shapes(conv_spec=BASE_SPEC, in_channels=1, d_enc=1024, enc_layers=24, d_dec=384, dec_layers=12, n_tokens=200)
    assert shapes == ref_shapes, "synth.jepa_shapes (large) differs from the reference state_dict"
    assert m.n_encoder_heads == 16 and m.encoder_embedding_dim == 1024 and len(m.encoder.layers) == 24
    sd = synth.synth_state_dict(shapes, seed=11)
    for k in list(sd):
        if k.startswith("teacher_encoder.") and k.endswith("weight") and sd[k].ndim == 2:
            sd[k] = (sd[k] * np.float32(0.97)).astype(np.float32)
    own = m.state_dict()
    for k, v in sd.items():
        own[k].copy_(torch.from_numpy(v))
    N = 2
    audio = torch.from_numpy(synth.synth_audio(N, 1, 32159, seed=5))
    ctx, tgt, vis = (torch.from_numpy(masks[k][:N]) for k in ("as_ctx", "as_tgt", "as_vis"))
    m.train()
    out = m(audio, ctx, tgt, vis)
    out["loss"].backward()
    fx = dict(loss=out["loss"].detach().numpy(), n=N, weight_seed=11, audio_seed=5, teacher_scale=0.97, n_tensors=len(shapes),
              enc_heads=m.n_encoder_heads, d_enc=m.encoder_embedding_dim, enc_layers=len(m.encoder.layers))
    fx["local_features_slice"] = out["local_features"][:, ::25, ::64].detach().numpy()
    fx["contextual_features_slice"] = out["contextual_features"][::7, ::16].detach().numpy()
    fx["preds_slice"] = out["preds"][:, ::25, ::64].detach().numpy()
    fx["targets_slice"] = out["targets"][:, ::25, ::64].detach().numpy()
    groups = {"conv": "extract_audio.", "feature_norms": "feature_norms.", "mapper": "post_extraction_mapper.",
              "encoder": "encoder.", "enc2dec": "encoder_to_decoder_mapper.", "decoder": "decoder.",
              "dec2enc": "decoder_to_encoder_mapper.", "mask_token": "mask_token"}
    gn = {}
    for g, pre in groups.items():
        gn[g] = sum(float(p.grad.double().pow(2).sum()) for n, p in m.named_parameters() if n.startswith(pre) and p.grad is not None) ** 0.5
    fx["grad_group_names"] = np.array(list(gn.keys()))
    fx["grad_group_norms"] = np.array(list(gn.values()))
    fx["grad_slice::encoder.layers.23.linear1.weight"] = m.encoder.layers[23].linear1.weight.grad[::256, ::64].numpy()
    fx["grad_slice::encoder.layers.0.self_attn.in_proj_weight"] = m.encoder.layers[0].self_attn.in_proj_weight.grad[::192, ::64].numpy()
    fx["n_params_total"] = sum(v.numel() for v in m.state_dict().values())
    fx["n_params_trainable"] = sum(p.numel() for p in m.parameters() if p.requires_grad)
    fx["seconds"] = time.time() - t0
    np.savez_compressed(os.path.join(HERE, "large_forward.npz"), **fx)
    return m


def gen_base_traj(masks, steps=100):
    """North-star trajectory (BASELINE.json: "JEPA loss within 1e-3 of reference over 100 steps"): the BASE model, N = 4 clips
    per step, the reference's own training_step / EMA / clip / AdamW / per-step cosine schedule (Lightning's order, as
    run_traj), lr 4e-4 with a 10-step warm-up (the stock 100 000-step warm-up would leave lr <= 4e-7 over 100 steps and hide
    optimiser / EMA ordering errors, SURVEY 8a note on a14) and a fast EMA (0.99 -> 0.999 over 50 steps) so the teacher
    moves.  Recorded in fp32 and under torch.autocast("cpu", bfloat16) (train.py precision "bf16-mixed")."""
    fx = dict(steps=steps, n=4, warmup=10, total=200, ema=np.array([0.99, 0.999, 50.0]), weight_seed=7, teacher_scale=0.97,
              audio_seeds=np.array([200, 201, 202, 203]))
    batches = []
    for j in range(4):
        a = torch.from_numpy(synth.synth_audio(4, 1, 32159, seed=200 + j))
        sl = slice(4 * (j % 2), 4 * (j % 2) + 4)
        batches.append((a, torch.from_numpy(masks["as_ctx"][sl]), torch.from_numpy(masks["as_tgt"][sl]), torch.from_numpy(masks["as_vis"][sl])))
    path = os.path.join(HERE, "base_traj.npz")
    tags = [t for t in ("fp32", "bf16") if t in os.environ.get("BASE_TRAJ_TAGS", "fp32,bf16").split(",")]
    if os.path.exists(path) and len(tags) < 2:      # regenerate one precision, keep the other
        old = dict(np.load(path))
        fx.update({k: v for k, v in old.items() if "::" in k and k.split("::")[0] not in tags})
    for tag in tags:
        t0 = time.time()
        torch.manual_seed(0)
        m = build_ref(BASE_SPEC, 768, 12, 12, 384, 12, 12, top_k=8)
        shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        sd = synth.synth_state_dict(shapes, seed=7)
        for k in list(sd):
            if k.startswith("teacher_encoder.") and k.endswith("weight") and sd[k].ndim == 2:
                sd[k] = (sd[k] * np.float32(0.97)).astype(np.float32)
        own = m.state_dict()
        for k, v in sd.items():
            own[k].copy_(torch.from_numpy(v))
        m.hparams["ema_decay"], m.hparams["ema_end_decay"], m.ema_end_step = 0.99, 0.999, 50
        if tag == "bf16":
            bt = [(b[0].to(torch.bfloat16),) + b[1:] for b in batches]     # on_after_batch_transfer hands bf16 audio over
            l, g, e, lr = run_traj(m, bt, steps, 10, 200, autocast_bf16=True)
        else:
            l, g, e, lr = run_traj(m, batches, steps, 10, 200)
        n, s, a = checksums(m)
        fx.update({f"{tag}::loss": l, f"{tag}::gnorm": g, f"{tag}::ema": e, f"{tag}::lr": lr, f"{tag}::names": n, f"{tag}::sum": s,
                   f"{tag}::abs": a, f"{tag}::seconds": time.time() - t0})
        for k in ("encoder.layers.11.linear1.weight", "teacher_encoder.layers.11.linear1.weight", "extract_audio.cnn.2.0.weight",
                  "decoder.layers.0.self_attn.in_proj_weight"):
            fx[f"{tag}::final_slice::{k}"] = m.state_dict()[k].detach().float().numpy().reshape(-1)[::997].copy()
        print(tag, "trajectory", time.time() - t0, "s; loss", l[0], "->", l[-1], flush=True)
        np.savez_compressed(os.path.join(HERE, "base_traj.npz"), **fx)


def gen_channel(masks):
    """WavJEPA-Nat front-end pieces (BASELINE config 4): the reference's ConvChannelFeatureExtractor (per-channel and shared
    stacks) on a 2-channel input, get_binaural_pos_embed, and the channel-based masks of TimeInverseBlockMasker."""
    from wavjepa.extractors.audio_channel_feature_extractor import ConvChannelFeatureExtractor
    spec = [(32, 10, 5)] + [(32, 3, 2)] * 4 + [(32, 2, 2)]
    fx = {}
    torch.manual_seed(2024)
    x = torch.randn(2, 2, 8000)
    fx["audio"] = x.numpy()
    for tag, share in (("own", False), ("shared", True)):
        ext = ConvChannelFeatureExtractor(conv_layers_spec=spec, in_channels=2, share_weights_over_channels=share)
        with torch.no_grad():
            for n, p in ext.named_parameters():
                if n.endswith("2.weight"):
                    p.add_(0.1 * torch.randn_like(p))
                elif n.endswith("2.bias"):
                    p.add_(0.05 * torch.randn_like(p))
            y = ext(x)
        fx.update({f"{tag}::sd::{k}": v.detach().numpy().copy() for k, v in ext.state_dict().items()})
        fx[f"{tag}::out"] = y.numpy()
        import io, contextlib
        with contextlib.redirect_stdout(io.StringIO()):
            fx[f"{tag}::total_patches"] = ext.total_patches(8000)
    for d, t in ((768, 200), (64, 7)):
        tab = R.pos_embed.get_binaural_pos_embed(d, t)
        fx[f"binaural{d}_{t}_shape"] = np.array(tab.shape)
        fx[f"binaural{d}_{t}_slice"] = tab[::max(1, t // 5), ::max(1, d // 16)].copy()
        fx[f"binaural{d}_{t}_sum"] = float(tab.sum())
        fx[f"binaural{d}_{t}_row_last"] = tab[-1].copy()
    with PinnedRng(3000):
        mk = R.masking.TimeInverseBlockMasker(target_masks_per_context=4, context_mask_prob=0.65, context_mask_length=10,
                                              target_prob=0.25, target_length=10, ratio_cutoff=0.1, channel_based_masking=True)
        c, t, v = mk(batch_size=3, n_times=400, in_channels=2)
    fx.update(cb_ctx=c.numpy(), cb_tgt=t.numpy(), cb_vis=v.numpy(), cb_base=3000)
    np.savez_compressed(os.path.join(HERE, "channel_frontend.npz"), **fx)


def gen_misc():
    fx = {}
    for d in (768, 384, 64):
        tab = R.pos_embed.get_1d_sincos_pos_embed_from_grid(d, np.arange(200, dtype=np.float64))
        tab32 = torch.from_numpy(tab).float().numpy()
        fx[f"pos{d}_slice"] = tab32[::13, ::17]
        fx[f"pos{d}_sum"] = float(tab32.astype(np.float64).sum())
        fx[f"pos{d}_row199"] = tab32[199]
    torch.manual_seed(0)
    m = build_ref(TINY_SPEC, 64, 4, 2, 32, 4, 2, top_k=2)
    m.ema_end_step = 100000
    steps = [0, 1, 50000, 99999, 100000, 200000]
    dec = []
    for s in steps:
        m.global_step = s
        dec.append(m._get_ema_decay())
    fx["ema_steps"] = np.array(steps)
    fx["ema_decay"] = np.array(dec, dtype=np.float64)
    opt = torch.optim.AdamW([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
    sch = transformers.get_cosine_schedule_with_warmup(opt, num_warmup_steps=100000, num_training_steps=375000)
    pts = [0, 1, 1000, 99999, 100000, 100001, 237500, 374999, 375000]
    fx["lr_steps"] = np.array(pts)
    fx["lr_lambda"] = np.array([sch.lr_lambdas[0](p) for p in pts], dtype=np.float64)
    for T_sec, expect in ((2.0, None), (2.01, None), (4.01, None), (4.02, None)):
        pass
    ext = R.ConvFeatureExtractor(conv_layers_spec=list(TINY_SPEC), in_channels=1)
    lens = [16000, 32000, 32159, 32160, 64160, 64320]
    fx["patch_lens"] = np.array(lens)
    fx["patch_counts"] = np.array([ext.total_patches(L) for L in lens])
    np.savez_compressed(os.path.join(HERE, "misc.npz"), **fx)


def gen_scene():
    """Scene augmentation: the reference's own generate_scenes_batch.py on seeded inputs, all four cases.
    torchaudio is not installed; its `functional.fftconvolve` is supplied per torchaudio's published definition
    (irfft(rfft(x, n) * rfft(y, n), n), n = len(x) + len(y) - 1, mode "full") -- everything else is the reference's code."""
    import importlib.util
    RI.install_stubs()

    def fftconvolve(x, y, mode="full"):
        assert mode == "full"
        n = x.size(-1) + y.size(-1) - 1
        return torch.fft.irfft(torch.fft.rfft(x, n=n) * torch.fft.rfft(y, n=n), n=n)

    sys.modules["torchaudio"].functional.fftconvolve = fftconvolve
    spec = importlib.util.spec_from_file_location("ref_scene", os.path.join(RI.REFERENCE_ROOT, "data_modules", "scene_module",
                                                                            "generate_scenes_batch.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    B, T, L, n = 3, 6000, 700, 2
    g = torch.Generator().manual_seed(20)
    src = torch.randn(B, T, generator=g)
    noise = torch.randn(B, T, generator=g)
    srir = torch.randn(B, 2, L, generator=g) * torch.exp(-torch.arange(L) / 100.0)
    nrir = torch.randn(B, n, 2, L, generator=g) * torch.exp(-torch.arange(L) / 150.0)
    length = torch.tensor([3000, 6000, 1500])
    start = torch.tensor([100, 0, 2000])
    snr = torch.tensor([5.0, 0.0, -3.0])
    fx = dict(source=src.numpy(), noise=noise.numpy(), source_rir=srir.numpy(), noise_rirs=nrir.numpy(), length=length.numpy(),
              start=start.numpy(), snr=snr.numpy())
    fx["conv"] = ref.convolve_with_rir(src, srir).numpy()
    fx["agg"] = ref.aggregate_noise(nrir, noise).numpy()
    fx["case_rir_noise"] = ref.generate_scene(srir, nrir, src, noise, length, start, snr).numpy()
    fx["case_rir_only"] = ref.generate_scene(srir, nrir, src, [None], length, start, snr).numpy()
    fx["case_noise_only"] = ref.generate_scene([None], nrir, src.unsqueeze(1), noise.unsqueeze(1), length, start, snr).numpy()
    fx["mix_scalar"] = ref.add_noise(src[:1].unsqueeze(1), noise[:1].unsqueeze(1), 7.5, 500, 2500).numpy()   # scalar snr: B = 1 only (:140)
    np.savez_compressed(os.path.join(HERE, "scene.npz"), **fx)


def gen_denoiser():
    """Denoiser stage: the reference's own wavjepa/denoiser.py `Denoiser.forward` (fp32, alpha 0.3) with a small frozen
    reference JEPA as teacher -- losses, contextual features, every parameter-gradient norm and three full gradients.
    (torchaudio / webdataset are import-time stand-ins only; nothing numeric comes from them on this path.)"""
    RI.install_stubs()
    import wavjepa.denoiser as ref_den
    torch.manual_seed(77)
    spec = list(TINY_SPEC)
    teacher = build_ref(spec, 64, 2, 2, 32, 1, 2, top_k=2)
    ext = R.ConvFeatureExtractor(conv_layers_spec=list(spec), in_channels=1)
    den = ref_den.Denoiser(feature_extractor=ext, transformer_encoder_layers_cfg=R.TransformerLayerCFG.create(d_model=64, nhead=2),
                           transformer_encoder_cfg=R.TransformerEncoderCFG.create(num_layers=2), alpha=0.3, process_audio_seconds=2.01)
    with torch.no_grad():
        for n, p in list(den.named_parameters()) + list(teacher.named_parameters()):
            if n.endswith("bias"):
                p.add_(0.02 * torch.randn_like(p))
            elif "norm" in n and n.endswith("weight") or n.endswith("cnn.0.2.weight"):
                p.add_(0.1 * torch.randn_like(p))
    for p in teacher.parameters():
        p.requires_grad = False
    teacher.eval()
    den.teacher = teacher
    N = 2
    clean = torch.randn(N, 1, 32159)
    generated = clean + 0.5 * torch.randn(N, 1, 32159)
    fx = {f"sd::{k}": v for k, v in sd_numpy(den).items() if not k.startswith("teacher.")}
    fx.update({f"tsd::{k}": v for k, v in sd_numpy(teacher).items()})
    fx.update(clean=clean.numpy(), generated=generated.numpy(), alpha=np.float64(0.3))
    den.train()
    out = den(generated, clean)
    out["loss"].backward()
    for k in ("loss", "loss_clean", "loss_denoise_dereverb"):
        fx[f"out::{k}"] = out[k].detach().float().numpy()
    gn = {n: float(p.grad.norm()) for n, p in den.named_parameters() if p.grad is not None and not n.startswith("teacher.")}
    fx["grad_names"] = np.array(list(gn.keys()))
    fx["grad_norms"] = np.array(list(gn.values()), dtype=np.float64)
    fx["grad::extract_audio.cnn.0.0.weight"] = den.extract_audio.cnn[0][0].weight.grad.numpy()
    fx["grad::encoder.layers.1.linear1.weight"] = den.encoder.layers[1].linear1.weight.grad.numpy()
    fx["grad::encoder.norm.weight"] = den.encoder.norm.weight.grad.numpy()
    fx["state_dict_names"] = np.array([k for k in den.state_dict().keys() if not k.startswith("teacher.")])
    np.savez_compressed(os.path.join(HERE, "denoiser.npz"), **fx)


def gen_hear_scene():
    """Evaluation-time twin (reference hear_api/heaRIR/scene_module/generate_scenes.py), same fftconvolve stand-in as gen_scene."""
    import importlib.util
    RI.install_stubs()

    def fftconvolve(x, y, mode="full"):
        n = x.size(-1) + y.size(-1) - 1
        return torch.fft.irfft(torch.fft.rfft(x, n=n) * torch.fft.rfft(y, n=n), n=n)

    sys.modules["torchaudio"].functional.fftconvolve = fftconvolve
    spec = importlib.util.spec_from_file_location("ref_hear_scene", os.path.join(RI.REFERENCE_ROOT, "hear_api", "heaRIR", "scene_module",
                                                                                 "generate_scenes.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    g = torch.Generator().manual_seed(31)
    sr, T, L = 16000, 9000, 800
    src = torch.randn(T, generator=g)
    noise_same = torch.randn(T, generator=g)
    noise_long = torch.randn(T + 2500, generator=g)
    noise_short = torch.randn(5000, generator=g)
    srir = torch.randn(2, L, generator=g) * torch.exp(-torch.arange(L) / 90.0)
    nrirs = [torch.randn(2, L, generator=g) * torch.exp(-torch.arange(L) / 140.0) for _ in range(2)]
    fx = dict(source=src.numpy(), noise_same=noise_same.numpy(), noise_long=noise_long.numpy(), noise_short=noise_short.numpy(),
              source_rir=srir.numpy(), noise_rir0=nrirs[0].numpy(), noise_rir1=nrirs[1].numpy(), sr=np.int64(sr))
    fx["conv"] = ref.convolve_with_rir(src, srir).numpy()
    fx["conv_1d_rir"] = ref.convolve_with_rir(src, srir[0]).numpy()
    w, n2 = torch.randn(3, 4000, generator=g), torch.randn(3, 4000, generator=g)
    snr3, len3 = torch.tensor([3.0, -2.0, 15.0]), torch.tensor([4000, 1000, 2500])
    fx.update(mix_w=w.numpy(), mix_n=n2.numpy(), mix_snr=snr3.numpy(), mix_len=len3.numpy())
    fx["mix_full"] = ref.add_noise(w, n2, snr3).numpy()
    fx["mix_lengths"] = ref.add_noise(w, n2, snr3, len3).numpy()
    fx["fade_long"] = ref.fade_noise(noise_long.clone(), src, sr).numpy()
    fx["fade_short"] = ref.fade_noise(noise_short.clone(), src, sr).numpy()
    fx["scene_same"] = ref.generate_scene(srir, nrirs, src.clone(), noise_same.clone(), 7.0, sr).numpy()
    fx["scene_long"] = ref.generate_scene(srir, nrirs, src.clone(), noise_long.clone(), 0.0, sr).numpy()
    np.random.seed(5)
    fx["scene_short"] = ref.generate_scene(srir, nrirs, src.clone(), noise_short.clone(), 12.0, sr).numpy()
    fx["scene_no_noise"] = ref.generate_scene(srir, [], src.clone(), None, 5.0, sr).numpy()
    np.savez_compressed(os.path.join(HERE, "hear_scene.npz"), **fx)


def gen_hear_helpers():
    """The HEAR wrapper's pure helpers as the reference computes them (hear_api/runtime.py:12-35,145-155 imported through the stub
    recipe; utils.py:1-43 for the run-identity strings): padding mask + cut-off over a sweep of clip lengths for the three window set-ups
    in use, timestamps, window normalisation."""
    import importlib
    import importlib.util
    RI.install_stubs()
    if RI.REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, RI.REFERENCE_ROOT)
    ref = importlib.import_module("hear_api.runtime")

    class M:
        device = torch.device("cpu")

    cases, cuts, lens, trues, trailing = [], [], [], [], []
    for unit, steps, ps in ((32159, 200, 2), (64319, 200, 4), (16000, 100, 1), (32000, 200, 2)):
        for n in list(range(1000, 200000, 3777)) + [unit, 2 * unit, unit - 1, unit + 1]:
            pad = unit - (n % unit)
            mask, cut = ref.calculate_padding_mask(pad, n + pad, 16000, steps, ps, M(), 2)
            cases.append((unit, steps, ps, n))
            cuts.append(cut)
            lens.append(mask.shape[1])
            trues.append(int(mask[0].sum()))
            k = int(mask[0].sum())
            trailing.append(bool(mask[0, mask.shape[1] - k:].all()) and bool(torch.equal(mask[0], mask[1])))
    fx = dict(cases=np.array(cases, np.int64), cut=np.array(cuts, np.int64), mask_len=np.array(lens, np.int64),
              mask_true=np.array(trues, np.int64), mask_is_trailing=np.array(trailing))
    fx["ts_50000_137"] = ref.get_timestamps(16000, 2, 50000, torch.zeros(2, 137, 8)).numpy()
    g = torch.Generator().manual_seed(3)
    win = torch.randn(2, 2, 4000, generator=g) * 3.0 + 0.7
    fx["norm_in"], fx["norm_out"] = win.numpy(), ref.normalize(win).numpy()
    # feature_helper.FeatureExtractor._wav2feature (CPU part of the reference's pre-processing): loudness + channel fixing
    fh = importlib.import_module("hear_api.feature_helper")
    clips = {1: torch.randn(2, 1, 3000, generator=g) * 0.2, 2: torch.randn(2, 2, 3000, generator=g) * 0.05, 4: torch.randn(2, 4, 3000, generator=g)}
    for c_in, batch in clips.items():
        fx[f"feat_in_{c_in}"] = batch.numpy()
        for c_out in (1, 2, 4):
            if (c_in, c_out) in ((2, 4),):
                continue                                     # upstream raises for stereo -> 4 channels
            fx[f"feat_{c_in}_to_{c_out}"] = fh.FeatureExtractor(in_channels=c_out)._wav2feature(batch).numpy()
    flat = torch.randn(2, 3000, generator=g) * 0.3           # [B, n]: 1-D clips
    fx["feat_in_flat"], fx["feat_flat_to_2"] = flat.numpy(), fh.FeatureExtractor(in_channels=2)._wav2feature(flat).numpy()
    tposed = torch.randn(1, 3000, 2, generator=g) * 0.1      # [n, channels] clips are transposed when n > 100
    fx["feat_in_tposed"], fx["feat_tposed_to_2"] = tposed.numpy(), fh.FeatureExtractor(in_channels=2)._wav2feature(tposed).numpy()
    fx["feat_silent_to_1"] = fh.FeatureExtractor(in_channels=1)._wav2feature(torch.zeros(1, 1, 500)).numpy()
    spec = importlib.util.spec_from_file_location("ref_root_utils", os.path.join(RI.REFERENCE_ROOT, "utils.py"))
    ru = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ru)
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from wavjepa_amd.config import load_config
    croot = os.path.join(os.path.dirname(os.path.dirname(HERE)), "configs")
    fx["identity_base"] = np.array(ru.get_identity_from_cfg(load_config(croot, [])))
    fx["identity_librispeech_bs16"] = np.array(ru.get_identity_from_cfg(load_config(croot, ["masker=LibriSpeech", "trainer.batch_size=16"])))
    fx["identity_denoise"] = np.array(ru.get_identity_from_cfg_denoise(load_config(croot, [], config_name="denoise")))
    np.savez_compressed(os.path.join(HERE, "hear_helpers.npz"), **fx)


def gen_dataset_functions():
    """data_modules/dataset_functions.py of the reference on seeded clips (pre_process, pre_process_noise, instance_normalize,
    pad_or_truncate(_batch)): inputs are regenerated from the seed in the test, outputs stored."""
    import importlib.util
    RI.install_stubs()
    spec = importlib.util.spec_from_file_location("ref_dataset_functions", os.path.join(RI.REFERENCE_ROOT, "data_modules", "dataset_functions.py"))
    R = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(R)
    g = torch.Generator().manual_seed(77)
    fx = {}
    for n in (1000, 160000, 170001):
        w = torch.randn(n, generator=g) * 0.3
        fx[f"pre_process_{n}"] = R.pre_process(w, 16000).numpy()[:, ::97]
        fx[f"pre_process_noise_{n}"] = R.pre_process_noise(w).numpy()[..., ::97]
        fx[f"instance_normalize_{n}"] = R.instance_normalize(w).numpy()[..., ::97]
        fx[f"shape_pre_process_{n}"] = np.array(R.pre_process(w, 16000).shape, np.int64)
        fx[f"shape_pre_process_noise_{n}"] = np.array(R.pre_process_noise(w).shape, np.int64)
    f2 = torch.randn(2, 50, generator=g)
    for tl in (30, 50, 70):
        fx[f"pad_or_truncate_{tl}"] = R.pad_or_truncate(f2, tl).numpy()
        fx[f"pad_or_truncate_batch_{tl}"] = R.pad_or_truncate_batch(f2[None], tl).numpy()
    np.savez_compressed(os.path.join(HERE, "dataset_functions.npz"), **fx)


SITE_KEYS = ("data_dirs", "data_dir", "rir_dir", "noise_dir", "save_dir", "teacher_ckpt_weights")     # the authors' cluster paths


def gen_configs():
    """The reference's configs/ tree parsed (yaml.safe_load per file), site-specific path values dropped: the schema and every
    hyper-parameter value a run of the reference would see."""
    import json
    import yaml
    root = os.path.join(RI.REFERENCE_ROOT, "configs")
    tree = {}
    for d, _, files in os.walk(root):
        for f in sorted(files):
            if f.endswith(".yaml"):
                doc = yaml.safe_load(open(os.path.join(d, f)))
                tree[os.path.relpath(os.path.join(d, f), root)] = {k: ("<site path>" if k in SITE_KEYS else v) for k, v in doc.items()}
    with open(os.path.join(HERE, "configs_ref.json"), "w") as fh:
        json.dump(tree, fh, indent=1, sort_keys=True)


def gen_signatures():
    """Constructor / call signatures of the reference's public classes on and around the path (parameter names in order + defaults that
    JSON can hold): the drop-in surface a caller of the reference relies on."""
    import importlib
    import json
    RI.install_stubs()
    if RI.REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, RI.REFERENCE_ROOT)
    targets = {
        "JEPA.__init__": ("wavjepa.jepa", "JEPA", "__init__"), "JEPA.forward": ("wavjepa.jepa", "JEPA", "forward"),
        "JEPA.get_audio_representation": ("wavjepa.jepa", "JEPA", "get_audio_representation"),
        "JEPA.training_step": ("wavjepa.jepa", "JEPA", "training_step"),
        "Denoiser.__init__": ("wavjepa.denoiser", "Denoiser", "__init__"), "Denoiser.forward": ("wavjepa.denoiser", "Denoiser", "forward"),
        "ConvFeatureExtractor.__init__": ("wavjepa.extractors.audio_feature_extractor", "ConvFeatureExtractor", "__init__"),
        "ConvChannelFeatureExtractor.__init__": ("wavjepa.extractors.audio_channel_feature_extractor", "ConvChannelFeatureExtractor", "__init__"),
        "TimeInverseBlockMasker.__init__": ("wavjepa.masking", "TimeInverseBlockMasker", "__init__"),
        "TimeInverseBlockMasker.__call__": ("wavjepa.masking", "TimeInverseBlockMasker", "__call__"),
        "SpeechMasker.__init__": ("wavjepa.masking", "SpeechMasker", "__init__"),
        "RuntimeJEPA.__init__": ("hear_api.runtime", "RuntimeJEPA", "__init__"),
        "RuntimeNatJEPA.__init__": ("hear_api.runtime_natjepa", "RuntimeNatJEPA", "__init__"),
        "WebAudioDataModule.__init__": ("data_modules.WebAudioDataModule", "WebAudioDataModule", "__init__"),
        "WebAudioDataModuleDenoiser.__init__": ("data_modules.WebAudioDataModuleDenoiser", "WebAudioDataModuleDenoiser", "__init__"),
    }
    out = {}
    for name, (mod, cls, fn) in targets.items():
        f = getattr(getattr(importlib.import_module(mod), cls), fn)
        params = []
        for p in inspect.signature(f).parameters.values():
            d = p.default
            if d is inspect.Parameter.empty:
                d = "<required>"
            elif not isinstance(d, (int, float, str, bool, type(None), list, tuple)):
                d = "<object>"
            params.append([p.name, str(p.kind).split(".")[-1], list(d) if isinstance(d, tuple) else d])
        out[name] = params
    with open(os.path.join(HERE, "signatures_ref.json"), "w") as fh:
        json.dump(out, fh, indent=1)


def gen_hear_runtime():
    """The reference's own RuntimeJEPA (hear_api/runtime.py:38-145) on the CPU in fp32: base model, weights = synth_state_dict(seed 23) as
    the GPU test builds them, two clips of 50 000 samples -> 2 windows.  Only change for the run: FeatureExtractor.forward's `.cuda()`
    (feature_helper.py:86-88) is skipped -- the container has no GPU.  Stored: embeddings on a (step, channel) sub-grid, timestamps,
    scene embeddings."""
    import importlib
    RI.install_stubs()
    if RI.REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, RI.REFERENCE_ROOT)
    ref_rt = importlib.import_module("hear_api.runtime")
    fh = importlib.import_module("hear_api.feature_helper")
    from wavjepa.extractors import ConvFeatureExtractor
    fh.FeatureExtractor.forward = lambda self, x: self._wav2feature(x)
    ext = ConvFeatureExtractor(conv_layers_spec=[(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512, 2, 2)], in_channels=1)
    rt = ref_rt.RuntimeJEPA(in_channels=1, weights={"state_dict": {}}, is_spectrogram=False, process_seconds=2.01, extractor=ext,
                            model_size="base", sr=16000)
    own = rt.model.state_dict()
    shapes = {k: tuple(v.shape) for k, v in own.items()}
    sd = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, seed=23).items()}
    for k in ("pos_encoding_encoder", "pos_encoding_decoder"):
        sd[k] = own[k].detach().clone()
    rt.model.load_state_dict(sd)
    rt.model.float()
    wave = torch.from_numpy(synth.synth_audio(2, 1, 50000, seed=29 + 50000)).float()[:, 0]
    with torch.no_grad():
        emb, ts = rt.get_timestamp_embeddings(wave)
        scene = rt.get_scene_embeddings(wave)
    fx = dict(n_samples=np.int64(50000), seed_weights=np.int64(23), seed_audio=np.int64(29 + 50000), emb_shape=np.array(emb.shape, np.int64),
              emb_sub=emb[:, ::7, ::5].numpy(), emb_norm=np.float64(emb.double().norm()), ts=ts.numpy(), scene=scene.numpy())
    np.savez_compressed(os.path.join(HERE, "hear_runtime.npz"), **fx)


if __name__ == "__main__":
    which = sys.argv[1:] or ["masks", "tiny", "traj", "crops", "misc", "base", "channel", "scene", "denoiser", "hear_scene", "hear_helpers", "dataset_functions", "configs", "signatures"]
    masks = gen_masks() if "masks" in which else dict(np.load(os.path.join(HERE, "masks.npz")))
    if "tiny" in which:
        gen_tiny(masks)
    if "traj" in which:
        gen_traj(masks)
    if "crops" in which:
        gen_crops()
    if "misc" in which:
        gen_misc()
    if "base" in which:
        gen_base(masks)
    if "channel" in which:
        gen_channel(masks)
    if "scene" in which:
        gen_scene()
    if "denoiser" in which:
        gen_denoiser()
    if "hear_scene" in which:
        gen_hear_scene()
    if "hear_helpers" in which:
        gen_hear_helpers()
    if "dataset_functions" in which:
        gen_dataset_functions()
    if "configs" in which:
        gen_configs()
    if "signatures" in which:
        gen_signatures()
    if "hear_runtime" in which:       # ~1 min of CPU (base model, 2 windows x 2 clips): not part of the default list
        gen_hear_runtime()
    if "large" in which:              # ~2 min of CPU, 8 GB: not part of the default list
        gen_large(masks)
    if "base_traj" in which:          # ~15 min of CPU: not part of the default list
        gen_base_traj(masks)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")
