"""End-to-end parity of the HIP path (wavjepa_amd.JEPA through the C ABI) against the oracle on the same seeded
inputs (GPU only).  Tolerances: loss within 1e-3 relative of the oracle's bf16 mode (the north-star criterion);
bf16 activations within 1.25 x the stock-autocast distance (ACT_TOL); mask gather bit-exact; parameter gradients within 1e-2 (BASE) / 1.5e-2 relative L2 per
tensor group (two independent bf16 pipelines)."""
import os

import numpy as np
import pytest
import torch

from tests import launch
from tests import parity_yardstick as Y
import synth
from oracle import jepa_oracle as J

pytestmark = pytest.mark.gpu

SMALL_SPEC = [(64, 10, 5)] + [(64, 3, 2)] * 4 + [(64, 2, 2)]
SMALL = dict(conv_spec=SMALL_SPEC, d_enc=128, h_enc=2, l_enc=2, d_dec=64, h_dec=2, l_dec=2, top_k=2)
SMALL16 = dict(SMALL, h_enc=4, h_dec=4)      # BASELINE config 1's head layout: student 4 x 32, predictor 4 x 16
BASE = dict(conv_spec=list(J.WAVJEPA_CONV_SPEC), d_enc=768, h_enc=12, l_enc=12, d_dec=384, h_dec=12, l_dec=12, top_k=8)
# the reference's size="large" branch (jepa.py:114-118): the constructor gets the BASE layer configs and overrides the student to ViT-Large
LARGE = dict(BASE, d_enc=1024, h_enc=16, l_enc=24)


def dev():
    return torch.device("cuda:0")



def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def build(cfg, seed=7, teacher_scale=0.97, seconds=2.01, tokens=200, in_channels=1, channel_stacks=None, **kw):
    from wavjepa_amd.extractors import ConvChannelFeatureExtractor, ConvFeatureExtractor
    from wavjepa_amd.jepa import JEPA
    from wavjepa_amd.types import TransformerEncoderCFG, TransformerLayerCFG
    ctor = cfg
    if kw.get("size") == "large":       # as train.py builds it: base layer configs in, the size switch widens the student
        ctor = dict(cfg, d_enc=768, h_enc=12, l_enc=12)
    if channel_stacks is None:
        ext = ConvFeatureExtractor(conv_layers_spec=cfg["conv_spec"], in_channels=in_channels)
    else:   # "own" / "shared": every channel through a mono stack (WavJEPA-Nat, BASELINE config 4)
        ext = ConvChannelFeatureExtractor(conv_layers_spec=cfg["conv_spec"], in_channels=in_channels,
                                          share_weights_over_channels=channel_stacks == "shared")
    m = JEPA(feature_extractor=ext,
             transformer_encoder_cfg=TransformerEncoderCFG.create(num_layers=ctor["l_enc"]),
             transformer_encoder_layers_cfg=TransformerLayerCFG.create(d_model=ctor["d_enc"], nhead=ctor["h_enc"]),
             transformer_decoder_cfg=TransformerEncoderCFG.create(num_layers=cfg["l_dec"]),
             transformer_decoder_layers_cfg=TransformerLayerCFG.create(d_model=cfg["d_dec"], nhead=cfg["h_dec"]),
             lr=4e-4, adam_betas=(0.9, 0.98), adam_weight_decay=0.04, average_top_k_layers=cfg["top_k"],
             process_audio_seconds=seconds, nr_samples_per_audio=2, **kw)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd = synth.synth_state_dict(shapes, seed=seed)
    for k in list(sd):
        if k.startswith("teacher_encoder.") and k.endswith("weight") and sd[k].ndim == 2:
            sd[k] = (sd[k] * np.float32(teacher_scale)).astype(np.float32)
    sd = {k: torch.from_numpy(v) for k, v in sd.items()}
    sd["pos_encoding_encoder"] = J.sincos_positions(cfg["d_enc"], tokens)
    sd["pos_encoding_decoder"] = J.sincos_positions(cfg["d_dec"], tokens)
    m.load_state_dict(sd)
    P = {k: v.clone().to(dev()) for k, v in sd.items()}
    return m.to(dev()), P


def masks(golden_dir, n):
    fx = dict(np.load(os.path.join(golden_dir, "masks.npz")))
    if n > len(fx["as_ctx"]):
        # more clips than the reference-generated fixture holds: the oracle's masker (pinned bit for bit to the reference's by
        # tests/test_oracle_golden.py::test_masks_bit_exact) under a seeded generator sequence, AudioSet settings
        from oracle import masking_oracle as MO
        rng = np.random.default_rng(1234)
        return tuple(torch.from_numpy(a) for a in MO.time_inverse_block_masks(
            n, 200, 1, new_rng=lambda: np.random.default_rng(rng.integers(1 << 31))))
    return tuple(torch.from_numpy(fx[k][:n]) for k in ("as_ctx", "as_tgt", "as_vis"))


class PinnedRng:
    """The maskers draw from np.random.default_rng(None) -- OS entropy, as upstream -- on every call: inside this context the k-th
    default_rng() call returns default_rng(base + k), so that a parity bound is checked on the same masks in every run (with fresh masks
    per run the noisiest gradient group of the channel-extractor test moved between 1.6 % and 2.6 %)."""
    def __init__(self, base):
        self.base, self.k, self.orig = base, 0, np.random.default_rng

    def __call__(self, seed=None):
        g = self.orig(self.base + self.k)
        self.k += 1
        return g

    def __enter__(self):
        np.random.default_rng = self
        return self

    def __exit__(self, *a):
        np.random.default_rng = self.orig


def oracle_kw(cfg):
    return dict(spec=cfg["conv_spec"], enc_heads=cfg["h_enc"], dec_heads=cfg["h_dec"], top_k=cfg["top_k"])


def channel_group_of(name):
    """group_of, with every channel's conv stack of a ConvChannelFeatureExtractor as its own group"""
    return ".".join(name.split(".")[:3]) if name.startswith("extract_audio.cnns") else group_of(name)


def group_of(name):
    for g in ("extract_audio", "feature_norms", "post_extraction_mapper", "encoder_to_decoder_mapper",
              "decoder_to_encoder_mapper", "encoder", "decoder", "mask_token"):
        if name.startswith(g):
            return g
    return "other"


# Tolerances (relative L2), anchored on the yardstick test below: PyTorch's own bf16 autocast sits 0.61 / 0.69 / 0.65 % from the fp32
# truth on local / contextual features / targets of the BASE model (profiles/r03_parity_report.log); two independent bf16 pipelines
# (HIP path vs the oracle's bf16 flow) measure 0.65-0.69 % on BASE and 0.69-0.87 % on the small model.  Bounds = 1.25 x those
# distances; gradient groups: <= 1e-2 on BASE (measured <= 0.48 %), <= 1.5e-2 on the small models (<= 0.74 %).  A regression that
# doubles any error fails.
ACT_TOL = {"base": dict(local_features=8.6e-3, targets=8.6e-3, contextual_features=8.6e-3, preds=7.0e-3),
           "small": dict(local_features=9.2e-3, targets=9.6e-3, contextual_features=1.08e-2, preds=7.5e-3)}
ACT_TOL["small16"] = ACT_TOL["small"]
ACT_TOL["large"] = ACT_TOL["base"]
GRAD_TOL = {"base": 1.0e-2, "small": 1.5e-2, "small16": 1.5e-2, "large": 1.0e-2}


@pytest.mark.parametrize("cfg_name,n,ragged", [("small", 4, True), ("small", 4, False), ("small", 1, True), ("small", 1, False),
                                                ("small16", 4, True), ("small16", 4, False),  # 16-wide predictor heads (config 1)
                                                ("base", 2, True), ("base", 2, False),       # n = 1: a single clip (fewer rows than one GEMM tile)
                                                ("base", 64, True),
                                                ("large", 2, True), ("large", 2, False)])    # size="large": d = 1024, 16 x 64 heads, 24 layers
def test_forward_backward_parity(golden_dir, cfg_name, n, ragged):
    """ragged = visible-token execution (the default); dense = the reference's key-masked full-length shapes.  Both must
    give the oracle's loss, outputs and gradients.
    base-64: the size at which the headline step's kernels run -- teacher M = 12 800 rows = 450 / 600 / 150 output tiles (the
    persistent eight-phase GEMM: in_proj, linear1 + GELU, out_proj / linear2 stay on the one-tile schedules), predictor
    M ~ 21 600 (persistent in_proj / linear1 / MUL_GELU_GRAD, half-width N = 384 items), grouped weight gradients at their real
    split factors, ragged arena sized from the row counts -- end to end against the oracle, not only op by op."""
    cfg = {"small": SMALL, "small16": SMALL16, "base": BASE, "large": LARGE}[cfg_name]
    m, P = build(cfg, **(dict(size="large") if cfg_name == "large" else {}))
    if cfg_name == "large":
        assert (m.encoder_embedding_dim, m.n_encoder_heads, m.encoder.num_layers) == (1024, 16, 24)
    m._ensure_engine().ragged = ragged
    ctx, tgt, vis = masks(golden_dir, n)
    audio = torch.from_numpy(synth.synth_audio(n, 1, 32159, seed=3)).to(torch.bfloat16).to(dev())
    out = m(audio, ctx, tgt, vis)
    assert m._engine.ragged_step == ragged
    names = J.trainable_names(P)
    for k in names:
        P[k].requires_grad_(True)
    ref = J.jepa_forward(P, audio, ctx.to(dev()), tgt.to(dev()), vis.to(dev()), mode="bf16", **oracle_kw(cfg))
    ref32 = J.jepa_forward({k: v.detach() for k, v in P.items()}, audio.float(), ctx.to(dev()), tgt.to(dev()), vis.to(dev()),
                           mode="fp32", **oracle_kw(cfg))
    report = {k: rel(out[k].float(), ref[k].float()) for k in ("local_features", "contextual_features", "targets")}
    # predictor rows whose output exists: on a ragged step the target rows (the last layer drops the context rows after its
    # attention: nothing reads their outputs); on a dense step every row
    seen = (tgt if ragged else ~vis).reshape(-1, vis.shape[-1]).to(dev())
    assert out["preds"].shape == ref["preds"].shape
    report["preds"] = rel(out["preds"][seen].float(), ref["preds"][seen].float())
    if ragged:
        assert float(out["preds"][~seen].float().abs().max()) == 0.0
    else:
        report["preds_all"] = rel(out["preds"].float(), ref["preds"].float())
        assert report["preds_all"] < ACT_TOL[cfg_name]["preds"]
    lo, lr_, l32 = float(out["loss"]), float(ref["loss"]), float(ref32["loss"])
    print(cfg_name, "rel errors vs oracle bf16:", report, "loss hip/oracle-bf16/oracle-fp32:", lo, lr_, l32)
    assert out["local_features"].dtype == torch.float32 and out["preds"].dtype == torch.bfloat16
    assert out["contextual_features"].shape == ref["contextual_features"].shape
    # distance of each bf16 pipeline from the fp32 truth (same weights, same clips): the HIP path must be as close to it as the oracle's
    # bf16 flow is.  Two INDEPENDENT bf16 pipelines then sit up to sqrt(2) x that distance apart (measured at 64 clips: 0.86 % between
    # them with both 0.61-0.63 % from fp32; at 2 clips the rounding errors happen to correlate and 0.66 % is seen), which is what the
    # fixed bounds of ACT_TOL (1.25 x the 2-clip distances) cannot express for a large batch.
    to32 = {k: (rel(out[k].float(), ref32[k].float()), rel(ref[k].float(), ref32[k].float()))
            for k in ("local_features", "targets", "contextual_features")}
    to32["preds"] = (rel(out["preds"][seen].float(), ref32["preds"][seen].float()), rel(ref["preds"][seen].float(), ref32["preds"][seen].float()))
    print(cfg_name, "distance from the fp32 oracle (HIP, oracle-bf16):", to32)
    for k, (d_hip, d_orc) in to32.items():
        # the yardstick form (tests/parity_yardstick.py), at every batch size: as close to fp32 as the oracle's bf16 flow on this draw,
        # and within the distance of two independent bf16 pipelines from that flow
        assert d_hip < Y.ACT_FACTOR * d_orc + Y.ACT_EPS, (k, d_hip, d_orc)
        assert report[k] < Y.PAIR_FACTOR * d_orc + Y.ACT_EPS, (k, report[k], d_orc)
    if n < 16:
        # small batches also keep the fixed regression bounds anchored on the stock-autocast distances
        for k, bound in ACT_TOL[cfg_name].items():
            assert report[k] < bound, (k, report[k], bound)
    assert abs(lo - lr_) < 1e-3 * abs(lr_), (lo, lr_)          # north-star: loss within 1e-3 rel of the bf16 reference flow
    assert abs(lo - l32) < 2e-2 * abs(l32)
    # backward
    out["loss"].backward()
    ref["loss"].backward()
    got = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    gbf = {k: P[k].grad for k in names}
    errs = Y.group_errors(got, gbf, names, group_of)
    print(cfg_name, "grad rel errors per group:", errs)
    for g, e in errs.items():
        assert e < GRAD_TOL[cfg_name], (g, e)
    # gradient yardstick: per group, HIP is as close to the oracle's fp32 gradient as the oracle's own bf16 flow is on this draw
    _, g32 = Y.oracle_fp32_grads(J, P, audio, ctx.to(dev()), tgt.to(dev()), vis.to(dev()), names, **oracle_kw(cfg))
    table = Y.grad_yardstick(got, gbf, g32, names, group_of)
    print(cfg_name, "grad yardstick (d_hip, d_orc, pair, ratio):", {g: tuple(round(v, 5) for v in r.values()) for g, r in table.items()})
    Y.assert_grad_yardstick(table)


def test_batch_size_changes_between_steps(golden_dir):
    """A loader's last, shorter batch: the arena follows the clip count (4 -> 2 -> 4 clips on one model); every forward / backward must
    equal what a fresh model gives for that batch."""
    m, P = build(SMALL)
    ctx, tgt, vis = masks(golden_dir, 4)

    def step(model, n):
        audio = torch.from_numpy(synth.synth_audio(n, 1, 32159, seed=60 + n)).to(torch.bfloat16).to(dev())
        out = model(audio, ctx[:n], tgt[:n], vis[:n])
        out["loss"].backward()
        torch.cuda.synchronize()
        return float(out["loss"].detach()), model._flat.g32.clone()

    l4, g4 = step(m, 4)
    l2, g2 = step(m, 2)
    l4b, g4b = step(m, 4)
    f2, fg2 = step(build(SMALL)[0], 2)
    assert abs(l2 - f2) < 1e-6 * abs(f2) and rel(g2, fg2) < 1e-5, (l2, f2, rel(g2, fg2))
    assert abs(l4 - l4b) < 1e-6 * abs(l4) and rel(g4b, g4) < 1e-5, (l4, l4b, rel(g4b, g4))
    assert abs(l4 - l2) > 1e-4 * abs(l4)


def test_stock_pytorch_autocast_yardstick_base_model(golden_dir):
    """SURVEY 3.2 / north star ("encoder outputs within 1e-3 rel in bf16"): what does PyTorch's OWN bf16 autocast on this GPU do to the
    same model?  The oracle's fp32 code path is run under `torch.autocast("cuda", dtype=torch.bfloat16)` -- PyTorch's cast policy on
    the box decides every dtype (conv / linear / matmul in bf16, norms / softmax / loss in fp32), not the oracle's emulation of it --
    next to the plain fp32 run (the truth) and the HIP path, base model, same weights and clips.  The HIP path must sit as close to
    the fp32 truth as stock autocast does (x1.25), and within that same distance of stock autocast itself; the loss within 1e-3."""
    m, P = build(BASE)
    ctx, tgt, vis = masks(golden_dir, 2)
    audio = torch.from_numpy(synth.synth_audio(2, 1, 32159, seed=3)).to(torch.bfloat16).to(dev())
    with torch.no_grad():
        out = m(audio, ctx, tgt, vis)
        args = (audio.float(), ctx.to(dev()), tgt.to(dev()), vis.to(dev()))
        Pd = {k: v.detach() for k, v in P.items()}
        ref32 = J.jepa_forward(Pd, *args, mode="fp32", **oracle_kw(BASE))
        with torch.autocast("cuda", dtype=torch.bfloat16):
            stock = J.jepa_forward(Pd, *args, mode="fp32", **oracle_kw(BASE))
    rep = {}
    for k in ("local_features", "contextual_features", "targets"):
        rep[k] = dict(hip_vs_fp32=rel(out[k].float(), ref32[k].float()), stock_autocast_vs_fp32=rel(stock[k].float(), ref32[k].float()),
                      hip_vs_stock_autocast=rel(out[k].float(), stock[k].float()))
    lh, ls, l32 = float(out["loss"]), float(stock["loss"]), float(ref32["loss"])
    print("stock-autocast yardstick (rel L2):", rep, "loss hip / stock autocast / fp32:", lh, ls, l32)
    for k, r in rep.items():
        assert r["hip_vs_fp32"] < 1.25 * r["stock_autocast_vs_fp32"] + 1e-4, (k, r)
        assert r["hip_vs_stock_autocast"] < 1.25 * r["stock_autocast_vs_fp32"] + 1e-4, (k, r)
    assert abs(lh - ls) < 1e-3 * abs(ls), (lh, ls)


def test_forward_backward_parity_speech_masks(golden_dir):
    """The LibriSpeech masker's masks (short, clustered targets; long contexts) through the ragged path."""
    m, P = build(SMALL)
    fx = dict(np.load(os.path.join(golden_dir, "masks.npz")))
    ctx, tgt, vis = (torch.from_numpy(fx[k][:4]) for k in ("ls_ctx", "ls_tgt", "ls_vis"))
    audio = torch.from_numpy(synth.synth_audio(4, 1, 32159, seed=13)).to(torch.bfloat16).to(dev())
    out = m(audio, ctx, tgt, vis)
    assert m._engine.ragged_step
    names = J.trainable_names(P)
    for k in names:
        P[k].requires_grad_(True)
    ref = J.jepa_forward(P, audio, ctx.to(dev()), tgt.to(dev()), vis.to(dev()), mode="bf16", **oracle_kw(SMALL))
    lo, lr_ = float(out["loss"].detach()), float(ref["loss"].detach())
    assert abs(lo - lr_) < 1e-3 * abs(lr_), (lo, lr_)
    out["loss"].backward()
    ref["loss"].backward()
    got = dict(m.named_parameters())
    num, den = {}, {}
    for k in names:
        g = group_of(k)
        a, b = got[k].grad.double(), P[k].grad.double()
        num[g] = num.get(g, 0.0) + float((a - b).pow(2).sum())
        den[g] = den.get(g, 0.0) + float(b.pow(2).sum())
    errs = {g: (num[g] / max(den[g], 1e-300)) ** 0.5 for g in num}
    print("speech masks: loss", lo, lr_, "grad rel errors per group:", errs)
    for g, e in errs.items():
        assert e < GRAD_TOL["small"], (g, e)


@pytest.mark.parametrize("ragged", [True, False])
def test_forward_backward_parity_400_tokens(golden_dir, ragged):
    """4.01 s clips (64 160 samples -> 400 tokens; SURVEY 8(f1) shapes, bf16): teacher attention over 400 tokens, ragged or
    dense student / predictor, against the oracle."""
    m, P = build(SMALL, seconds=4.01, tokens=400)
    assert m.total_patches == 400 and m.target_length == 64160
    m._ensure_engine().ragged = ragged
    fx = dict(np.load(os.path.join(golden_dir, "masks.npz")))
    ctx, tgt, vis = (torch.from_numpy(fx[k][:3]) for k in ("as400_ctx", "as400_tgt", "as400_vis"))
    audio = torch.from_numpy(synth.synth_audio(3, 1, 64160, seed=17)).to(torch.bfloat16).to(dev())
    out = m(audio, ctx, tgt, vis)
    assert m._engine.ragged_step == ragged
    names = J.trainable_names(P)
    for k in names:
        P[k].requires_grad_(True)
    ref = J.jepa_forward(P, audio, ctx.to(dev()), tgt.to(dev()), vis.to(dev()), mode="bf16", **oracle_kw(SMALL))
    lo, lr_ = float(out["loss"].detach()), float(ref["loss"].detach())
    assert abs(lo - lr_) < 1e-3 * abs(lr_), (lo, lr_)
    assert rel(out["targets"].float(), ref["targets"].float()) < ACT_TOL["small"]["targets"]
    out["loss"].backward()
    ref["loss"].backward()
    got = dict(m.named_parameters())
    num, den = {}, {}
    for k in names:
        g = group_of(k)
        a, b = got[k].grad.double(), P[k].grad.double()
        num[g] = num.get(g, 0.0) + float((a - b).pow(2).sum())
        den[g] = den.get(g, 0.0) + float(b.pow(2).sum())
    errs = {g: (num[g] / max(den[g], 1e-300)) ** 0.5 for g in num}
    print("400 tokens, ragged" if ragged else "400 tokens, dense", "loss", lo, lr_, "grad rel errors per group:", errs)
    for g, e in errs.items():
        assert e < GRAD_TOL["small"], (g, e)


def test_forward_backward_parity_seven_layer_wav2vec2_spec(golden_dir):
    """The reference's other extractor config (`configs/extractor/wav2vec2.yaml:1`: seven conv layers, stride 320):
    4.02 s clips -> 200 tokens through the same kernels, ragged student / predictor, against the oracle."""
    spec = [(64, 10, 5)] + [(64, 3, 2)] * 4 + [(64, 2, 2)] * 2
    cfg = dict(SMALL, conv_spec=spec)
    m, P = build(cfg, seconds=4.02, tokens=200)
    assert m.total_patches == 200
    n_samples = m.target_length
    ctx, tgt, vis = masks(golden_dir, 3)
    audio = torch.from_numpy(synth.synth_audio(3, 1, n_samples, seed=23)).to(torch.bfloat16).to(dev())
    out = m(audio, ctx, tgt, vis)
    assert m._engine.ragged_step
    names = J.trainable_names(P)
    for k in names:
        P[k].requires_grad_(True)
    ref = J.jepa_forward(P, audio, ctx.to(dev()), tgt.to(dev()), vis.to(dev()), mode="bf16", **oracle_kw(cfg))
    lo, lr_ = float(out["loss"].detach()), float(ref["loss"].detach())
    assert abs(lo - lr_) < 1e-3 * abs(lr_), (lo, lr_)
    assert rel(out["local_features"].float(), ref["local_features"].float()) < 1e-2        # seven bf16 conv layers deep
    assert rel(out["targets"].float(), ref["targets"].float()) < 1e-2
    out["loss"].backward()
    ref["loss"].backward()
    got = dict(m.named_parameters())
    num, den = {}, {}
    for k in names:
        g = group_of(k)
        a, b = got[k].grad.double(), P[k].grad.double()
        num[g] = num.get(g, 0.0) + float((a - b).pow(2).sum())
        den[g] = den.get(g, 0.0) + float(b.pow(2).sum())
    errs = {g: (num[g] / max(den[g], 1e-300)) ** 0.5 for g in num}
    print("seven-layer spec: loss", lo, lr_, "grad rel errors per group:", errs)
    for g, e in errs.items():
        assert e < GRAD_TOL["small"], (g, e)                     # measured 0.39-0.74 %


def test_forward_backward_parity_two_channel_audio(golden_dir):
    """Binaural input through the 2-channel first conv (`in_channels=2`, 20 taps in conv0)."""
    m, P = build(SMALL, in_channels=2)
    ctx, tgt, vis = masks(golden_dir, 3)
    audio = torch.from_numpy(synth.synth_audio(3, 2, 32159, seed=19)).to(torch.bfloat16).to(dev())
    out = m(audio, ctx, tgt, vis)
    names = J.trainable_names(P)
    for k in names:
        P[k].requires_grad_(True)
    ref = J.jepa_forward(P, audio, ctx.to(dev()), tgt.to(dev()), vis.to(dev()), mode="bf16", **oracle_kw(SMALL))
    lo, lr_ = float(out["loss"].detach()), float(ref["loss"].detach())
    assert abs(lo - lr_) < 1e-3 * abs(lr_), (lo, lr_)
    assert rel(out["local_features"].float(), ref["local_features"].float()) < 1e-2
    out["loss"].backward()
    ref["loss"].backward()
    got = dict(m.named_parameters())
    for k in ("extract_audio.cnn.0.0.weight", "extract_audio.cnn.0.2.weight", "extract_audio.cnn.1.0.weight"):
        e = rel(got[k].grad.float(), P[k].grad.float())
        assert e < 2e-2, (k, e)                                  # single tensors of the conv stack (not a group)


@pytest.mark.parametrize("G", [2, 6])
def test_forward_backward_parity_other_group_counts(golden_dir, G):
    """The number of target groups per clip comes from target_indices.shape[1] (reference jepa.py:402-405;
    masker.target_masks_per_context is a config value), not from a constant: 2 and 6 groups against the oracle, then a
    switch back to 4 on the same module (the arena is rebuilt)."""
    m, P = build(SMALL)
    ctx, tgt, vis = masks(golden_dir, 3)
    sel = [0, 1, 2, 3, 0, 1][:G]
    tgt_g, vis_g = tgt[:, sel].contiguous(), vis[:, sel].contiguous()
    audio = torch.from_numpy(synth.synth_audio(3, 1, 32159, seed=21)).to(torch.bfloat16).to(dev())
    out = m(audio, ctx, tgt_g, vis_g)
    assert m._engine.G == G and out["preds"].shape[0] == 3 * G
    names = J.trainable_names(P)
    for k in names:
        P[k].requires_grad_(True)
    ref = J.jepa_forward(P, audio, ctx.to(dev()), tgt_g.to(dev()), vis_g.to(dev()), mode="bf16", **oracle_kw(SMALL))
    lo, lr_ = float(out["loss"].detach()), float(ref["loss"].detach())
    assert abs(lo - lr_) < 1e-3 * abs(lr_), (lo, lr_)
    out["loss"].backward()
    ref["loss"].backward()
    got = dict(m.named_parameters())
    num, den = {}, {}
    for k in names:
        g = group_of(k)
        a, b = got[k].grad.double(), P[k].grad.double()
        num[g] = num.get(g, 0.0) + float((a - b).pow(2).sum())
        den[g] = den.get(g, 0.0) + float(b.pow(2).sum())
    for g in num:
        assert (num[g] / max(den[g], 1e-300)) ** 0.5 < GRAD_TOL["small"], (G, g)
    out4 = m(audio, ctx, tgt, vis)
    assert m._engine.G == 4 and out4["preds"].shape[0] == 12
    with pytest.raises(ValueError):
        m(audio, ctx[:2], tgt, vis)                                   # masks for another batch size


def test_data_parallel_gradient_semantics_two_micro_batches(golden_dir):
    """What 2 data-parallel ranks compute, on one GPU: two DIFFERENT micro-batches through the engine, their flat gradient buffers
    averaged exactly as FlatGradAllReducer does (ReduceOp.AVG, every rank's loss normalised by its LOCAL target count), against
    the mean of the oracle's two per-rank gradients (reference jepa.py:359-362, train.py:174-179)."""
    m, P = build(SMALL)
    fx = dict(np.load(os.path.join(golden_dir, "masks.npz")))
    names = J.trainable_names(P)
    flat_sum, ref_sum, counts = None, {}, []
    for r, sl in enumerate((slice(0, 3), slice(3, 5))):          # micro-batches of 3 and 2 clips: different target counts
        ctx, tgt, vis = (torch.from_numpy(fx[k][sl]) for k in ("as_ctx", "as_tgt", "as_vis"))
        counts.append(int(tgt.sum()))
        audio = torch.from_numpy(synth.synth_audio(sl.stop - sl.start, 1, 32159, seed=60 + r)).to(torch.bfloat16).to(dev())
        out = m(audio, ctx, tgt, vis)
        out["loss"].backward()
        g = m._flat.g32.double().clone()
        flat_sum = g if flat_sum is None else flat_sum + g
        Pr = {k: v.detach().clone() for k, v in P.items()}
        for k in names:
            Pr[k].requires_grad_(True)
        ref = J.jepa_forward(Pr, audio, ctx.to(dev()), tgt.to(dev()), vis.to(dev()), mode="bf16", **oracle_kw(SMALL))
        ref["loss"].backward()
        for k in names:
            ref_sum[k] = Pr[k].grad.double() + ref_sum.get(k, 0.0)
    assert counts[0] != counts[1]
    avg = flat_sum / 2
    num, den = {}, {}
    for k in names:
        s = m._flat.by_name[k]
        got, want = avg[s.offset:s.offset + s.numel].view(s.shape), ref_sum[k] / 2
        g = group_of(k)
        num[g] = num.get(g, 0.0) + float((got - want).pow(2).sum())
        den[g] = den.get(g, 0.0) + float(want.pow(2).sum())
    for g in num:
        assert (num[g] / max(den[g], 1e-300)) ** 0.5 < GRAD_TOL["small"], (g, (num[g] / den[g]) ** 0.5)


def test_full_size_batch_256_clips_equals_its_slices():
    """BASELINE config 2 at size (WavJEPA-base, 256 clips per GPU) by a test, not only by the bench: one full forward +
    backward at N = 256 is finite, and -- the loss being sum_{targets} mean_d (p - y)^2 / count -- it equals the count-weighted
    combination of the same step run on its four 64-clip slices, and so do the gradients (clips are independent: no batch
    statistic couples them)."""
    from wavjepa_amd.masking import TimeInverseBlockMasker
    m, _ = build(BASE)
    eng = m._ensure_engine()
    N = 256
    with PinnedRng(2560):
        ctx, tgt, vis = TimeInverseBlockMasker(4, 0.65, 10, 0.25, 10, 0.1)(batch_size=N, n_times=200, in_channels=1)
    audio = torch.randn(N, 1, 32159, device=dev(), generator=torch.Generator(device=dev()).manual_seed(3)).to(torch.bfloat16)
    out = m(audio, ctx, tgt, vis)
    out["loss"].backward()
    loss_full, count_full = float(eng.loss[0]), float(eng.loss[1])
    assert np.isfinite(loss_full) and count_full == float(tgt.sum())
    g_full = m._flat.g32.double().clone()
    assert bool(torch.isfinite(g_full).all())
    keys = ("encoder.layers.5.linear1.weight", "decoder.layers.3.self_attn.in_proj_weight", "extract_audio.cnn.3.0.weight",
            "extract_audio.cnn.0.0.weight", "mask_token", "encoder.layers.0.norm1.weight")
    acc, loss_acc = torch.zeros_like(g_full), 0.0
    for i in range(0, N, 64):
        sl = slice(i, i + 64)
        o = m(audio[sl], ctx[sl], tgt[sl], vis[sl])
        o["loss"].backward()
        w = float(eng.loss[1]) / count_full
        loss_acc += float(eng.loss[0]) * w
        acc += m._flat.g32.double() * w
    assert abs(loss_acc - loss_full) < 1e-5 * abs(loss_full), (loss_acc, loss_full)
    for k in keys:
        s = m._flat.by_name[k]
        a, b = acc[s.offset:s.offset + s.numel], g_full[s.offset:s.offset + s.numel]
        assert float((a - b).norm() / (b.norm() + 1e-30)) < 2e-3, k


def test_bench_runs_under_torch_distributed_run_with_one_rank():
    """The driver's N > 1 launch path (python -m torch.distributed.run ... bench.py) with a 1-rank process group: RCCL
    init, parameter broadcast, bucketed all-reduce hooks and the replica checksum all execute on the single GPU box."""
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(launch.free_port()), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--clips-per-gpu", "16",
           "--dense-steps", "1", "--no-cpu-baseline", "--no-profile"]
    rc, out, err = launch.run(cmd, cwd=root, timeout=300)
    assert rc == 0, err[-4000:]
    line = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["replicas_equal"] is True and line["value"] > 0 and line["dense_ms_per_step"] > 0
    assert line["config"]["global_batch"] == 16


def test_bench_one_rank_gradient_buckets_through_the_librarys_rccl_binding():
    """WJ_RCCL_DIRECT=1: the buckets leave through wj_rccl_bucket_allreduce_{launch,wait} (own communicator, own stream) instead
    of torch.distributed.  One rank: the average is the identity, so the run must land on the default transport's loss (same seed)."""
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lines = {}
    for direct in ("0", "1"):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
               "--master-port", str(launch.free_port()), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
               "--clips-per-gpu", "16", "--dense-steps", "0", "--no-cpu-baseline", "--no-profile", "--seed", "4242"]
        rc, out, err = launch.run(cmd, cwd=root, env=dict(os.environ, WJ_RCCL_DIRECT=direct), timeout=300)
        assert rc == 0, err[-4000:]
        lines[direct] = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])
    a, b = lines["0"], lines["1"]
    assert b["replicas_equal"] is True and b["config"]["gradient_transport"] == "wj_rccl_bucket_allreduce" and a["config"]["gradient_transport"] == "torch.distributed"
    assert b["allreduce"] is not None and b["allreduce"]["buckets"] >= 2 and b["allreduce"]["exposed_ms"] >= 0
    # both processes see the same masks, crops and shuffles (--seed): the two transports must land on the same loss up to the fp32-atomic
    # noise of the weight gradients (a bucket that was lost, reduced twice or read before its launch moves the loss of the next step).
    # What this does NOT show: ncclAvg over more than one rank and the comm-stream ordering under a real collective -- a world of one
    # is the identity; that stays unverified until a multi-GPU node runs it.
    assert abs(a["final_loss"] - b["final_loss"]) < 2e-4 * abs(a["final_loss"]), (a["final_loss"], b["final_loss"])
    assert a["param_checksum"][1] == pytest.approx(b["param_checksum"][1], rel=1e-6)


def test_bench_emulated_allreduce_footprint_on_one_gpu():
    """bench.py --emulate-allreduce (one GPU, no process group): the bucket hooks fire from the backward, a paced copy kernel stands in
    for every bucket's all-reduce on a communication stream, the optimiser waits for it; the loss is that of the plain run (the
    stand-in rewrites the gradients unchanged), and the line reports the rehearsal under `allreduce_emulated`, never under `allreduce`."""
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lines = {}
    for emu in (False, True):
        cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--clips-per-gpu", "16", "--dense-steps", "0",
               "--no-cpu-baseline", "--no-profile", "--seed", "77"] + (["--emulate-allreduce"] if emu else [])
        env = dict(os.environ, WJ_EMULATE_BUSBW_GBPS="300")
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            env.pop(k, None)
        rc, out, err = launch.run(cmd, cwd=root, env=env, timeout=300)
        assert rc == 0, err[-4000:]
        lines[emu] = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])
    a, b = lines[False], lines[True]
    assert a["allreduce_emulated"] is None and b["allreduce"] is None
    e = b["allreduce_emulated"]
    assert e["emulation"] is True and e["buckets"] >= 2 and e["workgroups"] == 32 and e["assumed_busbw_gbps"] == 300.0
    assert e["backward_window_ms"] > 0 and e["exposed_ms"] >= 0
    assert abs(a["final_loss"] - b["final_loss"]) < 2e-4 * abs(a["final_loss"]), (a["final_loss"], b["final_loss"])


def test_bench_two_ranks_share_the_gpu_over_gloo():
    """The N = 2 launch of the bench end to end on the 1-GPU box: RCCL refuses two ranks on one device, so the collectives go over
    gloo (WJ_DIST_BACKEND, a development switch of init_distributed); everything else -- per-rank sources, broadcast, bucketed
    averages from the backward's hooks, max-over-ranks timing, whole-job value, the replica checksum -- is the driver's path."""
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(launch.free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--clips-per-gpu", "16",
           "--dense-steps", "0", "--no-cpu-baseline", "--no-profile"]
    rc, out, err = launch.run(cmd, cwd=root, env=dict(os.environ, WJ_DIST_BACKEND="gloo"), timeout=300)
    assert rc == 0, err[-4000:]
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 alone prints the JSON line"
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["replicas_equal"] is True and line["scaling"] == "weak"
    assert line["config"]["global_batch"] == 32 and line["config"]["parallelism"] == "dp2"
    assert abs(line["value"] - 32 / (line["ms_per_step"] * 1e-3)) < 1e-2 * line["value"]      # whole-job clips/s over the slowest rank's time
    # what a SCALE run will be read by: the all-reduce fields are present and sane before such a run exists
    ar = line["allreduce"]
    assert ar is not None and ar["buckets"] >= 2 and ar["steps"] == 3 and 4 * 111012864 <= ar["bytes"] < 4 * 111100000
    assert ar["backward_window_ms"] is not None and ar["backward_window_ms"] > 0       # first bucket issued BEFORE the backward ended
    assert ar["exposed_ms"] >= 0 and ar["exposed_ms"] < 1e4
    assert line["config"]["persistent_gemm_workgroups_per_xcd"] == 28                  # N > 1: CUs left free for the collective
    # every rank reported its stages (rendezvous, group up, broadcast, first reduced step) on stderr
    for msg in ("process group up", "parameters broadcast from rank 0", "first optimisation step (all gradient buckets reduced) done"):
        assert err.count(msg) >= 2, (msg, err[-3000:])


@pytest.mark.parametrize("stacks,ragged", [("own", True), ("shared", True), ("own", False)])
def test_forward_backward_parity_channel_extractor(stacks, ragged):
    """ConvChannelFeatureExtractor (reference extractors/audio_channel_feature_extractor.py:154-179; WavJEPA-Nat, BASELINE config
    4): a 2-channel clip -> every channel through its own (or the shared) mono conv stack -> 2 x 99 tokens flattened channel-major,
    with channel-based masks in the extractor's token order.  Loss, local features and every gradient (each channel's stack
    separately) against the oracle; sparse and dense conv backward."""
    from wavjepa_amd.masking import TimeInverseBlockMasker
    m, P = build(SMALL, seconds=1.0, tokens=198, in_channels=2, channel_stacks=stacks)
    assert m.total_patches == 198 and m.extract_audio.frames_per_channel(16000) == 99
    m._ensure_engine().ragged = ragged
    with PinnedRng(4100):                # reproducibility only: the bounds below are yardstick bounds, valid on any draw
        ctx, tgt, vis = TimeInverseBlockMasker(4, 0.65, 10, 0.25, 10, 0.1, channel_based_masking=True, channel_major=True)(
            batch_size=3, n_times=198, in_channels=2)
    assert ctx.shape == (3, 198) and torch.equal(ctx[:, :99], ctx[:, 99:])
    audio = torch.from_numpy(synth.synth_audio(3, 2, 16000, seed=31)).to(torch.bfloat16).to(dev())
    out = m(audio, ctx, tgt, vis)
    assert m._engine.ragged_step == ragged and m._engine.S == 2 and len(m._engine.stacks) == (2 if stacks == "own" else 1)
    names = J.trainable_names(P)
    for k in names:
        P[k].requires_grad_(True)
    ref = J.jepa_forward(P, audio, ctx.to(dev()), tgt.to(dev()), vis.to(dev()), mode="bf16", **oracle_kw(SMALL))
    lo, lr_ = float(out["loss"].detach()), float(ref["loss"].detach())
    assert abs(lo - lr_) < 1e-3 * abs(lr_), (lo, lr_)
    assert rel(out["local_features"].float(), ref["local_features"].float()) < 1e-2
    assert rel(out["targets"].float(), ref["targets"].float()) < 1e-2
    out["loss"].backward()
    ref["loss"].backward()
    got = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    gbf = {k: P[k].grad for k in names}
    _, g32 = Y.oracle_fp32_grads(J, P, audio, ctx.to(dev()), tgt.to(dev()), vis.to(dev()), names, **oracle_kw(SMALL))
    table = Y.grad_yardstick(got, gbf, g32, names, channel_group_of)          # each stack on its own
    print(stacks, "ragged" if ragged else "dense", "loss", lo, lr_, "grad yardstick (d_hip, d_orc, pair, ratio):",
          {g: tuple(round(v, 5) for v in r.values()) for g, r in table.items()})
    assert sum(1 for g in table if g.startswith("extract_audio.cnns")) == (2 if stacks == "own" else 1)
    # A per-channel conv stack sees half of the tokens of a clip and is the noisiest group: its distance from the oracle's bf16 flow
    # moves between 0.9 % and 2.5 % with the mask draw (the round-4 fixed bound of 2.5e-2 failed on some draws).  That is the
    # oracle's OWN bf16 distance from fp32 on those draws -- profiles/r05_grad_yardstick.txt, 2 x 16 unpinned draws: oracle-bf16 vs
    # fp32 1.0-2.2 %, HIP vs fp32 0.8-2.3 % -- so the bound is stated against it (factor 1.5 for these groups: per draw the two
    # are independent realisations of the same noise, ratio 0.79-1.36; 1.25 for every other group).
    Y.assert_grad_yardstick(table)
    for g, r in table.items():
        if not g.startswith("extract_audio.cnns"):
            assert r["pair"] < GRAD_TOL["small"], (g, r)
    # the stand-alone extractor forward gives the same tokens as the oracle's front-end
    tok = m.extract_audio(audio)
    want = J.conv_frontend({k: v.detach() for k, v in P.items()}, audio, SMALL_SPEC, "bf16")
    # raw conv tokens (before feature_norms): six bf16 layers deep, two independent bf16 pipelines -> 2e-2 (the normalised
    # local_features above meet 1e-2)
    assert tok.shape == want.shape == (3, 198, 64) and rel(tok.float(), want.float()) < 2e-2


def test_fp8_forward_path_base_model_400_tokens_vs_bf16_path():
    """BASELINE config 5 (WavJEPA-base, 4.01 s clips -> 400 tokens, fp8 MFMA on the attention-projection / MLP GEMMs).  The reference
    has no fp8, so the yardstick is this library's own bf16 path on the same weights and inputs (itself checked against the oracle at
    400 tokens by test_forward_backward_parity_400_tokens and, on the base model, by test_forward_backward_parity[base-*]): MX fp8
    (e4m3 elements, one power-of-two scale per 32) perturbs every quantised GEMM operand by <= 2^-4 relative per element.
    Stated tolerances (measured: 1.2e-3 / 3.9e-2 / 3.4-6.6e-2): loss within 1e-2 relative, teacher targets within 8e-2 relative L2,
    parameter-gradient groups within 0.15 relative L2 (the backward differentiates the bf16 graph from fp8-forward activations).
    Also runs the 400-token parity of the bf16 path on the BASE model against the oracle (the 2-layer model has it above)."""
    m, P = build(BASE, seconds=4.01, tokens=400)
    assert m.total_patches == 400
    fx = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "masks.npz")))
    ctx, tgt, vis = (torch.from_numpy(fx[k][:2]) for k in ("as400_ctx", "as400_tgt", "as400_vis"))
    audio = torch.from_numpy(synth.synth_audio(2, 1, 64160, seed=41)).to(torch.bfloat16).to(dev())
    eng = m._ensure_engine()
    res = {}
    for mode in ("bf16", "fp8"):
        eng.fp8 = mode == "fp8"
        out = m(audio, ctx, tgt, vis)
        out["loss"].backward()
        res[mode] = dict(loss=float(out["loss"].detach()), lf=out["local_features"].float().clone(), targets=out["targets"].float().clone(),
                         grads={k: p.grad.double().clone() for k, p in m.named_parameters() if p.grad is not None})
    eng.fp8 = False
    # bf16 path vs the oracle at 400 tokens on the base model
    names = J.trainable_names(P)
    for k in names:
        P[k].requires_grad_(True)
    ref = J.jepa_forward(P, audio, ctx.to(dev()), tgt.to(dev()), vis.to(dev()), mode="bf16", **oracle_kw(BASE))
    lr_ = float(ref["loss"].detach())
    assert abs(res["bf16"]["loss"] - lr_) < 1e-3 * abs(lr_), (res["bf16"]["loss"], lr_)
    assert rel(res["bf16"]["targets"], ref["targets"].float()) < 1e-2
    # fp8 vs bf16
    b, f = res["bf16"], res["fp8"]
    assert torch.equal(b["lf"], f["lf"])                                   # the conv front-end is not quantised
    dl = abs(f["loss"] - b["loss"]) / abs(b["loss"])
    dt = rel(f["targets"], b["targets"])
    num, den = {}, {}
    for k in b["grads"]:
        g = group_of(k)
        num[g] = num.get(g, 0.0) + float((f["grads"][k] - b["grads"][k]).pow(2).sum())
        den[g] = den.get(g, 0.0) + float(b["grads"][k].pow(2).sum())
    errs = {g: (num[g] / max(den[g], 1e-300)) ** 0.5 for g in num}
    print(f"fp8 vs bf16 path, base model, 400 tokens: loss {f['loss']:.6f} vs {b['loss']:.6f} (rel {dl:.3e}); targets rel L2 {dt:.3e}; "
          f"gradient groups rel L2 {errs}")
    assert dl < 1e-2 and dt < 8e-2
    for g, e in errs.items():
        assert e < 0.15, (g, e)


def test_fp8_inference_after_a_training_step_at_the_same_batch_size(monkeypatch):
    """A training step sizes the student's buffers for its ragged row count (cap_enc ~ 0.2 M rows); get_audio_representation at
    the same N keeps that arena and runs the student densely.  In fp8 mode the quantiser and the GELU q_out epilogue write M x 4D
    bytes into the stack's fp8 scratch, which therefore has to hold M rows whatever cap_enc is (round-3 advisor finding: it held
    cap_enc rows -- an out-of-bounds device write).  NaN-poisoned arena: a read of memory no kernel wrote shows up in the output."""
    monkeypatch.setenv("WJ_ARENA_FILL", "nan")
    cfg = dict(SMALL, d_enc=256, h_enc=4)                    # K = 256: eligible for the MX fp8 GEMM
    m, P = build(cfg)
    eng = m._ensure_engine()
    eng.fp8 = True
    n = 4
    fx = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "masks.npz")))
    ctx, tgt, vis = (torch.from_numpy(fx[k][:n]) for k in ("as_ctx", "as_tgt", "as_vis"))
    audio = torch.from_numpy(synth.synth_audio(n, 1, 32159, seed=5)).to(dev())
    out = m(audio.to(torch.bfloat16), ctx, tgt, vis)
    out["loss"].backward()
    assert eng.ragged_step and eng.cap_enc < eng.M
    q, sc = eng._a8["enc"]
    assert q.shape[0] >= eng.M and eng._a8s["enc"][0].shape[0] >= eng.M
    guard = torch.cuda.memory_allocated()
    rep = m.get_audio_representation(audio, None)
    torch.cuda.synchronize()
    assert eng.N == n and torch.cuda.memory_allocated() >= guard            # same arena, nothing re-allocated smaller
    assert bool(torch.isfinite(rep).all())
    eng.fp8 = False
    rep_bf = m.get_audio_representation(audio, None)
    assert rel(rep, rep_bf) < 8e-2                                            # fp8 forward vs the bf16 forward (tolerance of the fp8 test)


def test_mask_gather_bit_exact_and_shapes(golden_dir):
    m, P = build(SMALL)
    ctx, tgt, vis = masks(golden_dir, 3)
    audio = torch.from_numpy(synth.synth_audio(3, 1, 32159, seed=5)).to(torch.bfloat16).to(dev())
    with torch.no_grad():
        m(audio, ctx, tgt, vis)
    eng = m._engine
    n_ctx = int((~ctx).sum())
    assert eng.plan.n_ctx == n_ctx and eng.ragged_step
    sel = (~ctx).to(dev())
    # ragged step: the boolean-mask gather happens on the encoder INPUT (pure copies in (b, t) row-major order) ...
    assert torch.equal(eng.enc_in_b[:n_ctx], eng.lf_b.view(3, 200, -1)[sel])
    assert torch.equal(eng.enc_in[:n_ctx], eng.lf.view(3, 200, -1)[sel])
    ctx_ragged = eng.ctx_in[:n_ctx].clone()
    # ... dense step: on the encoder output, as the reference does (jepa.py:399)
    eng.ragged = False
    with torch.no_grad():
        m(audio, ctx, tgt, vis)
    assert not eng.ragged_step
    want = eng.enc_out_b.view(3, 200, -1)[sel]
    assert torch.equal(eng.ctx_in[:n_ctx], want)
    assert rel(ctx_ragged.float(), want.float()) < 1e-2              # same rows either way (bf16 summation-order noise only)


@pytest.mark.selfcheck
def test_ragged_equals_dense_step(golden_dir):
    """Visible-token execution changes no result: loss and every parameter gradient agree with the dense key-masked step
    to accumulation-order noise."""
    ctx, tgt, vis = masks(golden_dir, 4)
    audio = torch.from_numpy(synth.synth_audio(4, 1, 32159, seed=9)).to(torch.bfloat16).to(dev())
    res = {}
    for ragged in (True, False):
        m, _ = build(SMALL)
        m._ensure_engine().ragged = ragged
        out = m(audio, ctx, tgt, vis)
        out["loss"].backward()
        res[ragged] = (float(out["loss"]), {k: p.grad.double().clone() for k, p in m.named_parameters() if p.grad is not None})
    (l1, g1), (l0, g0) = res[True], res[False]
    assert abs(l1 - l0) < 2e-4 * abs(l0), (l1, l0)
    num = sum(float((g1[k] - g0[k]).pow(2).sum()) for k in g0)
    den = sum(float(g0[k].pow(2).sum()) for k in g0)
    assert (num / den) ** 0.5 < 1e-2, (num / den) ** 0.5


@pytest.mark.selfcheck
def test_step_is_reproducible_and_sparse_conv_backward_equals_dense(golden_dir):
    """(1) The forward holds no float atomics (conv0 GroupNorm statistics and the teacher's per-clip sums are stored per
    workgroup and folded in a fixed order), so the loss of the same step on freshly built models is BIT-identical, and with it the
    whole dgrad chain; parameter gradients then differ only by the fp32 rounding of their split-K / column-sum atomics.
    (2) Conv backward over the active rows only (gather GEMMs) == the dense conv backward, including on a second step with
    different masks (the gradient buffers must be back to all-zero between steps).
    (3) Each of them against the oracle's gradient, not only against each other."""
    fx = dict(np.load(os.path.join(golden_dir, "masks.npz")))
    sets = [tuple(torch.from_numpy(fx[k][i:i + 3]) for k in ("as_ctx", "as_tgt", "as_vis")) for i in (0, 3)]
    audios = [torch.from_numpy(synth.synth_audio(3, 1, 32159, seed=11 + i)).to(torch.bfloat16).to(dev()) for i in range(2)]
    grads, losses = {}, {}
    for tag, sparse in (("sparse", True), ("dense", False), ("sparse2", True), ("dense2", False)):
        m, P = build(SMALL)
        eng = m._ensure_engine()
        eng.sparse_conv = sparse
        for i in range(2):
            m.zero_grad(set_to_none=True)
            out = m(audios[i], *sets[i])
            out["loss"].backward()
        assert eng.ragged_step
        losses[tag] = float(out["loss"].detach())
        grads[tag] = {k: p.grad.double().clone() for k, p in m.named_parameters() if k.startswith(("extract_audio", "feature_norms"))}
    assert len(set(losses.values())) == 1, losses                     # (1) bit-identical forward, run to run

    def err(a, b):
        return {k: float((grads[a][k] - grads[b][k]).norm() / (grads[b][k].norm() + 1e-30)) for k in grads[b]}

    noise_d, noise_s, got = err("dense2", "dense"), err("sparse2", "sparse"), err("sparse", "dense")
    print("conv backward rel err per tensor, dense vs dense:", noise_d, "sparse vs sparse:", noise_s, "sparse vs dense:", got)
    for k in got:
        assert noise_d[k] < 1e-5 and noise_s[k] < 1e-5, (k, noise_d[k], noise_s[k])   # fp32 atomic rounding only
        assert got[k] < 1e-5, (k, got[k])                             # (2) same gradient, different summation order
    # (3) the oracle's gradient of the second step (bf16 flow) for the same tensors
    P = {k: v.detach().clone() for k, v in P.items()}
    names = J.trainable_names(P)
    for k in names:
        P[k].requires_grad_(True)
    ref = J.jepa_forward(P, audios[1], *(t.to(dev()) for t in sets[1]), mode="bf16", **oracle_kw(SMALL))
    ref["loss"].backward()
    for tag in ("sparse", "dense"):
        for k in grads[tag]:
            e = float((grads[tag][k] - P[k].grad.double()).norm() / (P[k].grad.double().norm() + 1e-30))
            assert e < 3e-2, (tag, k, e)


@pytest.mark.selfcheck
def test_target_outside_visible_set_falls_back_to_dense(golden_dir):
    """If a target position is key-masked the predictor row must still be computed as a query: such a batch takes the
    dense path (the reference maskers never produce it)."""
    m, _ = build(SMALL)
    ctx, tgt, vis = masks(golden_dir, 2)
    vis = vis.clone()
    b, g, t = [int(v[0]) for v in torch.nonzero(tgt, as_tuple=True)]
    vis[b, g, t] = True
    audio = torch.from_numpy(synth.synth_audio(2, 1, 32159, seed=5)).to(torch.bfloat16).to(dev())
    with torch.no_grad():
        m(audio, ctx, tgt, vis)
    assert m._engine.ragged and not m._engine.ragged_step


def test_training_trajectory_vs_oracle(golden_dir):
    """10 optimisation steps (forward, EMA, backward, clip 5, AdamW, cosine/warm-up) against the oracle's train_step."""
    m, P = build(SMALL, warmup_steps=3)
    P = {k: v.detach().clone() for k, v in P.items()}
    m.trainer.max_steps = 20
    m.hparams["ema_decay"], m.hparams["ema_end_decay"], m.ema_end_step = 0.9, 0.99, 10
    oc = m.configure_optimizers()
    opt, sch = oc["optimizer"], oc["lr_scheduler"]["scheduler"]
    opt.max_grad_norm = 5.0
    ctx, tgt, vis = masks(golden_dir, 6)
    state = {}
    worst = 0.0
    for i in range(10):
        sl = slice(2 * (i % 3), 2 * (i % 3) + 2)
        audio = torch.from_numpy(synth.synth_audio(2, 1, 32159, seed=100 + i % 3)).to(torch.bfloat16).to(dev())
        batch = (audio, ctx[sl], tgt[sl], vis[sl])
        m.global_step = i
        out = m.training_step(batch, i)
        out["loss"].backward()
        opt.step()
        sch.step()
        r = J.train_step(P, state, i, (audio, ctx[sl].to(dev()), tgt[sl].to(dev()), vis[sl].to(dev())), mode="bf16", warmup=3,
                         total_steps=20, ema=(0.9, 0.99, 10), **oracle_kw(SMALL))
        lo = float(out["loss"])
        worst = max(worst, abs(lo - r["loss"]) / abs(r["loss"]))
        gn = float(opt.grad_norm())
        assert abs(gn - r["grad_norm"]) < 3e-2 * r["grad_norm"], (i, gn, r["grad_norm"])
    print("worst relative loss deviation over 10 steps:", worst)
    assert worst < 2e-3
    sd = m.state_dict()
    for k in ("encoder.layers.1.linear1.weight", "teacher_encoder.layers.1.linear1.weight", "extract_audio.cnn.2.0.weight"):
        assert rel(sd[k], P[k]) < 2e-3, k


def test_north_star_100_step_trajectory_vs_reference(golden_dir):
    """BASELINE.json north star: "JEPA loss within 1e-3 of reference over 100 steps".  The BASE model (196 M parameters), N = 4 clips
    per step, 100 optimisation steps (training_step incl. EMA -> backward -> clip 5 -> AdamW -> per-step cosine schedule, 10-step
    warm-up to lr 4e-4, EMA 0.99 -> 0.999 over 50 steps) against the trajectory the REFERENCE itself produced for the same weights,
    audio and masks (tests/golden/base_traj.npz, written by make_golden.py importing /root/reference): its bf16-autocast run (the
    precision train.py uses) and its fp32 run.  Tolerances: every one of the 100 losses within 1e-3 (absolute, the north-star
    criterion) AND within 2e-2 relative of the bf16 reference while the loss falls by 2.5 orders of magnitude; gradient norms within
    3e-2 relative; final parameter slices within 2e-3 relative L2."""
    fx = dict(np.load(os.path.join(golden_dir, "base_traj.npz")))
    steps, warm, total = int(fx["steps"]), int(fx["warmup"]), int(fx["total"])
    e0, e1, e_end = (float(v) for v in fx["ema"])
    m, _ = build(BASE, warmup_steps=warm)
    m.trainer.max_steps = total
    m.hparams["ema_decay"], m.hparams["ema_end_decay"], m.ema_end_step = e0, e1, int(e_end)
    oc = m.configure_optimizers()
    opt, sch = oc["optimizer"], oc["lr_scheduler"]["scheduler"]
    opt.max_grad_norm = 5.0
    mk = dict(np.load(os.path.join(golden_dir, "masks.npz")))
    batches = []
    for j, seed in enumerate(fx["audio_seeds"].tolist()):
        sl = slice(4 * (j % 2), 4 * (j % 2) + 4)
        audio = torch.from_numpy(synth.synth_audio(4, 1, 32159, seed=int(seed))).to(torch.bfloat16).to(dev())
        batches.append((audio,) + tuple(torch.from_numpy(mk[k][sl]) for k in ("as_ctx", "as_tgt", "as_vis")))
    losses, gnorms = [], []
    for i in range(steps):
        m.global_step = i
        out = m.training_step(batches[i % len(batches)], i)
        out["loss"].backward()
        opt.step()
        sch.step()
        losses.append(out["loss"].detach())
        gnorms.append(opt.grad_norm().clone())
    losses = torch.stack(losses).double().cpu().numpy()
    gnorms = torch.stack(gnorms).double().cpu().numpy().reshape(-1)
    for tag in ("bf16", "fp32"):
        ref = fx[f"{tag}::loss"]
        d = np.abs(losses - ref)
        print(f"vs reference {tag}: max |dloss| {d.max():.3e} at step {int(d.argmax())}, max rel {np.max(d / ref):.3e} at step "
              f"{int(np.argmax(d / ref))}; loss {losses[0]:.5f} -> {losses[-1]:.6f} (reference {ref[0]:.5f} -> {ref[-1]:.6f}); "
              f"grad norm max rel dev {np.max(np.abs(gnorms - fx[f'{tag}::gnorm']) / fx[f'{tag}::gnorm']):.3e}")
    ref = fx["bf16::loss"]
    assert np.max(np.abs(losses - ref)) < 1e-3                                      # north star, all 100 steps
    assert np.max(np.abs(losses - fx["fp32::loss"])) < 1e-3
    # yardstick: the reference's OWN two precisions differ by 3.1e-4 absolute / 3.2e-3 relative in loss over these 100 steps, and
    # by up to 20 % in gradient norm once it has fallen below 0.01 (steps > 80): relative bounds are set a small multiple above
    rel_ref = np.max(np.abs(fx["bf16::loss"] - fx["fp32::loss"]) / fx["fp32::loss"])
    assert np.max(np.abs(losses - ref) / ref) < max(1e-2, 3 * rel_ref)
    big = fx["bf16::gnorm"] > 0.05
    assert np.max((np.abs(gnorms - fx["bf16::gnorm"]) / fx["bf16::gnorm"])[big]) < 3e-2
    sd = m.state_dict()
    for k in ("encoder.layers.11.linear1.weight", "teacher_encoder.layers.11.linear1.weight", "extract_audio.cnn.2.0.weight",
              "decoder.layers.0.self_attn.in_proj_weight"):
        got = sd[k].detach().float().cpu().numpy().reshape(-1)[::997]
        want = fx[f"bf16::final_slice::{k}"]
        assert np.linalg.norm(got - want) / np.linalg.norm(want) < 2e-3, k


def test_inference_representation(golden_dir):
    m, P = build(SMALL)
    audio = torch.from_numpy(synth.synth_audio(2, 1, 32159, seed=9)).to(dev())
    pad = torch.zeros(2, 200, dtype=torch.bool)
    pad[:, 150:] = True
    rep = m.get_audio_representation(audio, pad.to(dev()))
    ref = J.audio_representation(P, audio.to(torch.bfloat16), pad.to(dev()), spec=SMALL_SPEC, enc_heads=2, mode="bf16")
    assert rep.shape == (2, 200, 128) and rel(rep[:, :150], ref[:, :150]) < 1e-2
    # state loaded AFTER the engine exists -- weights and the frozen position table -- reaches the kernels
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    sd["pos_encoding_encoder"] = sd["pos_encoding_encoder"] * 0.5
    sd["encoder.layers.0.linear1.weight"] = sd["encoder.layers.0.linear1.weight"] * 1.25
    m.load_state_dict(sd)
    P2 = dict(P)
    P2["pos_encoding_encoder"] = P["pos_encoding_encoder"] * 0.5
    P2["encoder.layers.0.linear1.weight"] = P["encoder.layers.0.linear1.weight"] * 1.25
    rep2 = m.get_audio_representation(audio, pad.to(dev()))
    ref2 = J.audio_representation(P2, audio.to(torch.bfloat16), pad.to(dev()), spec=SMALL_SPEC, enc_heads=2, mode="bf16")
    assert rel(rep2[:, :150], ref2[:, :150]) < 1e-2 and rel(rep2[:, :150], ref[:, :150]) > 5e-2


def test_hear_runtime_timestamp_embeddings_vs_oracle():
    """HEAR-2021 wrapper (reference hear_api/runtime.py): loudness normalisation, 2.01 s windows, padded-token key mask,
    cut-off and timestamps, on the base model, against the oracle's restatement (CPU, fp32)."""
    from hear_api.runtime import RuntimeJEPA
    from oracle import hear_oracle as HO
    from wavjepa_amd.extractors import ConvFeatureExtractor
    ext = ConvFeatureExtractor(conv_layers_spec=list(J.WAVJEPA_CONV_SPEC), in_channels=1)
    rt = RuntimeJEPA(in_channels=1, weights=None, is_spectrogram=False, process_seconds=2.01, extractor=ext, model_size="base", sr=16000)
    shapes = {k: tuple(v.shape) for k, v in rt.model.state_dict().items()}
    sd = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, seed=23).items()}
    sd["pos_encoding_encoder"] = J.sincos_positions(768, 200)
    sd["pos_encoding_decoder"] = J.sincos_positions(384, 200)
    rt.model.load_state_dict(sd)
    assert rt.scene_embedding_size == 768 and rt.timestamp_embedding_size == 768 and rt.sample_rate == 16000
    for n_samples in (16000, 50000):
        wave = torch.from_numpy(synth.synth_audio(2, 1, n_samples, seed=29 + n_samples)).float()[:, 0]      # [B, n]
        emb, ts = rt.get_timestamp_embeddings(wave)
        feats = rt.to_feature(wave).cpu()
        ref, ref_ts = HO.timestamp_embeddings(sd, feats, mode="fp32")
        assert emb.shape == ref.shape and ts.shape == ref_ts.shape
        assert torch.allclose(ts.cpu(), ref_ts, atol=1e-3)
        assert rel(emb, ref) < 2e-2, (n_samples, rel(emb, ref))
        scene = rt.get_scene_embeddings(wave)
        assert scene.shape == (2, 768) and rel(scene, ref.mean(1)) < 2e-2
        if n_samples == 50000:
            # the same clips and weights through the REFERENCE's RuntimeJEPA (CPU fp32; tests/golden/hear_runtime.npz, make_golden.py)
            fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hear_runtime.npz"))
            assert int(fx["n_samples"]) == n_samples and tuple(fx["emb_shape"]) == tuple(emb.shape)
            assert np.allclose(ts.cpu().numpy(), fx["ts"], atol=1e-3)
            assert rel(emb[:, ::7, ::5], torch.from_numpy(fx["emb_sub"])) < 2e-2, rel(emb[:, ::7, ::5], torch.from_numpy(fx["emb_sub"]))
            assert rel(scene, torch.from_numpy(fx["scene"])) < 2e-2
            assert rel(ref[:, ::7, ::5], torch.from_numpy(fx["emb_sub"])) < 1e-4        # the oracle restates the reference


def test_hear_nat_runtime_timestamp_embeddings_vs_oracle():
    """Multi-channel HEAR wrapper (reference hear_api/runtime_natjepa.py:90-93,139-147): a 2-channel clip through the channel
    extractor's two token streams per 2.01 s window; the padded-step key mask repeated per stream; embedding = mean over the streams;
    cut-off and timestamps as in the mono wrapper.  Base model, against the oracle's restatement (CPU, fp32); a mono clip is
    duplicated to both channels by the feature helper."""
    from hear_api.runtime_natjepa import RuntimeNatJEPA
    from oracle import hear_oracle as HO
    from wavjepa_amd.extractors import ConvChannelFeatureExtractor
    ext = ConvChannelFeatureExtractor(conv_layers_spec=list(J.WAVJEPA_CONV_SPEC), in_channels=2, share_weights_over_channels=False)
    rt = RuntimeNatJEPA(in_channels=2, weights=None, is_spectrogram=False, process_seconds=2.01, extractor=ext, model_size="base", sr=16000)
    assert rt.output_steps == 200 and rt.model.total_patches == 400 and rt.scene_embedding_size == 768
    own = rt.model.state_dict()
    shapes = {k: tuple(v.shape) for k, v in own.items()}
    sd = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, seed=41).items()}
    for k in ("pos_encoding_encoder", "pos_encoding_decoder"):
        sd[k] = own[k].detach().cpu().clone()
    rt.model.load_state_dict(sd)
    wave = torch.from_numpy(synth.synth_audio(2, 2, 40000, seed=77)).float()                      # [B, 2, n]: 2 windows, 1.2 of them audio
    emb, ts = rt.get_timestamp_embeddings(wave)
    feats = rt.to_feature(wave).cpu()
    assert feats.shape == (2, 2, 40000)
    ref, ref_ts = HO.timestamp_embeddings(sd, feats, mode="fp32", channel_streams=2)
    assert emb.shape == ref.shape == (2, ref.shape[1], 768) and 200 < ref.shape[1] < 400 and ts.shape == ref_ts.shape
    assert torch.allclose(ts.cpu(), ref_ts, atol=1e-3)
    assert rel(emb, ref) < 2e-2, rel(emb, ref)
    scene = rt.get_scene_embeddings(wave)
    assert scene.shape == (2, 768) and rel(scene, ref.mean(1)) < 2e-2
    mono = rt.to_feature(wave[:, 0])                                                                # [B, n] -> both channels
    assert mono.shape == (2, 2, 40000) and torch.equal(mono[:, 0], mono[:, 1])
    with pytest.raises(ValueError):
        RuntimeNatJEPA(in_channels=4, weights=None, is_spectrogram=False, process_seconds=2.01, extractor=ext, model_size="base", sr=16000)


def test_trainer_checkpoint_resume_continues_the_same_trajectory(tmp_path):
    """Checkpoint / resume (reference train.py:244 `trainer.fit(..., ckpt_path=...)`; Lightning checkpoint keys): six steps straight
    through against three steps, a checkpoint, and a NEW process-like start (differently initialised model, fresh optimiser and
    scheduler) that resumes from it for the last three.  Parameters, EMA teacher, AdamW moments, step counter and learning rate
    must land on the same values.  Two bounds, because the comparison has two regimes:
      * ONE step past the restart (the step=4 checkpoints of both runs) the two runs started from bit-identical state, and the fp32
        atomics of the weight gradients' split-K sums are the only order-dependent arithmetic: 1e-5 on everything.  A state that was
        not carried over (moments, teacher, step counter, learning rate, bf16 shadows) shows here at O(1e-2..1).
      * at step 6 that 1e-7 noise has been through two more updates: once in ~10 runs it carries one fp32 weight across a bf16
        rounding boundary, that weight's bf16 shadow moves by 2^-8 and the next backward's gradients by ~1e-4 relative (seen once in a
        full-suite run of round 6: adam_m 1.0e-4, parameters still < 1e-5; the masks come from OS entropy, so the draw differs per
        run).  Parameters and teacher stay at 1e-5 (lr x the moments' difference); the moments get 5e-3."""
    from wavjepa_amd.data import SyntheticAudioSource
    from wavjepa_amd.masking import TimeInverseBlockMasker
    from wavjepa_amd.trainer import Trainer

    def source():
        return SyntheticAudioSource(TimeInverseBlockMasker(4, 0.65, 10, 0.25, 10, 0.1), batch_size=2, samples_per_audio=2, n_tokens=200,
                                    seconds=3.0, seed=11, n_mask_sets=4, device=dev())

    mask_sets = source().mask_sets             # the masker draws from OS entropy (as upstream): one set of masks for both runs

    def loader(skip):
        src = source()
        src.mask_sets = mask_sets
        i = 0
        while True:
            b = src.next_batch()
            torch.manual_seed(1000 + i)        # the crop offsets of on_after_batch_transfer come from the global generator
            if i >= skip:
                yield b
            i += 1

    def run(seed, root, ckpt=None, skip=0):
        m, _ = build(SMALL, seed=seed, warmup_steps=2)
        tr = Trainer(max_steps=6, default_root_dir=str(root), checkpoint_every_n_steps=1, log_every_n_steps=0)
        return m, tr.fit(m, train_dataloaders=loader(skip), ckpt_path=ckpt)

    ma, ra = run(7, tmp_path / "a")
    ck = tmp_path / "a" / "step=3.ckpt"
    assert ck.exists() and (tmp_path / "a" / "last.ckpt").exists()
    saved = torch.load(ck, map_location="cpu", weights_only=False)
    assert saved["global_step"] == 3 and {"state_dict", "hyper_parameters", "optimizer", "lr_scheduler"} <= set(saved)
    mb, rb = run(8, tmp_path / "b", ckpt=str(ck), skip=3)
    assert ma.global_step == mb.global_step == 6 and ra.optimizer._t == rb.optimizer._t == 6
    assert ra.scheduler.get_last_lr() == rb.scheduler.get_last_lr()
    a4 = torch.load(tmp_path / "a" / "step=4.ckpt", map_location="cpu", weights_only=False)
    b4 = torch.load(tmp_path / "b" / "step=4.ckpt", map_location="cpu", weights_only=False)
    assert a4["global_step"] == b4["global_step"] == 4 and a4["optimizer"]["step"] == b4["optimizer"]["step"] == 4
    assert a4["optimizer"]["lr"] == b4["optimizer"]["lr"] and a4["lr_scheduler"] == b4["lr_scheduler"]
    assert set(a4["state_dict"]) == set(b4["state_dict"])
    for k, va in a4["state_dict"].items():
        vb = b4["state_dict"][k]
        if va.is_floating_point() and va.numel() > 1 and float(va.float().norm()) > 0:
            assert rel(va.float(), vb.float()) < 1e-5, ("step 4", k, rel(va.float(), vb.float()))
        else:
            assert torch.equal(va, vb), ("step 4", k)
    for k in ("m", "v"):
        assert rel(a4["optimizer"][k], b4["optimizer"][k]) < 1e-5, ("step 4", k, rel(a4["optimizer"][k], b4["optimizer"][k]))
    for name, a, b, tol in (("student", ma._flat.p32, mb._flat.p32, 1e-5), ("teacher", ma._flat.t32, mb._flat.t32, 1e-5),
                            ("adam_m", ma._flat.adam_m, mb._flat.adam_m, 5e-3), ("adam_v", ma._flat.adam_v, mb._flat.adam_v, 5e-3)):
        assert rel(a, b) < tol, (name, rel(a, b))


def test_state_dict_roundtrip_and_reference_checkpoint_layout():
    m, P = build(SMALL)
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    shapes = synth.jepa_shapes(conv_spec=SMALL_SPEC, in_channels=1, d_enc=128, enc_layers=2, d_dec=64, dec_layers=2, n_tokens=200)
    assert {k: tuple(v.shape) for k, v in sd.items()} == shapes
    m2, _ = build(SMALL, seed=11)
    m2.load_state_dict(sd)
    for k, v in m2.state_dict().items():
        assert torch.equal(v.cpu(), sd[k]), k


def test_overlapped_adamw_is_the_same_update(golden_dir):
    """The training loop's optimiser split (FusedAdamW.overlap_next_forward: the transformer stacks' update on the side stream beside the
    next step's conv front-end, one workgroup per CU; the forward / state_dict wait for its event) against the single launch on the compute
    stream: the same trajectory -- parameters, teacher, Adam moments, loss -- up to the run-to-run noise of the fp32 split-K atomics in the
    weight gradients.  A missing wait shows as a stale or half-written parameter buffer, orders of magnitude above that noise."""
    from wavjepa_amd.trainer import StepRunner
    ctx, tgt, vis = masks(golden_dir, 4)
    runs = []
    for overlap in (False, True):
        m, _ = build(SMALL, warmup_steps=2)
        m.trainer.max_steps = 20
        run = StepRunner(m)
        assert run.optimizer.overlap_next_forward               # the loop switches it on for JEPA
        run.optimizer.overlap_next_forward = overlap
        for i in range(5):
            audio = torch.from_numpy(synth.synth_audio(4, 1, 32159, seed=500 + i)).to(torch.bfloat16).to(dev())
            out = m.training_step((audio, ctx, tgt, vis), i)
            out["loss"].backward()
            run.reducer.wait()
            run.optimizer.step()
            run.scheduler.step()
            m.global_step = i + 1
        assert (m._engine._opt_ev is not None) == overlap        # an update is still in flight behind the last step
        sd = {k: v.detach().float().clone() for k, v in m.state_dict().items()}    # (state_dict waits for it)
        assert m._engine._opt_ev is None
        osd = run.optimizer.state_dict()
        torch.cuda.synchronize()
        runs.append((sd, osd["m"].float(), osd["v"].float(), float(out["loss"].detach())))
    (sd0, m0, v0, l0), (sd1, m1, v1, l1) = runs
    assert abs(l0 - l1) < 1e-5 * max(1.0, abs(l0))
    for k in sd0:
        assert rel(sd1[k], sd0[k]) < 2e-5, (k, rel(sd1[k], sd0[k]))
    assert rel(m1, m0) < 1e-4 and rel(v1, v0) < 1e-4
