"""Pins the CPU oracle (oracle/) against fixtures produced by the reference itself
(tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import jepa_oracle as J
from oracle import masking_oracle as M
import synth

TINY_SPEC = [(32, 10, 5)] + [(32, 3, 2)] * 4 + [(32, 2, 2)]
TINY = dict(spec=TINY_SPEC, enc_heads=4, dec_heads=4, top_k=2)


def load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name), allow_pickle=False))


def params_from(fx, prefix="sd::"):
    return {k[len(prefix):]: torch.from_numpy(v.copy()) for k, v in fx.items() if k.startswith(prefix)}


def rel(a, b):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    return float((a - b).norm() / (b.norm() + 1e-30))


class Pinned:
    def __init__(self, base):
        self.base, self.k = base, 0

    def __call__(self):
        g = np.random.default_rng(self.base + self.k)
        self.k += 1
        return g


def test_masks_bit_exact(golden_dir):
    fx = load(golden_dir, "masks.npz")
    c, t, v = M.time_inverse_block_masks(8, 200, 1, new_rng=Pinned(int(fx["as_base"])))
    assert np.array_equal(c, fx["as_ctx"]) and np.array_equal(t, fx["as_tgt"]) and np.array_equal(v, fx["as_vis"])
    c, t, v = M.speech_masks(8, 200, 1, new_rng=Pinned(int(fx["ls_base"])))
    assert np.array_equal(c, fx["ls_ctx"]) and np.array_equal(t, fx["ls_tgt"]) and np.array_equal(v, fx["ls_vis"])
    c, t, v = M.time_inverse_block_masks(4, 400, 1, new_rng=Pinned(int(fx["as400_base"])))
    assert np.array_equal(c, fx["as400_ctx"]) and np.array_equal(t, fx["as400_tgt"]) and np.array_equal(v, fx["as400_vis"])


def test_mask_invariants(golden_dir):
    fx = load(golden_dir, "masks.npz")
    for tag, cutoff in (("as", 0.1), ("ls", 0.5)):
        c, t, v = fx[f"{tag}_ctx"], fx[f"{tag}_tgt"], fx[f"{tag}_vis"]
        ctx = ~c
        assert not (ctx[:, None, :] & t).any()                      # context and targets are disjoint
        assert ((ctx.sum(-1) / c.shape[-1]) >= cutoff).all()
        assert np.array_equal(v, np.logical_xor(c[:, None, :], t))
        assert np.array_equal(~v, ctx[:, None, :] | t)             # visible = context U group targets


def test_tiny_forward_fp32(golden_dir):
    fx = load(golden_dir, "tiny_model.npz")
    P = params_from(fx)
    audio, ctx, tgt, vis = (torch.from_numpy(fx[k]) for k in ("audio", "ctx", "tgt", "vis"))
    out = J.jepa_forward(P, audio, ctx, tgt, vis, mode="fp32", **TINY)
    for k in ("local_features", "contextual_features", "preds", "targets"):
        assert out[k].shape == fx[f"out::{k}"].shape
        assert rel(out[k], fx[f"out::{k}"]) < 2e-5, k
    assert abs(float(out["loss"]) - float(fx["out::loss"])) < 1e-5 * abs(float(fx["out::loss"]))
    # gather is a pure copy: bit-exact against the numpy statement
    enc = J.encoder_stack(P, "encoder", out["local_features"], 4, ctx, "fp32")
    g = M.gather_rows(enc.numpy(), ctx.numpy())
    assert np.array_equal(g, enc[~ctx].numpy())


def test_tiny_backward_fp32(golden_dir):
    fx = load(golden_dir, "tiny_model.npz")
    P = params_from(fx)
    names = J.trainable_names(P)
    for n in names:
        P[n].requires_grad_(True)
    audio, ctx, tgt, vis = (torch.from_numpy(fx[k]) for k in ("audio", "ctx", "tgt", "vis"))
    out = J.jepa_forward(P, audio, ctx, tgt, vis, mode="fp32", **TINY)
    out["loss"].backward()
    want = dict(zip(fx["grad_names"].tolist(), fx["grad_norms"].tolist()))
    assert set(want) == {n for n in names if P[n].grad is not None}
    for n, g in want.items():
        got = float(P[n].grad.norm())
        assert abs(got - g) <= 2e-4 * g + 1e-9, (n, got, g)
    for n in ("extract_audio.cnn.0.0.weight", "encoder.layers.0.self_attn.in_proj_weight", "mask_token"):
        assert rel(P[n].grad, fx[f"grad::{n}"]) < 2e-4, n


def test_tiny_bf16_mode_close_to_reference_cpu_autocast(golden_dir):
    fx = load(golden_dir, "tiny_model.npz")
    P = params_from(fx)
    audio, ctx, tgt, vis = (torch.from_numpy(fx[k]) for k in ("audio", "ctx", "tgt", "vis"))
    out = J.jepa_forward(P, audio.to(torch.bfloat16), ctx, tgt, vis, mode="bf16", **TINY)
    # CPU autocast keeps GroupNorm/LayerNorm in bf16 while the oracle follows the CUDA/ROCm policy (fp32):
    # a loose check only; the bf16 mode is also within bf16 noise of the fp32 fixture.
    assert abs(float(out["loss"]) - float(fx["out_bf16cpu::loss"])) < 3e-2 * float(fx["out_bf16cpu::loss"])
    assert abs(float(out["loss"]) - float(fx["out::loss"])) < 2e-2 * float(fx["out::loss"])
    assert rel(out["local_features"], fx["out::local_features"]) < 2e-2
    assert out["preds"].dtype == torch.bfloat16 and out["targets"].dtype == torch.float32
    assert out["local_features"].dtype == torch.float32 and out["contextual_features"].dtype == torch.bfloat16


def test_audio_representation(golden_dir):
    fx = load(golden_dir, "tiny_model.npz")
    P = params_from(fx)
    rep = J.audio_representation(P, torch.from_numpy(fx["audio"]), torch.from_numpy(fx["pad_mask"]),
                                 spec=TINY_SPEC, enc_heads=4)
    assert rel(rep, fx["out::audio_representation"]) < 2e-5


@pytest.mark.parametrize("tag", ["short", "stock"])
def test_training_trajectory(golden_dir, tag):
    fx = load(golden_dir, "tiny_traj.npz")
    masks = load(golden_dir, "masks.npz")
    P = params_from(fx, "sd0::")
    batches = []
    for j in range(3):
        sl = slice(2 * j, 2 * j + 2)
        batches.append((torch.from_numpy(fx["audio"][j]), torch.from_numpy(masks["as_ctx"][sl]),
                        torch.from_numpy(masks["as_tgt"][sl]), torch.from_numpy(masks["as_vis"][sl])))
    if tag == "short":
        cfg = dict(warmup=3, total_steps=20, ema=(0.9, 0.99, 10))
        steps = 12
    else:
        cfg = dict(warmup=100000, total_steps=375000, ema=(0.999, 0.99999, 100000))
        steps = 4
    state = {}
    for i in range(steps):
        r = J.train_step(P, state, i, batches[i % 3], mode="fp32", **cfg, **TINY)
        assert abs(r["loss"] - fx[f"{tag}::loss"][i]) < 3e-4 * abs(fx[f"{tag}::loss"][i]), (i, r["loss"])
        assert abs(r["grad_norm"] - fx[f"{tag}::gnorm"][i]) < 3e-3 * fx[f"{tag}::gnorm"][i], i
        assert abs(r["ema"] - fx[f"{tag}::ema"][i]) < 1e-12
        assert abs(4e-4 * J.lr_lambda(i, cfg["warmup"], cfg["total_steps"]) - fx[f"{tag}::lr"][i]) < 1e-15
    want_abs = dict(zip(fx[f"{tag}::names"].tolist(), fx[f"{tag}::abs"].tolist()))
    for n, a in want_abs.items():
        got = float(P[n].double().abs().sum())
        assert abs(got - a) <= 1e-4 * a + 1e-9, (n, got, a)
    if tag == "short":
        for n in ("encoder.layers.1.linear1.weight", "teacher_encoder.layers.1.linear1.weight"):
            assert rel(P[n], fx[f"short::final::{n}"]) < 1e-4, n


def test_crop_normalize(golden_dir):
    fx = load(golden_dir, "crops.npz")
    out = J.crop_normalize(torch.from_numpy(fx["src"]), torch.from_numpy(fx["starts"]), int(fx["target_length"]),
                           torch.from_numpy(fx["perm"]))
    want = torch.from_numpy(fx["out_bits"]).view(torch.bfloat16)
    assert tuple(out.shape) == tuple(fx["out_shape"])
    diff = (out.float() - want.float()).abs()
    # same fp32 formula; allow a final-ulp bf16 difference from summation order in mean/std
    assert float(diff.max()) <= 2 ** -6 and float((diff > 0).float().mean()) < 1e-3


def test_misc_tables(golden_dir):
    fx = load(golden_dir, "misc.npz")
    for d in (768, 384, 64):
        tab = J.sincos_positions(d, 200)[0].numpy()
        assert np.array_equal(tab[::13, ::17], fx[f"pos{d}_slice"])
        assert np.array_equal(tab[199], fx[f"pos{d}_row199"])
        assert abs(float(tab.astype(np.float64).sum()) - float(fx[f"pos{d}_sum"])) < 1e-9
    for s, d in zip(fx["ema_steps"], fx["ema_decay"]):
        assert abs(J.ema_decay(int(s)) - d) < 1e-15
    for s, l in zip(fx["lr_steps"], fx["lr_lambda"]):
        assert abs(J.lr_lambda(int(s), 100000, 375000) - l) < 1e-15
    for L, n in zip(fx["patch_lens"], fx["patch_counts"]):
        assert J.conv_token_count(int(L), TINY_SPEC) == int(n)


def test_base_forward_backward(golden_dir):
    """Base dims (d=768 x 12, predictor 384 x 12, conv 512) with hash-synthesised weights, N=2."""
    fx = load(golden_dir, "base_forward.npz")
    masks = load(golden_dir, "masks.npz")
    shapes = synth.jepa_shapes(conv_spec=J.WAVJEPA_CONV_SPEC, in_channels=1, d_enc=768, enc_layers=12, d_dec=384,
                               dec_layers=12, n_tokens=200)
    sd = synth.synth_state_dict(shapes, seed=int(fx["weight_seed"]))
    for k in list(sd):
        if k.startswith("teacher_encoder.") and k.endswith("weight") and sd[k].ndim == 2:
            sd[k] = (sd[k] * np.float32(float(fx["teacher_scale"]))).astype(np.float32)
    P = {k: torch.from_numpy(v) for k, v in sd.items()}
    P["pos_encoding_encoder"] = J.sincos_positions(768, 200)
    P["pos_encoding_decoder"] = J.sincos_positions(384, 200)
    assert sum(v.numel() for v in P.values()) == int(fx["n_params_total"])
    names = J.trainable_names(P)
    assert sum(P[n].numel() for n in names) == int(fx["n_params_trainable"])
    for n in names:
        P[n].requires_grad_(True)
    N = int(fx["n"])
    audio = torch.from_numpy(synth.synth_audio(N, 1, 32159, seed=int(fx["audio_seed"])))
    ctx, tgt, vis = (torch.from_numpy(masks[k][:N]) for k in ("as_ctx", "as_tgt", "as_vis"))
    out = J.jepa_forward(P, audio, ctx, tgt, vis, mode="fp32")
    assert abs(float(out["loss"]) - float(fx["loss"])) < 2e-5 * float(fx["loss"])
    assert rel(out["local_features"][:, ::25, ::64], fx["local_features_slice"]) < 1e-4
    assert rel(out["contextual_features"][::7, ::16], fx["contextual_features_slice"]) < 1e-4
    assert rel(out["preds"][:, ::25, ::64], fx["preds_slice"]) < 1e-4
    assert rel(out["targets"][:, ::25, ::64], fx["targets_slice"]) < 1e-4
    out["loss"].backward()
    groups = {"conv": "extract_audio.", "feature_norms": "feature_norms.", "mapper": "post_extraction_mapper.",
              "encoder": "encoder.", "enc2dec": "encoder_to_decoder_mapper.", "decoder": "decoder.",
              "dec2enc": "decoder_to_encoder_mapper.", "mask_token": "mask_token"}
    want = dict(zip(fx["grad_group_names"].tolist(), fx["grad_group_norms"].tolist()))
    for g, pre in groups.items():
        tot = sum(float(P[n].grad.double().pow(2).sum()) for n in names if n.startswith(pre)) ** 0.5
        assert abs(tot - want[g]) < 1e-3 * want[g], (g, tot, want[g])
    assert rel(P["extract_audio.cnn.0.0.weight"].grad, fx["grad::extract_audio.cnn.0.0.weight"]) < 1e-3
    assert rel(P["encoder.layers.0.linear1.weight"].grad[::128, ::64], fx["grad_slice::encoder.layers.0.linear1.weight"]) < 1e-3


def test_large_forward_backward(golden_dir):
    """The reference's size="large" model (jepa.py:114-118: student d = 1024, 16 heads, 24 layers; tests/golden/large_forward.npz) with
    hash-synthesised weights, N = 2: the oracle's fp32 flow against the reference's loss, output slices and per-group gradient norms."""
    fx = load(golden_dir, "large_forward.npz")
    masks = load(golden_dir, "masks.npz")
    assert (int(fx["d_enc"]), int(fx["enc_heads"]), int(fx["enc_layers"])) == (1024, 16, 24)
    shapes = synth.jepa_shapes(conv_spec=J.WAVJEPA_CONV_SPEC, in_channels=1, d_enc=1024, enc_layers=24, d_dec=384, dec_layers=12, n_tokens=200)
    assert len(shapes) == int(fx["n_tensors"])
    sd = synth.synth_state_dict(shapes, seed=int(fx["weight_seed"]))
    for k in list(sd):
        if k.startswith("teacher_encoder.") and k.endswith("weight") and sd[k].ndim == 2:
            sd[k] = (sd[k] * np.float32(float(fx["teacher_scale"]))).astype(np.float32)
    P = {k: torch.from_numpy(v) for k, v in sd.items()}
    P["pos_encoding_encoder"] = J.sincos_positions(1024, 200)
    P["pos_encoding_decoder"] = J.sincos_positions(384, 200)
    assert sum(v.numel() for v in P.values()) == int(fx["n_params_total"])
    names = J.trainable_names(P)
    assert sum(P[n].numel() for n in names) == int(fx["n_params_trainable"])
    for n in names:
        P[n].requires_grad_(True)
    N = int(fx["n"])
    audio = torch.from_numpy(synth.synth_audio(N, 1, 32159, seed=int(fx["audio_seed"])))
    ctx, tgt, vis = (torch.from_numpy(masks[k][:N]) for k in ("as_ctx", "as_tgt", "as_vis"))
    out = J.jepa_forward(P, audio, ctx, tgt, vis, mode="fp32", enc_heads=16, dec_heads=12, top_k=8)
    assert abs(float(out["loss"]) - float(fx["loss"])) < 2e-5 * float(fx["loss"])
    assert rel(out["local_features"][:, ::25, ::64], fx["local_features_slice"]) < 1e-4
    assert rel(out["contextual_features"][::7, ::16], fx["contextual_features_slice"]) < 1e-4
    assert rel(out["preds"][:, ::25, ::64], fx["preds_slice"]) < 1e-4
    assert rel(out["targets"][:, ::25, ::64], fx["targets_slice"]) < 1e-4
    out["loss"].backward()
    groups = {"conv": "extract_audio.", "feature_norms": "feature_norms.", "mapper": "post_extraction_mapper.",
              "encoder": "encoder.", "enc2dec": "encoder_to_decoder_mapper.", "decoder": "decoder.",
              "dec2enc": "decoder_to_encoder_mapper.", "mask_token": "mask_token"}
    want = dict(zip(fx["grad_group_names"].tolist(), fx["grad_group_norms"].tolist()))
    for g, pre in groups.items():
        tot = sum(float(P[n].grad.double().pow(2).sum()) for n in names if n.startswith(pre)) ** 0.5
        assert abs(tot - want[g]) < 1e-3 * want[g], (g, tot, want[g])
    assert rel(P["encoder.layers.23.linear1.weight"].grad[::256, ::64], fx["grad_slice::encoder.layers.23.linear1.weight"]) < 1e-3
    assert rel(P["encoder.layers.0.self_attn.in_proj_weight"].grad[::192, ::64], fx["grad_slice::encoder.layers.0.self_attn.in_proj_weight"]) < 1e-3


def test_channel_frontend_binaural_positions_and_channel_masks(golden_dir):
    """WavJEPA-Nat front-end pieces against the reference's outputs (tests/golden/channel_frontend.npz): the oracle's
    ConvChannelFeatureExtractor restatement (own and shared stacks, channel-major flatten), get_binaural_pos_embed (oracle AND the
    product's numpy restatement, bit-exact: float64 table), and the channel-based masks (bit-exact under the pinned generator
    sequence; product masker = reference order by default, channel-major on request)."""
    from oracle import masking_oracle as M
    from wavjepa_amd.extractors import ConvChannelFeatureExtractor
    from wavjepa_amd.masking import TimeInverseBlockMasker
    from wavjepa_amd.pos_embed import get_binaural_pos_embed
    fx = dict(np.load(os.path.join(golden_dir, "channel_frontend.npz")))
    spec = [(32, 10, 5)] + [(32, 3, 2)] * 4 + [(32, 2, 2)]
    audio = torch.from_numpy(fx["audio"])
    for tag, share in (("own", False), ("shared", True)):
        P = {"extract_audio." + k.split("::", 2)[2]: torch.from_numpy(v) for k, v in fx.items() if k.startswith(f"{tag}::sd::")}
        out = J.conv_frontend(P, audio, spec, "fp32")
        ref = torch.from_numpy(fx[f"{tag}::out"])
        assert out.shape == ref.shape == (2, 2 * 49, 32)
        assert float((out - ref).norm() / ref.norm()) < 2e-6
        ext = ConvChannelFeatureExtractor(conv_layers_spec=spec, in_channels=2, share_weights_over_channels=share)
        assert {k: tuple(v.shape) for k, v in ext.state_dict().items()} == {k[len("extract_audio."):]: tuple(v.shape) for k, v in P.items()}
        assert ext.total_patches(8000) == int(fx[f"{tag}::total_patches"]) == 98 and ext.embedding_dim == 32
    for d, t in ((768, 200), (64, 7)):
        for tab in (J.binaural_positions(d, t).numpy(), get_binaural_pos_embed(d, t)):
            assert tuple(tab.shape) == tuple(fx[f"binaural{d}_{t}_shape"])
            assert np.array_equal(tab[::max(1, t // 5), ::max(1, d // 16)], fx[f"binaural{d}_{t}_slice"])
            assert np.array_equal(tab[-1], fx[f"binaural{d}_{t}_row_last"]) and tab.sum() == float(fx[f"binaural{d}_{t}_sum"])
    base = int(fx["cb_base"])

    class Pinned:                       # k-th default_rng() call -> default_rng(base + k), as in make_golden.PinnedRng
        def __init__(self):
            self.k, self.orig = 0, np.random.default_rng

        def __call__(self, seed=None):
            g = self.orig(base + self.k)
            self.k += 1
            return g

    pin = Pinned()
    np.random.default_rng = pin
    try:
        c, t, v = TimeInverseBlockMasker(4, 0.65, 10, 0.25, 10, 0.1, channel_based_masking=True)(batch_size=3, n_times=400, in_channels=2)
    finally:
        np.random.default_rng = pin.orig
    assert np.array_equal(c.numpy(), fx["cb_ctx"]) and np.array_equal(t.numpy(), fx["cb_tgt"]) and np.array_equal(v.numpy(), fx["cb_vis"])
    pin = Pinned()
    np.random.default_rng = pin
    try:
        c2, t2, v2 = TimeInverseBlockMasker(4, 0.65, 10, 0.25, 10, 0.1, channel_based_masking=True, channel_major=True)(
            batch_size=3, n_times=400, in_channels=2)
    finally:
        np.random.default_rng = pin.orig
    # the same per-time masks, flattened "B (C S)": entry c*200 + s == reference entry s*2 + c
    assert np.array_equal(c2.numpy().reshape(3, 2, 200), fx["cb_ctx"].reshape(3, 200, 2).transpose(0, 2, 1))
    assert np.array_equal(t2.numpy().reshape(3, 4, 2, 200), fx["cb_tgt"].reshape(3, 4, 200, 2).transpose(0, 1, 3, 2))
    assert np.array_equal(v2.numpy().reshape(3, 4, 2, 200), fx["cb_vis"].reshape(3, 4, 200, 2).transpose(0, 1, 3, 2))


def test_scene_oracle_matches_reference_fixture(golden_dir):
    """oracle/scene_oracle.py against the reference's own generate_scenes_batch.py (fixture scene.npz): RIR convolution, noise
    aggregation, segmental-SNR mixing and the cases of generate_scene.  The reference computes in fp32 (tolerance 2e-5 of the
    output RMS); the oracle's float32 mode must land within the same band."""
    from oracle import scene_oracle as S

    fx = dict(np.load(os.path.join(golden_dir, "scene.npz")))
    src, noise, srir, nrir = fx["source"], fx["noise"], fx["source_rir"], fx["noise_rirs"]

    def close(got, ref, tol=2e-5):
        assert got.shape == ref.shape
        err = np.abs(got - ref).max() / np.sqrt((ref.astype(np.float64) ** 2).mean())
        assert err < tol, err

    for dt in (np.float64, np.float32):
        close(S.convolve_with_rir(src, srir, dt), fx["conv"])
        close(S.aggregate_noise(nrir, noise, dt), fx["agg"])
        close(S.generate_scene(srir, nrir, src, noise, fx["length"], fx["start"], fx["snr"], dt), fx["case_rir_noise"])
        close(S.generate_scene(srir, nrir, src, None, fx["length"], fx["start"], fx["snr"], dt), fx["case_rir_only"])
        close(S.generate_scene(None, nrir, src[:, None], noise[:, None], fx["length"], fx["start"], fx["snr"], dt), fx["case_noise_only"])
        close(S.add_noise(src[:1, None], noise[:1, None], 7.5, 500, 2500, dt), fx["mix_scalar"])
    # the full convolution really is the direct sum (independent of any FFT)
    b, c, t = 1, 1, 4321
    direct = sum(float(src[b, t - k]) * float(srir[b, c, k]) for k in range(min(t + 1, srir.shape[-1])))
    assert abs(S.convolve_with_rir(src, srir)[b, c, t] - direct) < 1e-9 * max(1.0, abs(direct))


def test_denoiser_oracle_matches_reference_fixture(golden_dir):
    """oracle/denoiser_oracle.py against the reference's own Denoiser.forward (fixture denoiser.npz: fp32, alpha 0.3, a small
    frozen reference JEPA as teacher): the three losses to 2e-5 relative, every parameter-gradient norm to 2e-4, three full
    gradients to 2e-4 relative L2; and the state_dict layout of wavjepa_amd.denoiser.Denoiser name by name."""
    from oracle import denoiser_oracle as DN
    fx = dict(np.load(os.path.join(golden_dir, "denoiser.npz")))
    P = {k[4:]: torch.from_numpy(v) for k, v in fx.items() if k.startswith("sd::")}
    PT = {k[5:]: torch.from_numpy(v) for k, v in fx.items() if k.startswith("tsd::")}
    spec = [(32, 10, 5)] + [(32, 3, 2)] * 4 + [(32, 2, 2)]
    names = [n for n in fx["grad_names"]]
    for n in names:
        P[n].requires_grad_(True)
    out = DN.denoiser_forward(P, PT, torch.from_numpy(fx["generated"]), torch.from_numpy(fx["clean"]), alpha=float(fx["alpha"]), spec=spec,
                              enc_heads=2, mode="fp32")
    for k in ("loss", "loss_clean", "loss_denoise_dereverb"):
        ref = float(fx[f"out::{k}"])
        assert abs(float(out[k]) - ref) < 2e-5 * abs(ref), (k, float(out[k]), ref)
    out["loss"].backward()
    for n, gn in zip(names, fx["grad_norms"]):
        assert abs(float(P[n].grad.norm()) - gn) < 2e-4 * gn + 1e-9, (n, float(P[n].grad.norm()), gn)
    for k in ("extract_audio.cnn.0.0.weight", "encoder.layers.1.linear1.weight", "encoder.norm.weight"):
        g, r = P[k].grad.double(), torch.from_numpy(fx[f"grad::{k}"]).double()
        assert float((g - r).norm() / r.norm()) < 2e-4, k
    from wavjepa_amd.denoiser import Denoiser
    from wavjepa_amd.extractors import ConvFeatureExtractor
    from wavjepa_amd.types import TransformerEncoderCFG, TransformerLayerCFG
    m = Denoiser(ConvFeatureExtractor(conv_layers_spec=spec, in_channels=1), TransformerLayerCFG.create(d_model=64, nhead=2),
                 TransformerEncoderCFG.create(num_layers=2), alpha=0.3)
    mine = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert sorted(mine) == sorted(str(n) for n in fx["state_dict_names"])
    for k, shp in mine.items():
        assert shp == tuple(fx[f"sd::{k}"].shape), k


def test_resample_tap_tables_match_the_pinned_closed_form(golden_dir):
    """tests/golden/resample_kernel.npz (make_resample_kernel.py): torchaudio's closed-form tap tables evaluated with exact rational tap
    times and scipy's Bessel function, for the reference's parameters.  The oracle's float64 table and the PRODUCT's float32 table
    (wavjepa_amd.resample.sinc_resample_kernel: what the GPU kernel and the loader workers convolve with) must equal them."""
    from oracle import resample_oracle as R
    from wavjepa_amd.resample import KAISER_BEST, sinc_resample_kernel
    import torch
    fx = np.load(os.path.join(golden_dir, "resample_kernel.npz"))
    for key in fx.files:
        if key.startswith("f32:"):
            # the table as torchaudio evaluates it for a FLOAT32 waveform (both reference call sites): every intermediate in float32.
            # The fixture rounds with numpy float32 scalars + scipy's i0, the product with torch's float32 kernels: equal up to the last
            # place of sin / i0 (4 float32 ulps of the largest tap), and both within 2e-6 of the exact closed form.
            o, n, m = key[4:].split("_")
            kw = dict(resampling_method="sinc_interp_kaiser", **KAISER_BEST) if m == "kaiser" else {}
            kp, *_ = sinc_resample_kernel(int(o), int(n), dtype=torch.float32, **kw)
            want, exact = fx[key], fx[key[4:]]
            big = float(np.abs(exact).max())
            assert kp.dtype == np.float32 and kp.shape == want.shape
            assert np.abs(kp - want).max() <= 4 * np.spacing(np.float32(big)), (key, np.abs(kp - want).max())
            assert np.abs(kp - exact).max() < 2e-6 * big + 2e-7, (key, np.abs(kp - exact).max())
            continue
        o, n, m = key.split("_")
        o, n = int(o), int(n)
        want = fx[key]
        if m == "kaiser":
            ko, width, orig, new = R.kernel(o, n, KAISER_BEST["lowpass_filter_width"], KAISER_BEST["rolloff"], "sinc_interp_kaiser", KAISER_BEST["beta"])
            kp, wp, op, npp = sinc_resample_kernel(o, n, resampling_method="sinc_interp_kaiser", **KAISER_BEST)
        else:
            ko, width, orig, new = R.kernel(o, n)
            kp, wp, op, npp = sinc_resample_kernel(o, n)
        assert ko.shape == want.shape == kp.shape and (width, orig, new) == (wp, op, npp), key
        assert np.abs(ko - want).max() < 1e-13, (key, np.abs(ko - want).max())
        assert kp.dtype == np.float32 and np.array_equal(kp, want.astype(np.float32)), key     # the product table is the float32 rounding


def test_resample_oracle_known_answers():
    """oracle/resample_oracle.py (torchaudio's published kaiser-sinc resampling, restated): properties that follow from the
    definition at the reference's call site (32 kHz -> 16 kHz, width 64, rolloff 0.9476, beta 14.77), and the host-side kernel
    table of the product path against the oracle's loop-built one."""
    from oracle import resample_oracle as R
    from wavjepa_amd.resample import KAISER_BEST, sinc_resample_kernel
    k, width, orig, new = R.kernel(32000, 16000, 64, KAISER_BEST["rolloff"], "sinc_interp_kaiser", KAISER_BEST["beta"])
    assert (k.shape, width, orig, new) == ((1, 274), 136, 2, 1)
    k2, w2, o2, n2 = sinc_resample_kernel(32000, 16000, resampling_method="sinc_interp_kaiser", **KAISER_BEST)
    assert (w2, o2, n2) == (width, orig, new) and np.abs(k - k2).max() < 1e-7
    assert abs(k.sum() - 1.0) < 1e-6                                                  # unit DC gain
    t32, t16 = np.arange(32000) / 32000.0, np.arange(16000) / 16000.0
    y = R.resample(np.sin(2 * np.pi * 1000 * t32)[None], 32000, 16000)[0]
    assert y.shape == (16000,) and np.abs(y[300:-300] - np.sin(2 * np.pi * 1000 * t16)[300:-300]).max() < 1e-8   # in-band: unchanged
    y = R.resample(np.sin(2 * np.pi * 12000 * t32)[None], 32000, 16000)[0]
    assert np.abs(y[300:-300]).max() < 1e-6                                           # above the new Nyquist: rejected
    assert R.resample(np.zeros((2, 3, 1001)), 32000, 16000).shape == (2, 3, 501)      # ceil(new * L / orig)
    k3, w3, o3, n3 = sinc_resample_kernel(44100, 16000)                               # hann default, non-trivial ratio
    kk, *_ = R.kernel(44100, 16000)
    assert k3.shape == (160, 2 * w3 + 441) and np.abs(kk - k3).max() < 1e-7


def test_hear_scene_oracle_matches_reference_fixture(golden_dir):
    """The evaluation-time twin (reference hear_api/heaRIR/scene_module/generate_scenes.py, fixture hear_scene.npz): per-clip RIR
    convolution, torchaudio-style add_noise with and without lengths, the noise fades, and generate_scene for noise of equal /
    greater / smaller length (the last with the reference's np.random placement replayed) and without noise."""
    from oracle import scene_oracle as S
    fx = dict(np.load(os.path.join(golden_dir, "hear_scene.npz")))
    sr = int(fx["sr"])
    src, srir, nr = fx["source"], fx["source_rir"], [fx["noise_rir0"], fx["noise_rir1"]]

    def close(got, ref, tol=2e-5):
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() / np.sqrt((ref.astype(np.float64) ** 2).mean()) < tol

    close(S.convolve_with_rir(src[None], srir[None])[0], fx["conv"])
    close(S.convolve_with_rir(src[None], srir[None, :1])[0], fx["conv_1d_rir"])
    close(S.add_noise_full(fx["mix_w"], fx["mix_n"], fx["mix_snr"]), fx["mix_full"])
    close(S.add_noise_full(fx["mix_w"], fx["mix_n"], fx["mix_snr"], fx["mix_len"]), fx["mix_lengths"])
    close(S.fade_noise(fx["noise_long"], src, sr), fx["fade_long"])
    close(S.fade_noise(fx["noise_short"], src, sr), fx["fade_short"])
    close(S.hear_generate_scene(srir, nr, src, fx["noise_same"], 7.0, sr), fx["scene_same"])
    close(S.hear_generate_scene(srir, nr, src, fx["noise_long"], 0.0, sr), fx["scene_long"])
    rng = np.random.RandomState(5)
    close(S.hear_generate_scene(srir, nr, src, fx["noise_short"], 12.0, sr, rng=rng), fx["scene_short"])
    close(S.hear_generate_scene(srir, [], src, None, 5.0, sr), fx["scene_no_noise"])


def test_hear_scene_iterators_read_the_dataset_layout(tmp_path):
    """SceneIterator / NoiseIterator (reference hear_api/heaRIR/iterators): JSON scene descriptions + .npy RIRs looked up by
    basename, padded / cut to 2 s at 32 kHz; .wav noise clips as float32 [channels, samples]."""
    import json
    from scipy.io import wavfile
    from hear_api.heaRIR.iterators import NoiseIterator, SceneIterator
    rirs, scenes, noises = tmp_path / "rirs", tmp_path / "scenes", tmp_path / "noise"
    for d in (rirs, scenes, noises):
        d.mkdir()
    rng = np.random.default_rng(0)
    np.save(rirs / "src_b.npy", rng.standard_normal((2, 1000)).astype(np.float32))
    np.save(rirs / "src_a.npy", rng.standard_normal((4, 70000)).astype(np.float32))
    np.save(rirs / "n0_b.npy", rng.standard_normal((2, 64000)).astype(np.float32))
    np.save(rirs / "n0_a.npy", rng.standard_normal((4, 10)).astype(np.float32))
    region = {"region": {"scene": {"source": {"rir": {"binaural_rir_path": "/elsewhere/src_b.npy", "ambisonic_rir_path": "/x/src_a.npy"},
                                              "azimuth": 30.0, "elevation": -5.0},
                                   "noise": [{"rir": {"binaural_rir_path": "/y/n0_b.npy", "ambisonic_rir_path": "n0_a.npy"}}]}}}
    (scenes / "s.json").write_text(json.dumps({"sampled_regions": [region]}))
    it = iter(SceneIterator(str(rirs), str(scenes)))
    src, nz, pos = next(it)
    assert src.shape == (2, 64000) and src.dtype == torch.float32 and len(nz) == 1 and nz[0].shape == (2, 64000) and pos == [30.0, -5.0]
    assert float(src[:, 1000:].abs().max()) == 0.0
    src, nz, _ = next(SceneIterator(str(rirs), str(scenes), with_noise=False, ambisonic=True))
    assert src.shape == (4, 64000) and nz == []
    wavfile.write(noises / "a.wav", 16000, (rng.standard_normal(800) * 8000).astype(np.int16))
    wavfile.write(noises / "b.wav", 8000, (rng.standard_normal((500, 2)) * 8000).astype(np.int16))
    seen = set()
    ni = iter(NoiseIterator(str(noises)))
    for _ in range(40):
        data, sr = next(ni)
        assert data.dtype == torch.float32 and float(data.abs().max()) < 1.0
        assert (tuple(data.shape), sr) in {((1, 800), 16000), ((2, 500), 8000)}
        seen.add(sr)
    assert seen == {16000, 8000}
