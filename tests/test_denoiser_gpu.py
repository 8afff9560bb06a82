"""Denoiser stage (reference wavjepa/denoiser.py) through the C ABI against the oracle and the reference fixture.
GPU only.  Tolerances as for the JEPA step: losses 1e-3 relative to the oracle's bf16 flow, parameter-gradient groups 3e-2 relative
L2 (two independent bf16 pipelines); fp32 streaming kernels (resampling, grouped MSE) 2e-5 / 1e-6."""
import os

import numpy as np
import pytest
import torch

import synth
from oracle import denoiser_oracle as DN
from oracle import jepa_oracle as J
from oracle import resample_oracle as RS
from oracle import scene_oracle as S

pytestmark = pytest.mark.gpu
SPEC = [(64, 10, 5)] + [(64, 3, 2)] * 4 + [(64, 2, 2)]


def dev():
    return torch.device("cuda:0")


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("orig,new,kw", [(32000, 16000, dict(lowpass_filter_width=64, rolloff=0.9475937167399596, method="sinc_interp_kaiser",
                                                             beta=14.769656459379492)),
                                         (48000, 32000, dict(lowpass_filter_width=6, rolloff=0.99, method="sinc_interp_hann", beta=None)),
                                         (16000, 24000, dict(lowpass_filter_width=16, rolloff=0.9, method="sinc_interp_kaiser", beta=None))])
def test_resample_fir_vs_oracle(orig, new, kw):
    from wavjepa_amd.resample import resample_waveform
    rng = np.random.default_rng(orig + new)
    x = rng.standard_normal((3, 2, 10007)).astype(np.float32)
    y = resample_waveform(torch.from_numpy(x).to(dev()), orig, new, lowpass_filter_width=kw["lowpass_filter_width"], rolloff=kw["rolloff"],
                          resampling_method=kw["method"], beta=kw["beta"])
    ref = RS.resample(x, orig, new, **kw)
    assert tuple(y.shape) == ref.shape
    assert float(np.abs(y.cpu().numpy() - ref).max()) < 2e-5 * float(np.sqrt((ref ** 2).mean()))


def test_resample_reference_call_site_and_unsupported_table():
    from wavjepa_amd import _abi
    from wavjepa_amd.resample import resample, resample_waveform
    x = torch.randn(2, 1, 320000, device=dev())
    y = resample(x, resample_sr=16000)                          # denoiser.py:29-42
    assert y.shape == (2, 1, 160000)
    tone = torch.from_numpy(np.sin(2 * np.pi * 440.0 * np.arange(320000) / 32000.0)).float().to(dev())[None, None]     # phases in fp64
    out = resample(tone, 16000)[0, 0]
    want = torch.from_numpy(np.sin(2 * np.pi * 440.0 * np.arange(160000) / 16000.0)).float().to(dev())
    assert float((out - want)[1000:-1000].abs().max()) < 2e-6
    assert resample(x, resample_sr=32000) is x                  # same rate: untouched
    with pytest.raises(_abi.WavJepaHipError):
        resample_waveform(x, 44100, 16000)                      # 160 x 475 taps + window do not fit the kernel's LDS budget
    with pytest.raises(_abi.WavJepaHipError):
        resample(torch.zeros(1, 1, 100), 16000)                 # CPU tensor: no fallback


def test_mse_groups_vs_torch():
    from wavjepa_amd import ops
    n, G = 3 * 200 * 128 + 5, 2
    g = torch.Generator().manual_seed(4)
    p = torch.randn(G, n, generator=g).to(dev())
    t = torch.randn(n, generator=g).to(dev())
    w = torch.tensor([0.3, 0.7], device=dev())
    loss = torch.zeros(1 + G, device=dev())
    ws = torch.empty(ops.workspace_bytes("wj_mse_groups", G=G, n=n) // 4, device=dev())
    dp = torch.empty_like(p)
    gs = torch.tensor([2.5], device=dev())
    ops.mse_groups(p, t, w, loss, ws, n=n, G=G, dpreds=dp, gscale=gs)
    pr = p.double().clone().requires_grad_(True)                 # fp64 yard-stick; the kernel accumulates in fp32
    l0 = torch.nn.functional.mse_loss(pr[0], t.double())
    l1 = torch.nn.functional.mse_loss(pr[1], t.double())
    tot = 0.3 * l0 + 0.7 * l1
    (2.5 * tot).backward()
    assert abs(float(loss[1]) - float(l0)) < 1e-5 * float(l0) and abs(float(loss[2]) - float(l1)) < 1e-5 * float(l1)
    assert abs(float(loss[0]) - float(tot)) < 1e-5 * float(tot)
    assert rel(dp, pr.grad) < 1e-6
    l2 = torch.zeros_like(loss)
    ops.mse_groups(p, t, w, l2, ws, n=n, G=G)
    assert torch.equal(l2, loss)                                # fixed-order fold: bit-identical


def build(d=128, heads=2, layers=2, alpha=0.3, seed=11):
    from wavjepa_amd.denoiser import Denoiser
    from wavjepa_amd.extractors import ConvFeatureExtractor
    from wavjepa_amd.jepa import JEPA
    from wavjepa_amd.types import TransformerEncoderCFG, TransformerLayerCFG
    den = Denoiser(ConvFeatureExtractor(conv_layers_spec=SPEC, in_channels=1), TransformerLayerCFG.create(d_model=d, nhead=heads),
                   TransformerEncoderCFG.create(num_layers=layers), alpha=alpha, lr=1e-3, nr_samples_per_audio=2)
    tea = JEPA(feature_extractor=ConvFeatureExtractor(conv_layers_spec=SPEC, in_channels=1),
               transformer_encoder_cfg=TransformerEncoderCFG.create(num_layers=layers),
               transformer_encoder_layers_cfg=TransformerLayerCFG.create(d_model=d, nhead=heads),
               transformer_decoder_cfg=TransformerEncoderCFG.create(num_layers=2),
               transformer_decoder_layers_cfg=TransformerLayerCFG.create(d_model=64, nhead=2), average_top_k_layers=2,
               process_audio_seconds=2.01, nr_samples_per_audio=2)
    P, PT = {}, {}
    for mod, store, sd_seed in ((den, P, seed), (tea, PT, seed + 1)):
        shapes = {k: tuple(v.shape) for k, v in mod.state_dict().items()}
        sd = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, seed=sd_seed).items() if k in shapes}
        sd["pos_encoding_encoder"] = J.sincos_positions(d, 200)
        if "pos_encoding_decoder" in shapes:
            sd["pos_encoding_decoder"] = J.sincos_positions(64, 200)
        mod.load_state_dict(sd)
        store.update({k: v.clone().to(dev()) for k, v in sd.items()})
    den = den.to(dev())
    den._set_teacher(tea.to(dev()))
    return den, P, PT


def group_of(name):
    for g in ("extract_audio", "feature_norms", "post_extraction_mapper", "encoder"):
        if name.startswith(g):
            return g
    return "other"


def test_denoiser_forward_backward_parity_vs_oracle():
    den, P, PT = build()
    clean = torch.from_numpy(synth.synth_audio(3, 1, 32159, seed=41)).to(torch.bfloat16).to(dev())
    noise = torch.from_numpy(synth.synth_audio(3, 1, 32159, seed=42)).to(dev())
    generated = (clean.float() + 0.5 * noise).to(torch.bfloat16)
    out = den(generated, clean)
    names = [k for k in P if k != "pos_encoding_encoder"]
    for k in names:
        P[k].requires_grad_(True)
    ref = DN.denoiser_forward(P, PT, generated, clean, alpha=0.3, spec=SPEC, enc_heads=2, mode="bf16")
    for k in ("loss", "loss_clean", "loss_denoise_dereverb"):
        a, b = float(out[k].detach()), float(ref[k].detach())
        assert abs(a - b) < 1e-3 * abs(b), (k, a, b)
    assert abs(float(out["loss"].detach()) - (0.3 * float(out["loss_clean"]) + 0.7 * float(out["loss_denoise_dereverb"]))) < 1e-6
    out["loss"].backward()
    ref["loss"].backward()
    got = dict(den.named_parameters())
    num, den_ = {}, {}
    for k in names:
        g = group_of(k)
        a, b = got[k].grad.double(), P[k].grad.double()
        num[g] = num.get(g, 0.0) + float((a - b).pow(2).sum())
        den_[g] = den_.get(g, 0.0) + float(b.pow(2).sum())
    errs = {g: (num[g] / max(den_[g], 1e-300)) ** 0.5 for g in num}
    print("denoiser: losses", {k: float(out[k].detach()) for k in ("loss", "loss_clean", "loss_denoise_dereverb")}, "grad rel errors:", errs)
    for g, e in errs.items():
        assert e < 3e-2, (g, e)
    # student features of the inference entry == the oracle's
    cf = den.encoder_forward(clean)
    assert rel(cf, ref["contextual_features_clean"].float()) < 1e-2


def test_denoiser_matches_reference_fixture(golden_dir):
    """The reference's own Denoiser.forward (fp32) on the fixture's weights and clips: the bf16 HIP path lands within 1e-2 of its
    three losses (the oracle's bf16 flow sits at the same distance)."""
    from wavjepa_amd.denoiser import Denoiser
    from wavjepa_amd.extractors import ConvFeatureExtractor
    from wavjepa_amd.jepa import JEPA
    from wavjepa_amd.types import TransformerEncoderCFG, TransformerLayerCFG
    fx = dict(np.load(os.path.join(golden_dir, "denoiser.npz")))
    spec = [(32, 10, 5)] + [(32, 3, 2)] * 4 + [(32, 2, 2)]
    den = Denoiser(ConvFeatureExtractor(conv_layers_spec=spec, in_channels=1), TransformerLayerCFG.create(d_model=64, nhead=2),
                   TransformerEncoderCFG.create(num_layers=2), alpha=float(fx["alpha"]))
    den.load_state_dict({k[4:]: torch.from_numpy(v) for k, v in fx.items() if k.startswith("sd::")})
    tea = JEPA(feature_extractor=ConvFeatureExtractor(conv_layers_spec=spec, in_channels=1),
               transformer_encoder_cfg=TransformerEncoderCFG.create(num_layers=2),
               transformer_encoder_layers_cfg=TransformerLayerCFG.create(d_model=64, nhead=2),
               transformer_decoder_cfg=TransformerEncoderCFG.create(num_layers=2),
               transformer_decoder_layers_cfg=TransformerLayerCFG.create(d_model=32, nhead=1), average_top_k_layers=2,
               process_audio_seconds=2.01, nr_samples_per_audio=2)
    tea.load_state_dict({k[5:]: torch.from_numpy(v) for k, v in fx.items() if k.startswith("tsd::")})
    den = den.to(dev())
    den._set_teacher(tea.to(dev()))
    with torch.no_grad():
        out = den(torch.from_numpy(fx["generated"]).to(dev()), torch.from_numpy(fx["clean"]).to(dev()))
    for k in ("loss", "loss_clean", "loss_denoise_dereverb"):
        assert abs(float(out[k]) - float(fx[f"out::{k}"])) < 1e-2 * abs(float(fx[f"out::{k}"])), (k, float(out[k]), float(fx[f"out::{k}"]))


def test_denoiser_batch_hook_vs_oracle():
    """on_after_batch_transfer (denoiser.py:217-309): scene -> 32 k -> 16 k resampling -> shared crops -> per-crop normalisation ->
    bf16 -> shared shuffle, against the oracles with the hook's own random draws replayed."""
    den, _, _ = build()
    B, T32, L, n = 3, 96000, 900, 2
    g = torch.Generator().manual_seed(9)
    audio = torch.randn(B, T32, generator=g)
    noise = torch.randn(B, T32, generator=g)
    srir = torch.randn(B, 2, L, generator=g) * torch.exp(-torch.arange(L) / 120.0)
    nrir = torch.randn(B, n, 2, L, generator=g) * torch.exp(-torch.arange(L) / 200.0)
    length = torch.tensor([30000, 96000, 9000])
    start = torch.tensor([500, 0, 70000])
    snr = torch.tensor([10.0, 0.0, 5.0])
    batch = tuple(t.to(dev()) for t in (audio, srir, noise, length, start, nrir, snr))
    torch.manual_seed(123)
    gen_c, clean_c = den.on_after_batch_transfer(batch, 0)
    S_ = den.nr_samples_per_audio
    assert gen_c.shape == clean_c.shape == (B * S_, 1, den.target_length) and gen_c.dtype == torch.bfloat16
    torch.manual_seed(123)
    starts = torch.randint(0, 48000 - den.target_length + 1, (B, S_), device=dev()).cpu()
    idx = torch.randperm(B * S_)
    scene_ref = S.generate_scene(srir.numpy(), nrir.numpy(), audio.numpy(), noise.numpy(), length.numpy(), start.numpy(), snr.numpy())
    gen16 = torch.from_numpy(RS.resample(scene_ref, 32000, 16000)).float()
    clean16 = torch.from_numpy(RS.resample(audio.numpy()[:, None, :], 32000, 16000)).float()
    for got, src in ((gen_c, gen16), (clean_c, clean16)):
        want = J.crop_normalize(src, starts, den.target_length, idx)
        assert rel(got.float(), want.float()) < 4e-3            # bf16 outputs of two fp32 pipelines


def test_denoiser_training_steps_reduce_the_loss():
    den, _, _ = build(alpha=0.0)
    opt = den.configure_optimizers()["optimizer"]
    opt.param_groups[0]["lr"] = 1e-3
    clean = torch.from_numpy(synth.synth_audio(4, 1, 32159, seed=51)).to(torch.bfloat16).to(dev())
    gen = (clean.float() + 0.3 * torch.from_numpy(synth.synth_audio(4, 1, 32159, seed=52)).to(dev())).to(torch.bfloat16)
    before = {k: v.detach().clone() for k, v in den.named_parameters() if v.requires_grad}
    losses = []
    for step in range(6):
        out = den.training_step((gen, clean), step)
        out["loss"].backward()
        opt.step()
        losses.append(float(out["loss"].detach()))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    assert float(out["loss"].detach()) == pytest.approx(float(out["loss_denoise_dereverb"]), rel=1e-6)     # alpha = 0
    changed = sum(int(not torch.equal(before[k], v.detach())) for k, v in den.named_parameters() if v.requires_grad)
    assert changed == len(before)
    mine = {id(p) for p in den.parameters()}
    assert not any(id(p) in mine for p in den.teacher.parameters())          # the teacher is outside parameters() / the optimiser
    # ... but in the state_dict, under the reference's names (its Denoiser holds the teacher as a sub-module): strict interchange
    sd = den.state_dict()
    tsd = den.teacher.state_dict()
    assert {k for k in sd if k.startswith("teacher.")} == {"teacher." + k for k in tsd}
    assert all(torch.equal(sd["teacher." + k], v) for k, v in tsd.items())
    poisoned = {k: (torch.full_like(v, 0.5) if k == "teacher.feature_norms.weight" else v) for k, v in sd.items()}
    den.load_state_dict(poisoned)                                             # strict: every key known, teacher entries routed
    assert float(den.teacher.feature_norms.weight.mean()) == 0.5
    den.load_state_dict({k: v for k, v in sd.items() if not k.startswith("teacher.")})   # a student-only dict still strict-loads


def test_denoise_py_runs_end_to_end(tmp_path):
    """denoise.py (reference denoise.py): a Lightning-format WavJEPA checkpoint initialises the student and is the frozen teacher,
    WebAudioDataModuleDenoiser worker processes stream FLAC clips + RIR sets + noise, the batch hook generates the scenes on the
    GPU, three optimisation steps run under the 1.0 gradient-norm clip."""
    import io
    import subprocess
    import sys
    import tarfile
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import flac_encoder as E
    from wavjepa_amd.extractors import ConvFeatureExtractor
    from wavjepa_amd.jepa import JEPA
    from wavjepa_amd.types import TransformerEncoderCFG, TransformerLayerCFG
    rng = np.random.default_rng(0)

    def shard(path, members):
        with tarfile.open(path, "w") as tf:
            for name, data in members:
                ti = tarfile.TarInfo(name)
                ti.size = len(data)
                tf.addfile(ti, io.BytesIO(data))

    def npy(a):
        b = io.BytesIO()
        np.save(b, a)
        return b.getvalue()
    clips = []
    for i in range(4):
        n = int(32000 * 2.5)
        pcm = np.round(6000 * np.sin(2 * np.pi * (180 + 50 * i) * np.arange(n) / 32000) + 400 * rng.standard_normal(n)).astype(np.int64)[:, None]
        clips.append((f"clip{i}.flac", E.encode(pcm, 32000, 16, subframes=dict(kind="fixed", order=2, porder=2))))
    shard(tmp_path / "audio.tar", clips)
    decay = np.exp(-np.arange(3000) / 400.0)
    shard(tmp_path / "rir.tar", [(f"r{i}.npy", npy((rng.standard_normal((3, 2, 3000)) * decay).astype(np.float32))) for i in range(3)])
    shard(tmp_path / "noise.tar", [(f"n{i}.npy", npy(rng.standard_normal(32000 * (3 + 9 * i)).astype(np.float32))) for i in range(2)])
    torch.manual_seed(0)
    tea = JEPA(feature_extractor=ConvFeatureExtractor(conv_layers_spec=[(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512, 2, 2)], in_channels=1),
               transformer_encoder_cfg=TransformerEncoderCFG.create(), transformer_encoder_layers_cfg=TransformerLayerCFG.create(),
               transformer_decoder_cfg=TransformerEncoderCFG.create(), transformer_decoder_layers_cfg=TransformerLayerCFG.create(d_model=384),
               process_audio_seconds=2.01)
    sd = {k.replace("encoder.", "encoder._orig_mod.", 1) if k.startswith("encoder.") else k: v for k, v in tea.state_dict().items()}
    ckpt = tmp_path / "teacher.ckpt"
    torch.save({"state_dict": sd, "hyper_parameters": {}, "global_step": 375000}, ckpt)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "denoise.py"), f"data.data_dir={tmp_path / 'audio.tar'}", f"data.rir_dir={tmp_path / 'rir.tar'}",
           f"data.noise_dir={tmp_path / 'noise.tar'}", f"trainer.teacher_ckpt_weights={ckpt}", "trainer.batch_size=2", "trainer.steps=3",
           "trainer.log_every_n_steps=1", f"save_dir={tmp_path / 'runs'}"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    losses = [float(ln.split("loss")[1].split()[0]) for ln in r.stdout.splitlines() if ln.startswith("step ")]
    assert len(losses) >= 3 and all(np.isfinite(losses)), r.stdout[-1500:]
    # the student started from the teacher's weights: before any update its clean-clip features equal the targets, so the loss of the
    # generated scene is what remains -- positive, and far below the ~2 of two unrelated random models
    assert 0.0 < losses[0] < 1.5, losses
    saved = [p for p in (tmp_path / "runs").rglob("last.ckpt")]
    assert saved and "state_dict" in torch.load(saved[0], weights_only=False)
