"""Scene augmentation through the C ABI (`wj_rir_convolve`, `wj_snr_mix`) and its host mirror
`wavjepa_amd.scene` against the oracle (fp64) and the reference's own outputs (tests/golden/scene.npz).  GPU only.
Tolerance: 2e-5 of the output RMS (fp32 FFT arithmetic on both sides; the oracle itself is exact to ~1e-12)."""
import os

import numpy as np
import pytest
import torch

from oracle import scene_oracle as S

pytestmark = pytest.mark.gpu
TOL = 2e-5


def dev():
    return torch.device("cuda:0")


def err(got, ref):
    got = got.detach().double().cpu().numpy() if isinstance(got, torch.Tensor) else got
    return float(np.abs(got - ref).max() / (np.sqrt((np.asarray(ref, dtype=np.float64) ** 2).mean()) + 1e-30))


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


@pytest.fixture()
def fx(golden_dir):
    return dict(np.load(os.path.join(golden_dir, "scene.npz")))


@pytest.mark.parametrize("fft_size", [0, 1024])
def test_scene_functions_match_reference_fixture_and_oracle(fx, fft_size, monkeypatch):
    """Every function of the host mirror on the reference's fixture inputs; fft_size 1024 makes the 6000-sample clips span 12
    signal blocks and the 700-tap RIRs 2 partitions (the block / partition bookkeeping the full-size runs rely on)."""
    from wavjepa_amd import scene
    monkeypatch.setattr(scene, "FFT_SIZE", fft_size)
    src, noise, srir, nrir = cu(fx["source"]), cu(fx["noise"]), cu(fx["source_rir"]), cu(fx["noise_rirs"])
    length, start, snr = cu(fx["length"]), cu(fx["start"]), cu(fx["snr"])
    conv = scene.convolve_with_rir(src, srir)
    assert conv.shape == (3, 2, 6000) and conv.dtype == torch.float32
    assert err(conv, fx["conv"]) < TOL
    assert err(conv, S.convolve_with_rir(fx["source"], fx["source_rir"])) < TOL
    agg = scene.aggregate_noise(nrir, noise)
    assert err(agg, fx["agg"]) < TOL
    assert err(scene.generate_scene(srir, nrir, src, noise, length, start, snr), fx["case_rir_noise"]) < TOL
    assert err(scene.generate_scene(srir, nrir, src, [None], length, start, snr), fx["case_rir_only"]) < TOL
    assert err(scene.generate_scene([None], nrir, src.unsqueeze(1), noise.unsqueeze(1), length, start, snr), fx["case_noise_only"]) < TOL
    assert err(scene.add_noise(src[:1].unsqueeze(1), noise[:1].unsqueeze(1), 7.5, 500, 2500), fx["mix_scalar"]) < TOL
    out = scene.generate_scene([None], nrir, src, [None], length, start, snr)
    assert out is src                                   # case 4: untouched


@pytest.mark.parametrize("B,C,T,L,fft", [(2, 1, 1, 1, 1024), (1, 3, 511, 513, 1024), (2, 2, 512, 512, 1024), (1, 1, 1537, 2049, 1024),
                                         (2, 1, 5000, 9000, 1024), (1, 2, 9000, 4097, 0), (1, 1, 100, 30000, 0)])
def test_rir_convolve_edge_shapes_vs_oracle(B, C, T, L, fft):
    """Ragged sizes: single samples, block-boundary lengths (Bk = 512 / 4096) +-1, RIR longer than the clip."""
    from wavjepa_amd import ops
    rng = np.random.default_rng(B * 1000 + T + L)
    x = rng.standard_normal((B, T)).astype(np.float32)
    h = (rng.standard_normal((B, C, L)) * np.exp(-np.arange(L) / (0.3 * L + 1))).astype(np.float32)
    y = torch.full((B, C, T), float("nan"), device=dev())
    dims = dict(B=B, C=C, T=T, L=L, fft_size=fft)
    ws = torch.empty(ops.workspace_bytes("wj_rir_convolve", **dims), device=dev(), dtype=torch.uint8)
    hx = cu(h)
    ops.rir_convolve(cu(x), hx, y, ws, h_stride_b=hx.stride(0), h_stride_c=hx.stride(1), **dims)
    ref = S.convolve_with_rir(x, h)
    assert err(y, ref) < TOL
    ops.rir_convolve(cu(x), hx, y, ws, h_stride_b=hx.stride(0), h_stride_c=hx.stride(1), accumulate=True, **dims)
    assert err(y, 2 * ref) < TOL                        # accumulate adds a second copy


def test_rir_convolve_rejects_bad_arguments():
    from wavjepa_amd import _abi, ops
    x = torch.zeros(1, 16, device=dev())
    h = torch.zeros(1, 1, 4, device=dev())
    y = torch.zeros(1, 1, 16, device=dev())
    ws = torch.zeros(1 << 16, device=dev(), dtype=torch.uint8)
    with pytest.raises(_abi.WavJepaHipError):
        ops.rir_convolve(x, h, y, ws, B=1, C=1, T=16, L=4, h_stride_b=4, h_stride_c=4, fft_size=4096)       # unsupported size
    with pytest.raises(_abi.WavJepaHipError):
        ops.rir_convolve(x, h, y, ws, B=1, C=1, T=0, L=4, h_stride_b=4, h_stride_c=4)
    with pytest.raises(_abi.WavJepaHipError):
        ops.rir_convolve(x, h, y, ws, B=1, C=1, T=16, L=4, h_stride_b=4, h_stride_c=2)                      # overlapping channels
    from wavjepa_amd import scene
    with pytest.raises(_abi.WavJepaHipError):
        scene.convolve_with_rir(torch.zeros(1, 16), torch.zeros(1, 1, 4))                                  # CPU tensors: no fallback


def test_full_size_scene_properties():
    """The denoiser stage's sizes (10 s at 32 kHz = 320 000 samples, 1.5 s RIRs = 48 000 taps, 8 clips): size-independent checks --
    a shifted unit impulse delays the clip exactly, the convolution is linear in the RIR, 64 random output samples equal their
    direct fp64 dot products, and two runs are bit-identical."""
    from wavjepa_amd import scene
    B, T, L = 8, 320000, 48000
    g = torch.Generator(device="cpu").manual_seed(5)
    x = torch.randn(B, T, generator=g)
    h1 = torch.randn(B, 1, L, generator=g) * torch.exp(-torch.arange(L) / 6000.0)
    h2 = torch.randn(B, 1, L, generator=g) * torch.exp(-torch.arange(L) / 9000.0)
    delta = torch.zeros(B, 1, L)
    shifts = [0, 1, 4095, 4096, 4097, 20000, 47999, 12345]
    for b, s in enumerate(shifts):
        delta[b, 0, s] = 1.0
    xd = x.to(dev())
    yd = scene.convolve_with_rir(xd, delta.to(dev())).cpu()
    for b, s in enumerate(shifts):
        want = torch.zeros(T)
        want[s:] = x[b, :T - s]
        assert float((yd[b, 0] - want).abs().max()) < 2e-5, (b, s)
    y1 = scene.convolve_with_rir(xd, h1.to(dev()))
    y2 = scene.convolve_with_rir(xd, h2.to(dev()))
    y12 = scene.convolve_with_rir(xd, (h1 + h2).to(dev()))
    rms = float(y12.double().pow(2).mean().sqrt())
    assert float((y1 + y2 - y12).abs().max()) < 3 * TOL * rms
    assert torch.equal(y1, scene.convolve_with_rir(xd, h1.to(dev())))
    rng = np.random.default_rng(0)
    y1c = y1.cpu().numpy()
    xn, hn = x.numpy().astype(np.float64), h1.numpy().astype(np.float64)
    for _ in range(64):
        b, t = int(rng.integers(B)), int(rng.integers(T))
        k = min(t + 1, L)
        direct = float(np.dot(xn[b, t - k + 1:t + 1][::-1], hn[b, 0, :k]))
        assert abs(y1c[b, 0, t] - direct) < TOL * rms, (b, t)


def test_snr_mix_vs_oracle_and_reproducible():
    from wavjepa_amd import scene
    B, C, T = 5, 2, 70001
    rng = np.random.default_rng(3)
    s = rng.standard_normal((B, C, T)).astype(np.float32)
    n = (rng.standard_normal((B, C, T)) * 0.3).astype(np.float32)
    start = np.array([0, 10, 69990, 35000, 70001], dtype=np.int64)        # last: empty window -> a = 0
    length = np.array([70001, 1, 100, 20000, 5], dtype=np.int64)          # third: window runs past the end (clipped)
    snr = np.array([0.0, 10.0, -5.0, 20.0, 3.0], dtype=np.float32)
    out = scene.add_noise(cu(s), cu(n), cu(snr), cu(start), cu(length))
    ref = S.add_noise(s, n, snr, start, length)
    assert err(out, ref) < TOL
    assert torch.equal(out, scene.add_noise(cu(s), cu(n), cu(snr), cu(start), cu(length)))
    assert torch.equal(out[4].cpu(), torch.from_numpy(s[4]))              # empty window: source untouched


def test_nat_workload_scene_front_end_feeds_two_channel_step():
    """BASELINE config 4 wired end to end: `NatSceneSource` builds binaural scenes with the scene kernels (source RIR + noise RIRs +
    SNR mix), the crops go through the 2-channel `ConvChannelFeatureExtractor` model and one optimisation step runs; the scene the
    source produced equals the oracle's, and `bench.py --workload 2s-nat` runs as a child process."""
    import json
    import subprocess
    import sys
    from wavjepa_amd.data import NatSceneSource
    from wavjepa_amd.masking import TimeInverseBlockMasker
    masker = TimeInverseBlockMasker(4, 0.65, 10, 0.25, 10, 0.1, channel_based_masking=True, channel_major=True)
    src = NatSceneSource(masker, batch_size=2, samples_per_audio=2, n_tokens=400, seed=3, n_mask_sets=1, device=dev(), seconds=1.0)
    state = src.gen.get_state()
    audio, ctx, tgt, vis = src.next_batch()
    assert audio.shape == (2, 2, 16000) and ctx.shape == (2, 2, 400) and bool(torch.isfinite(audio).all())
    src.gen.set_state(state)                      # replay the draws of next_batch() for the oracle
    g = src.gen
    s = torch.randn(2, 16000, generator=g, device=dev())
    n = torch.randn(2, 16000, generator=g, device=dev())
    snr = torch.rand(2, generator=g, device=dev()) * 35.0 + 5.0
    length = torch.randint(4000, 16000, (2,), generator=g, device=dev())
    start = ((16000 - length).float() * torch.rand(2, generator=g, device=dev())).long()
    conv = S.convolve_with_rir(s.cpu().numpy(), src.source_rir.cpu().numpy())
    agg = S.aggregate_noise(src.noise_rirs.cpu().numpy(), n.cpu().numpy())
    ref = S.add_noise(conv, agg, snr.cpu().numpy(), start.cpu().numpy(), length.cpu().numpy())
    assert err(audio, ref) < TOL
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--workload", "2s-nat", "--steps", "2", "--warmup", "1", "--clips-per-gpu", "16",
           "--dense-steps", "0", "--no-cpu-baseline", "--no-profile"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["config"]["workload_name"] == "2s-nat" and line["config"]["seq_len"] == 400 and line["value"] > 0
    assert np.isfinite(line["final_loss"])


def test_hear_api_scene_twin_matches_reference_fixture(golden_dir):
    """hear_api/heaRIR (evaluation-time augmentation, reference hear_api/heaRIR/): every function of scene_module against the
    reference's own outputs (hear_scene.npz) and Augmenter.augment against the oracle."""
    from hear_api.heaRIR import Augmenter
    from hear_api.heaRIR.scene_module import generate_scenes as H
    fx = dict(np.load(os.path.join(golden_dir, "hear_scene.npz")))
    sr = int(fx["sr"])
    src, srir = cu(fx["source"]), cu(fx["source_rir"])
    nr = [cu(fx["noise_rir0"]), cu(fx["noise_rir1"])]
    assert err(H.convolve_with_rir(src, srir), fx["conv"]) < TOL
    assert err(H.convolve_with_rir(src, srir[0]), fx["conv_1d_rir"]) < TOL
    assert err(H.add_noise(cu(fx["mix_w"]), cu(fx["mix_n"]), cu(fx["mix_snr"])), fx["mix_full"]) < TOL
    assert err(H.add_noise(cu(fx["mix_w"]), cu(fx["mix_n"]), cu(fx["mix_snr"]), cu(fx["mix_len"])), fx["mix_lengths"]) < TOL
    assert err(H.fade_noise(cu(fx["noise_long"]), src, sr), fx["fade_long"]) < 1e-6
    assert err(H.fade_noise(cu(fx["noise_short"]), src, sr), fx["fade_short"]) < 1e-6
    assert err(H.generate_scene(srir, nr, src.clone(), cu(fx["noise_same"]), 7.0, sr), fx["scene_same"]) < TOL
    assert err(H.generate_scene(srir, nr, src.clone(), cu(fx["noise_long"]), 0.0, sr), fx["scene_long"]) < TOL
    np.random.seed(5)
    assert err(H.generate_scene(srir, nr, src.clone(), cu(fx["noise_short"]), 12.0, sr), fx["scene_short"]) < TOL
    assert err(H.generate_scene(srir, [], src.clone(), None, 5.0, sr), fx["scene_no_noise"]) < TOL

    class OneScene:                                   # stands in for SceneIterator: one fixed scene, RIR longer than the clip
        def __next__(self):
            return torch.from_numpy(fx["source_rir"]), [torch.from_numpy(fx["noise_rir0"])], [0.0, 0.0]
    fsr = 1000                                       # (sr only sets the 0.2 s fades: 200 samples here)
    aug = Augmenter(OneScene(), sr=fsr, snr=9)
    clip = cu(fx["source"][:500])                    # 500 samples < 800 taps: the clip is zero-extended, the result cut back
    pad = np.concatenate([fx["source"][:500], np.zeros(300, np.float32)])
    ref = S.hear_generate_scene(fx["source_rir"], [fx["noise_rir0"]], pad, fx["noise_same"][:500], 9.0, fsr,
                                rng=np.random.RandomState(0))
    np.random.seed(0)
    out = aug.augment(clip.clone(), cu(fx["noise_same"][:500]))
    assert out.shape == (2, 500)
    assert err(out, ref[:, :500]) < TOL
    assert Augmenter(None, sr=sr, snr=None).augment(clip).shape == (1, 500)


def test_train_py_runs_on_flac_shards(tmp_path):
    """The real data path end to end: tar shards of FLAC clips -> WebAudioDataModule worker processes (native decoder,
    resampling, -14 dBFS, 10 s pad, masks) -> train.py's loop on the GPU for three optimisation steps."""
    import io
    import subprocess
    import sys
    import tarfile
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import flac_encoder as E
    rng = np.random.default_rng(0)
    shard = tmp_path / "shard-000.tar"
    with tarfile.open(shard, "w") as tf:
        for i in range(6):
            rate = (16000, 32000)[i % 2]
            n = int(rate * 2.6)
            pcm = np.round(6000 * np.sin(2 * np.pi * (200 + 40 * i) * np.arange(n) / rate) + 500 * rng.standard_normal(n)).astype(np.int64)[:, None]
            data = E.encode(pcm, rate, 16, blocksize=4096, subframes=dict(kind="fixed", order=2, porder=2))
            ti = tarfile.TarInfo(f"clip{i:03d}.flac")
            ti.size = len(data)
            tf.addfile(ti, io.BytesIO(data))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "train.py"), "data=audioset", f"data.data_dirs={shard}", "trainer.batch_size=2", "trainer.steps=3",
           "trainer.log_every_n_steps=1", f"save_dir={tmp_path / 'runs'}"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    losses = [float(ln.split("loss")[1].split()[0]) for ln in r.stdout.splitlines() if ln.startswith("step ")]
    assert len(losses) >= 3 and all(np.isfinite(losses)), r.stdout[-1500:]
    ckpts = list((tmp_path / "runs" / "saved_models_jepa_new_masking").rglob("last.ckpt"))       # the reference's run-identity directory
    assert len(ckpts) == 1 and "Data=AudioSet" in str(ckpts[0]) and "BatchSize=2" in str(ckpts[0])
