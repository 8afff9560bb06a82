"""The real data path on the CPU: native FLAC decoder (round trips through an independent encoder that can force
every bitstream feature, CRC / MD5 / truncation detection), the per-file preparation functions against the reference's own, the
CPU resampler against the oracle, and the shard-based data module end to end on temporary tar shards."""
import io
import os
import sys
import tarfile
import warnings

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import flac_encoder as E  # noqa: E402
from wavjepa_amd import audio_io as A  # noqa: E402


def tone_pcm(n, ch, bps, seed=0, rate=16000):
    rng = np.random.default_rng(seed)
    t = np.arange(n)
    amp = (1 << (bps - 1)) * 0.4
    cols = []
    for c in range(ch):
        x = amp * np.sin(2 * np.pi * (220.0 * (c + 1)) * t / rate) + amp * 0.02 * rng.standard_normal(n)
        cols.append(np.round(x).astype(np.int64))
    if ch == 2:
        cols[1] = (cols[0] * 0.8).astype(np.int64) + rng.integers(-20, 20, n)
    return np.stack(cols, 1)


SPECS = [dict(kind="verbatim"), dict(kind="fixed", order=0), dict(kind="fixed", order=1, porder=2), dict(kind="fixed", order=2, porder=3, rice2=True),
         dict(kind="fixed", order=3, params=[3, 9, 0]), dict(kind="fixed", order=4, porder=4, escape_parts=(0, 1, 5)),
         dict(kind="lpc", order=1, coefs=[1000], shift=10, precision=12), dict(kind="lpc", order=3, coefs=[1900, -1100, 150], shift=10, precision=12, porder=1),
         dict(kind="lpc", order=8, coefs=[700, -300, 120, -60, 30, -20, 10, -5], shift=9, precision=11, porder=2, rice2=True),
         dict(kind="lpc", order=32, coefs=[3] * 32, shift=7, precision=5, porder=0)]


@pytest.mark.parametrize("stereo", ["independent", "left_side", "right_side", "mid_side"])
def test_flac_round_trip_every_subframe_type_and_channel_assignment(stereo):
    pcm = tone_pcm(9000, 2, 16)
    for spec in SPECS:
        data = E.encode(pcm, 16000, 16, blocksize=4096, stereo=stereo, subframes=spec)
        out, si = A.decode_flac_pcm(data, verify_md5=True)
        assert np.array_equal(out, pcm), (stereo, spec)
        assert (si.sample_rate, si.channels, si.bits_per_sample, si.total_samples) == (16000, 2, 16, 9000)


@pytest.mark.parametrize("bps", [8, 12, 16, 20, 24])
@pytest.mark.parametrize("blocksize", [192, 200, 1000, 1152, 4096, 5000])
def test_flac_bit_depths_and_block_sizes(bps, blocksize):
    """Table block sizes, explicit 8-bit (200) and 16-bit (1000, 5000) sizes, a short last frame; mono and 3 channels."""
    for ch in (1, 3):
        pcm = tone_pcm(2 * blocksize + 77, ch, bps, seed=bps + blocksize)
        data = E.encode(pcm, 44100, bps, blocksize=blocksize, subframes=dict(kind="fixed", order=2, porder=0))
        out, si = A.decode_flac_pcm(data, verify_md5=True)
        assert np.array_equal(out, pcm) and si.bits_per_sample == bps and si.channels == ch
        wav, sr = A.decode_flac(data)
        assert sr == 44100 and wav.shape == (ch, pcm.shape[0]) and wav.dtype == torch.float32
        assert np.allclose(wav.numpy(), pcm.T / float(1 << (bps - 1)), atol=0)


def test_flac_header_variants_and_special_subframes():
    pcm = tone_pcm(6000, 2, 16, seed=5)
    pcm[:, 1] = 1234                                                     # a constant channel
    pcm[:, 0] &= ~0x7                                                    # three wasted bits
    sub = [dict(kind="fixed", order=2, wasted=3), dict(kind="constant")]
    blocks = [(4, b"\x00" * 40), (1, b"\x00" * 100)]                     # a VORBIS_COMMENT-sized block and PADDING after STREAMINFO
    for kw in (dict(), dict(variable=True), dict(sr_in_header="streaminfo", bps_in_header=False), dict(id3=True, extra_blocks=blocks),
               dict(total_in_header=False, md5=False)):
        data = E.encode(pcm, 32000, 16, blocksize=1024, subframes=sub, **kw)
        out, si = A.decode_flac_pcm(data, verify_md5=True)
        assert np.array_equal(out, pcm), kw
    for rate, mode in ((37000, "khz"), (12345, "hz"), (96010, "tens")):  # explicit sample-rate codes 12 / 13 / 14
        data = E.encode(pcm, rate, 16, blocksize=1024, subframes=sub, sr_in_header=mode)
        out, si = A.decode_flac_pcm(data)
        assert np.array_equal(out, pcm) and si.sample_rate == rate


def test_flac_detects_corruption():
    pcm = tone_pcm(5000, 1, 16, seed=9)
    data = bytearray(E.encode(pcm, 16000, 16, blocksize=1024))
    good = bytes(data)
    assert np.array_equal(A.decode_flac_pcm(good, verify_md5=True)[0], pcm)
    start = good.index(b"\xff\xf8", 42)
    bad = bytearray(good)
    bad[start + 40] ^= 0x10                                              # a bit in the first frame's body: frame CRC-16
    with pytest.raises(A.AudioDecodeError, match="CRC|malformed|truncated|unsupported"):
        A.decode_flac_pcm(bytes(bad))
    bad = bytearray(good)
    bad[start + 2] ^= 0x10                                               # header byte: CRC-8
    with pytest.raises(A.AudioDecodeError):
        A.decode_flac_pcm(bytes(bad))
    with pytest.raises(A.AudioDecodeError, match="truncated"):
        A.decode_flac_pcm(good[: len(good) // 2])
    bad = bytearray(good)
    bad[8 + 18 + 3] ^= 0xff                                              # STREAMINFO MD5
    A.decode_flac_pcm(bytes(bad))                                        # frames are fine ...
    with pytest.raises(A.AudioDecodeError, match="MD5"):
        A.decode_flac_pcm(bytes(bad), verify_md5=True)                   # ... the signature is not
    with pytest.raises(A.AudioDecodeError):
        A.decode_flac_pcm(b"RIFF" + bytes(100))
    with pytest.raises(A.AudioDecodeError):
        A.decode_audio(good, "mp3")


def test_flac_decoder_on_the_rfc9639_example_streams():
    """Independent vectors (tests/golden/make_flac_rfc9639.py): the three complete streams RFC 9639 prints, written by the reference
    encoder, each carrying that encoder's CRC-8 / CRC-16 / MD5 -- decoded here with the MD5 check on, and compared sample by sample."""
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "flac_rfc9639.npz"))
    expect = {"ex1": (44100, 2, 16), "ex2": (44100, 2, 16), "ex3": (32000, 1, 8)}
    for name, (rate, ch, bps) in expect.items():
        data = z[name + "_bytes"].tobytes()
        pcm, si = A.decode_flac_pcm(data, verify_md5=True)
        assert (si.sample_rate, si.channels, si.bits_per_sample) == (rate, ch, bps)
        assert any(si.md5) and np.array_equal(pcm.T, z[name + "_pcm"]), name
        wav, sr = A.decode_flac(data, verify_md5=True)
        assert sr == rate and wav.shape == (ch, z[name + "_pcm"].shape[1]) and float(wav.abs().max()) < 1.0
        # the counting pass (streams of unknown length) agrees with STREAMINFO
        assert A._lib().wj_flac_decode(data, len(data), None, 0) == si.total_samples
    # unknown total length (the field zeroed: the MD5 still covers the samples) takes the counting pass
    data = bytearray(z["ex2_bytes"].tobytes())
    data[8 + 13] &= 0xf0
    data[8 + 14:8 + 18] = bytes(4)
    pcm, si = A.decode_flac_pcm(bytes(data), verify_md5=True)
    assert si.total_samples == 0 and np.array_equal(pcm.T, z["ex2_pcm"])
    # an absurd STREAMINFO sample count is refused before any allocation
    data = bytearray(z["ex2_bytes"].tobytes())
    data[8 + 13] |= 0x0f
    with pytest.raises(A.AudioDecodeError, match="exceed"):
        A.decode_flac_pcm(bytes(data))


def test_flac_decoder_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    """csrc/flac_decode.cpp parses untrusted bitstreams: built with -fsanitize=address,undefined (no recovery) and driven by
    tests/flac_fuzz_main.cpp over the RFC streams and encoder outputs of every subframe kind, each through thousands of random
    mutations.  Any out-of-bounds access, signed overflow or bad shift aborts the driver."""
    import shutil
    import subprocess
    cxx = shutil.which("g++")
    if cxx is None:
        pytest.skip("g++ not available")
    here = os.path.dirname(os.path.abspath(__file__))
    src = os.path.join(os.path.dirname(here), "wavjepa_amd", "csrc", "flac_decode.cpp")
    exe = str(tmp_path / "flac_fuzz")
    r = subprocess.run([cxx, "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-o", exe, src,
                        os.path.join(here, "flac_fuzz_main.cpp")], capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in r.stderr and ("cannot find" in r.stderr or "unrecognized" in r.stderr):
        pytest.skip("sanitizer runtime not installed")
    assert r.returncode == 0, r.stderr
    z = np.load(os.path.join(here, "golden", "flac_rfc9639.npz"))
    files = []
    for name in ("ex1", "ex2", "ex3"):
        f = tmp_path / f"{name}.flac"
        f.write_bytes(z[name + "_bytes"].tobytes())
        files.append(str(f))
    pcm2 = tone_pcm(3000, 2, 16, seed=3)
    corpus = [E.encode(pcm2, 16000, 16, blocksize=576, stereo="mid_side", subframes=[SPECS[3], SPECS[8]]),
              E.encode(tone_pcm(2500, 1, 24, seed=4), 48000, 24, blocksize=1024, subframes=SPECS[5]),
              E.encode(tone_pcm(2000, 2, 12, seed=5), 8000, 12, blocksize=256, stereo="left_side", subframes=[SPECS[7], SPECS[9]])]
    for i, data in enumerate(corpus):
        f = tmp_path / f"enc{i}.flac"
        f.write_bytes(data)
        files.append(str(f))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([exe, "2500", "12345"] + files, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "mutations decoded" in r.stdout


def test_shuffle_buffer_holds_raw_samples_and_starts_early(tmp_path):
    """The reference shuffles the raw tar samples BEFORE decode / map (webdataset .shuffle(1000) ahead of .decode): the buffer must hold
    bytes, give its first sample after `initial` reads, grow to `shuffle`, cover the corpus, and unreadable shards must raise."""
    import random
    shards = []
    for sh in range(3):
        path = tmp_path / f"s{sh}.tar"
        with tarfile.open(path, "w") as tf:
            for i in range(40):
                data = E.encode(tone_pcm(64, 1, 16, seed=sh * 100 + i), 16000, 16, blocksize=64)
                info = tarfile.TarInfo(f"clip{sh}_{i:03d}.flac")
                info.size = len(data)
                tf.addfile(info, io.BytesIO(data))
        shards.append(str(path))
    reads = []
    import importlib
    W = importlib.import_module("wavjepa_amd.data_modules.WebAudioDataModule")      # the module (the package re-exports the class under this name)
    real = W.iterate_shard

    def counting(path):
        for smp in real(path):
            reads.append(smp["__key__"])
            yield smp
    W.iterate_shard = counting
    try:
        raw_samples = W.raw_samples
        it = raw_samples(shards, random.Random(0), shuffle=50, initial=10)
        first = next(it)
        assert isinstance(first["flac"], bytes) and len(reads) <= 20          # two reads per sample given while the buffer grows
        seen = {first["__key__"]}
        for _ in range(600):
            seen.add(next(it)["__key__"])
        assert len(seen) == 120                                              # every clip of every shard comes through
    finally:
        W.iterate_shard = real
    empty = tmp_path / "empty.tar"
    with tarfile.open(empty, "w"):
        pass
    with pytest.raises(RuntimeError, match="in a row"), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        next(raw_samples([str(empty)], random.Random(0), shuffle=10, initial=2, max_shard_failures=4))


def test_dataset_functions_match_the_reference():
    """Known answers + the reference's own data_modules/dataset_functions.py outputs (fixture)."""
    from wavjepa_amd.data_modules import dataset_functions as F
    g = torch.Generator().manual_seed(0)
    x = torch.randn(30000, generator=g) * 0.05
    y = F.normalize_audio(x)
    assert abs(20 * float(torch.log10(torch.sqrt(torch.mean(y ** 2)))) + 14.0) < 1e-4
    assert torch.equal(F.normalize_audio(torch.zeros(10)), torch.zeros(10))
    assert F.pre_process(x, 16000).shape == (1, 160000) and float(F.pre_process(x, 16000)[0, 30000:].abs().max()) == 0.0
    assert F.pre_process(torch.randn(200000, generator=g), 16000).shape == (1, 160000)
    assert F.pad_or_truncate(torch.ones(2, 5), 8).shape == (2, 8) and F.pad_or_truncate_batch(torch.ones(3, 2, 9), 4).shape == (3, 2, 4)
    # the reference's own outputs on the same seeded clips (tests/golden/dataset_functions.npz, make_golden.py)
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dataset_functions.npz"))
    g = torch.Generator().manual_seed(77)
    for n in (1000, 160000, 170001):
        w = torch.randn(n, generator=g) * 0.3
        for name, out in (("pre_process", F.pre_process(w, 16000)), ("pre_process_noise", F.pre_process_noise(w)),
                          ("instance_normalize", F.instance_normalize(w))):
            if f"shape_{name}_{n}" in fx.files:
                assert tuple(out.shape) == tuple(fx[f"shape_{name}_{n}"]), (name, n)
            sub = out.numpy()[:, ::97] if name == "pre_process" else out.numpy()[..., ::97]
            assert np.allclose(sub, fx[f"{name}_{n}"], rtol=1e-6, atol=1e-7), (name, n)
    f2 = torch.randn(2, 50, generator=g)
    for tl in (30, 50, 70):
        assert np.array_equal(F.pad_or_truncate(f2, tl).numpy(), fx[f"pad_or_truncate_{tl}"])
        assert np.array_equal(F.pad_or_truncate_batch(f2[None], tl).numpy(), fx[f"pad_or_truncate_batch_{tl}"])


def test_cpu_resampler_vs_oracle():
    from oracle import resample_oracle as RS
    from wavjepa_amd.resample import KAISER_BEST, resample_waveform_cpu
    rng = np.random.default_rng(1)
    x = rng.standard_normal((2, 5003)).astype(np.float32)
    for orig, new in ((44100, 16000), (32000, 16000), (48000, 16000), (8000, 16000)):
        y = resample_waveform_cpu(torch.from_numpy(x), orig, new, resampling_method="sinc_interp_kaiser", **KAISER_BEST)
        ref = RS.resample(x, orig, new)
        # the product evaluates the tap table in float32, as torchaudio does for a float32 waveform (the reference's call sites); the
        # oracle's table is float64.  With 441 input phases (44.1 -> 16 kHz) the float32 tap times carry ~1e-5 of sin's argument at the
        # far taps: measured 1.6e-5 of the output RMS there, <= 4e-6 for the small-ratio pairs
        assert tuple(y.shape) == ref.shape and np.abs(y.numpy() - ref).max() < (4e-5 if orig == 44100 else 2e-5) * np.sqrt((ref ** 2).mean())
    assert resample_waveform_cpu(torch.from_numpy(x), 16000, 16000) is not None


def make_shard(path, clips):
    """clips: list of (key, member bytes by extension)"""
    with tarfile.open(path, "w") as tf:
        for key, members in clips:
            for ext, data in members.items():
                ti = tarfile.TarInfo(f"{key}.{ext}")
                ti.size = len(data)
                tf.addfile(ti, io.BytesIO(data))


def test_web_audio_data_module_end_to_end(tmp_path):
    from wavjepa_amd.data_modules import WebAudioDataModule
    from wavjepa_amd.data_modules.WebAudioDataModule import iterate_shard
    from wavjepa_amd.masking import TimeInverseBlockMasker
    rng = np.random.default_rng(0)
    clips_a, clips_b = [], []
    for i in range(6):
        rate = [16000, 32000, 44100][i % 3]
        pcm = tone_pcm(int(rate * (0.6 + 0.2 * i)), 1 + (i % 2), 16, seed=i, rate=rate)
        flac = E.encode(pcm, rate, 16, blocksize=4096, stereo="mid_side" if pcm.shape[1] == 2 else "independent",
                        subframes=dict(kind="fixed", order=2, porder=2))
        clips_a.append((f"audio/clip{i:03d}", {"flac": flac, "json": b"{}"}))
    clips_a.append(("audio/broken", {"flac": clips_a[0][1]["flac"][:300]}))                 # undecodable member: skipped with a warning
    clips_a.append(("audio/noflac", {"txt": b"no audio here"}))
    silent = E.encode(np.zeros((8000, 1), np.int64), 16000, 16, subframes=dict(kind="constant"))
    clips_b.append(("b/silence", {"flac": silent}))
    (tmp_path / "a").mkdir()
    (tmp_path / "b").mkdir()
    make_shard(tmp_path / "a" / "shard-000.tar", clips_a[:4])
    make_shard(tmp_path / "a" / "shard-001.tar", clips_a[4:])
    make_shard(tmp_path / "b" / "shard-000.tar", clips_b)
    keys = [s["__key__"] for s in iterate_shard(str(tmp_path / "a" / "shard-000.tar"))]
    assert keys == [f"audio/clip{i:03d}" for i in range(4)]
    masker = TimeInverseBlockMasker(4, 0.65, 10, 0.25, 10, 0.1)

    class DM(WebAudioDataModule):
        SHUFFLE, NUM_WORKERS, PREFETCH_FACTOR = 4, 2, 1

    dm = DM(masker, str(tmp_path / "a" / "shard-{000..001}.tar"), None, batch_size=3, nr_samples_per_audio=2, nr_time_points=200, sr=16000, seed=7)
    dm.setup("fit")
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        it = iter(dm.train_dataloader())
        batches = [next(it) for _ in range(4)]
        del it
    assert DM.NUM_WORKERS == 2          # (worker processes: their warnings surface on stderr, not in this process)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        g0 = dm._batches(0, 1)
        for _ in range(4):
            next(g0)
    assert any("broken" in str(w.message) for w in caught)
    for audio, ctx, tgt, vis in batches:
        assert audio.shape == (3, 1, 160000) and audio.dtype == torch.float32 and bool(torch.isfinite(audio).all())
        assert ctx.shape == (3, 2, 200) and tgt.shape == (3, 2, 4, 200) and vis.shape == (3, 2, 4, 200) and ctx.dtype == torch.bool
        for clip in audio[:, 0]:
            n = int((clip != 0).nonzero().max()) + 1                     # the file's own samples (before the zero padding)
            rms_db = 20 * float(torch.log10(torch.sqrt(torch.mean(clip ** 2) * 160000 / 160000)))
            full = 20 * float(torch.log10(torch.sqrt(torch.sum(clip ** 2) / n)))
            assert n < 160000 and abs(full + 14.0) < 0.05, (n, rms_db, full)          # -14 dBFS over the un-padded part
    # same seed / rank -> same stream; another rank -> a different one
    def first(rank):
        d = DM(masker, str(tmp_path / "a"), None, batch_size=2, nr_samples_per_audio=2, nr_time_points=200, seed=3, rank=rank, world_size=2)
        d.NUM_WORKERS = 1
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            g = d._batches(0, 1)
            return [next(g)[0] for _ in range(3)]
    a0, a0b, a1 = first(0), first(0), first(1)
    assert all(torch.equal(x, y) for x, y in zip(a0, a0b))
    assert not all(torch.equal(x, y) for x, y in zip(a0, a1))
    # mixing: weight 0 for the speech-like shards -> only the silent source is drawn
    mix = DM(masker, [str(tmp_path / "a"), str(tmp_path / "b")], [0.0, 1.0], batch_size=2, nr_samples_per_audio=2, nr_time_points=200, seed=1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        audio = next(mix._batches(0, 1))[0]
    assert float(audio.abs().max()) == 0.0
    with pytest.raises(FileNotFoundError):
        DM(masker, str(tmp_path / "nothing-*.tar"), None).setup("fit")
    # a corpus whose shards are readable but in which no clip decodes: the stream raises after MAX_SAMPLE_FAILURES consecutive
    # failures instead of warning for ever while the trainer waits for its first batch
    (tmp_path / "c").mkdir()
    make_shard(tmp_path / "c" / "shard-000.tar", [(f"bad{i}", {"flac": clips_a[0][1]["flac"][:200 + i]}) for i in range(5)])

    class Strict(DM):
        MAX_SAMPLE_FAILURES = 12

    bad = Strict(masker, str(tmp_path / "c"), None, batch_size=2, nr_samples_per_audio=2, nr_time_points=200, seed=1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with pytest.raises(RuntimeError, match="12 samples in a row"):
            next(bad._batches(0, 1))


def test_denoiser_data_module_batches(tmp_path):
    """WebAudioDataModuleDenoiser (reference data_modules/WebAudioDataModuleDenoiser.py): the 7-field batch the Denoiser hook consumes,
    from FLAC clip shards + .npy RIR-set shards + .npy noise shards, with and without RIRs / noise."""
    from wavjepa_amd.data_modules import WebAudioDataModuleDenoiser
    from wavjepa_amd.data_modules.WebAudioDataModuleDenoiser import fade_noise
    rng = np.random.default_rng(3)
    clips = []
    for i in range(5):
        rate = (32000, 16000)[i % 2]
        pcm = tone_pcm(int(rate * 0.5), 1, 16, seed=20 + i, rate=rate)
        clips.append((f"c{i}", {"flac": E.encode(pcm, rate, 16, subframes=dict(kind="fixed", order=1))}))
    make_shard(tmp_path / "audio-000.tar", clips)

    def npy(a):
        b = io.BytesIO()
        np.save(b, a)
        return b.getvalue()
    make_shard(tmp_path / "rir-000.tar", [(f"r{i}", {"npy": npy(rng.standard_normal((3, 2, 400)).astype(np.float32))}) for i in range(4)])
    make_shard(tmp_path / "noise-000.tar", [("short", {"npy": npy(rng.standard_normal(50000).astype(np.float32))}),
                                             ("long", {"npy": npy(rng.standard_normal(400000).astype(np.float32))})])

    class DM(WebAudioDataModuleDenoiser):
        NUM_WORKERS, SHUFFLE = 0, 4

    dm = DM(str(tmp_path / "audio-000.tar"), str(tmp_path / "rir-000.tar"), str(tmp_path / "noise-000.tar"), batch_size=3, with_noise=True,
            with_rir=True, nr_samples_per_audio=2, nr_time_points=200, seed=5)
    it = iter(dm.train_dataloader())
    seen_short = False
    for _ in range(4):
        audio, source_rir, noise, noise_length, noise_start, noise_rirs, snr = next(it)
        assert audio.shape == (3, 320000) and source_rir.shape == (3, 2, 400) and noise.shape == (3, 320000) and noise_rirs.shape == (3, 2, 2, 400)
        assert noise_length.shape == noise_start.shape == snr.shape == (3,)
        assert bool(((snr >= -5) & (snr <= 5)).all())
        for b in range(3):
            n, st = int(noise_length[b]), int(noise_start[b])
            assert n in (50000, 320000) and 0 <= st <= 320000 - n
            assert float(noise[b, :st].abs().max() if st else 0.0) == 0.0 and float(noise[b, st + n:].abs().max() if st + n < 320000 else 0.0) == 0.0
            seen_short |= n == 50000
            assert abs(float(noise[b, st + n - 1])) < 1e-3                  # faded out
    assert seen_short
    plain = DM(str(tmp_path / "audio-000.tar"), "", "", batch_size=2, with_noise=False, with_rir=False, seed=1)
    audio, source_rir, noise, noise_length, noise_start, noise_rirs, snr = next(iter(plain.train_dataloader()))
    assert audio.shape == (2, 320000) and source_rir == [None, None] and noise == [None, None] and noise_rirs == [None, None] and snr == [None, None]
    a = torch.zeros(1000)
    torch.manual_seed(0)
    f = fade_noise(torch.ones(5000), a, 1000)
    assert f.shape == (1000,) and float(f[0]) == 1.0 and float(f[-1]) == 0.0
    f = fade_noise(torch.ones(600), a, 1000)
    assert f.shape == (600,) and float(f[0]) == 0.0 and float(f[-1]) == 0.0 and float(f[300]) == 1.0
