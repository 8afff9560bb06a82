"""CPU tests of the host side: C-ABI library loads and exports every declared symbol with matching struct layouts,
maskers / positions reproduce the reference fixtures, module + state_dict layout, config loader, conv geometry,
gradient-bucket tiling, and the product path's refusal to run without a GPU (no silent fallback)."""
import ctypes
import os

import numpy as np
import pytest
import torch

import synth

SPEC = [(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512, 2, 2)]
SMALL_SPEC = [(64, 10, 5)] + [(64, 3, 2)] * 4 + [(64, 2, 2)]


def small_model(**kw):
    from wavjepa_amd.extractors import ConvFeatureExtractor
    from wavjepa_amd.jepa import JEPA
    from wavjepa_amd.types import TransformerEncoderCFG, TransformerLayerCFG
    ext = ConvFeatureExtractor(conv_layers_spec=SMALL_SPEC, in_channels=1)
    return JEPA(feature_extractor=ext, transformer_encoder_cfg=TransformerEncoderCFG.create(num_layers=3),
                transformer_encoder_layers_cfg=TransformerLayerCFG.create(d_model=128, nhead=2),
                transformer_decoder_cfg=TransformerEncoderCFG.create(num_layers=2),
                transformer_decoder_layers_cfg=TransformerLayerCFG.create(d_model=64, nhead=2), average_top_k_layers=2,
                process_audio_seconds=2.01, nr_samples_per_audio=2, **kw)


def test_abi_library_exports_every_declared_symbol():
    from wavjepa_amd import _abi
    lib = _abi.load()
    assert lib.wj_abi_version() == _abi.DEFINES["WJ_ABI_VERSION"]
    assert len(_abi.FUNCTIONS) >= 25
    for fn in _abi.FUNCTIONS:
        assert hasattr(lib, fn), fn
    for name, cls in _abi.STRUCTS.items():
        assert lib.wj_struct_size(name.encode()) == ctypes.sizeof(cls), name
    assert lib.wj_struct_size(b"no_such_struct") == -1
    assert lib.wj_device_count() >= 0


def test_abi_rejects_bad_arguments_without_a_gpu():
    """Argument validation happens before any launch, so it can be exercised on a CPU-only host."""
    from wavjepa_amd import _abi
    lib = _abi.load()
    a = _abi.STRUCTS["wj_gemm_args"]()
    assert lib.wj_gemm_bf16(ctypes.byref(a), None) == -1            # null pointers
    a.A = a.B = a.C = 16
    a.M, a.N, a.K, a.lda, a.ldb, a.ldc = 8, 12, 8, 8, 8, 12           # N % 8 != 0
    assert lib.wj_gemm_bf16(ctypes.byref(a), None) == -1
    b = _abi.STRUCTS["wj_attn_fwd_args"]()
    b.qkv = b.out = 16
    b.B, b.T, b.H, b.hd, b.mask_group = 1, 500, 2, 64, 1              # T > 416
    assert lib.wj_attn_fwd(ctypes.byref(b), None) == -1
    b.T, b.hd = 200, 48
    assert lib.wj_attn_fwd(ctypes.byref(b), None) == -3               # unsupported head dim


def test_rccl_bucket_family_validates_arguments_and_order_without_a_gpu():
    """SURVEY 8(b) rccl_bucket_allreduce_{init,launch,wait}: argument and ordering errors are answered before RCCL is touched."""
    from wavjepa_amd import _abi
    lib = _abi.load()
    assert lib.wj_rccl_unique_id(None) == -1
    assert lib.wj_rccl_bucket_allreduce_init(None) == -1
    ini = _abi.STRUCTS["wj_rccl_init_args"]()
    assert lib.wj_rccl_bucket_allreduce_init(ctypes.addressof(ini)) == -1          # no id, world 0
    idbuf = ctypes.create_string_buffer(128)
    ini.unique_id, ini.rank, ini.world = ctypes.cast(idbuf, ctypes.c_void_p).value, 2, 2
    assert lib.wj_rccl_bucket_allreduce_init(ctypes.addressof(ini)) == -1          # rank outside the world
    la = _abi.STRUCTS["wj_rccl_launch_args"]()
    assert lib.wj_rccl_bucket_allreduce_launch(ctypes.byref(la), None) == -1       # null bucket
    la.buf, la.count, la.average = 256, 1024, 1
    assert lib.wj_rccl_bucket_allreduce_launch(ctypes.byref(la), None) == -3       # no communicator yet: refused, nothing launched
    assert lib.wj_rccl_bucket_allreduce_wait(None, None) == -1
    assert lib.wj_rccl_bucket_allreduce_finalize() == 0                            # idempotent
    from wavjepa_amd import ops
    with pytest.raises(ValueError):
        ops.rccl_bucket_allreduce_init(b"short", 0, 1)


def test_product_path_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    m = small_model()
    audio = torch.zeros(1, 1, 32159)
    ctx = torch.zeros(1, 200, dtype=torch.bool)
    tg = torch.zeros(1, 4, 200, dtype=torch.bool)
    with pytest.raises(RuntimeError):
        m(audio, ctx, tg, tg)
    with pytest.raises(RuntimeError):
        m.get_audio_representation(audio, None)
    with pytest.raises(RuntimeError):
        m.extract_audio(audio)
    from wavjepa_amd import scene
    with pytest.raises(RuntimeError):
        scene.convolve_with_rir(torch.zeros(2, 100), torch.zeros(2, 1, 10))
    with pytest.raises(RuntimeError):
        scene.add_noise(torch.zeros(2, 1, 100), torch.zeros(2, 1, 100), torch.zeros(2), torch.zeros(2), torch.zeros(2))
    from wavjepa_amd.denoiser import Denoiser
    from wavjepa_amd.extractors import ConvFeatureExtractor
    from wavjepa_amd.resample import resample
    from wavjepa_amd.types import TransformerEncoderCFG, TransformerLayerCFG
    with pytest.raises(RuntimeError):
        resample(torch.zeros(1, 1, 1000), 16000)
    den = Denoiser(ConvFeatureExtractor(conv_layers_spec=[(32, 10, 5)] + [(32, 3, 2)] * 4 + [(32, 2, 2)], in_channels=1),
                   TransformerLayerCFG.create(d_model=64, nhead=2), TransformerEncoderCFG.create(num_layers=1))
    den._set_teacher(m)
    with pytest.raises(RuntimeError):
        den(torch.zeros(1, 1, den.target_length), torch.zeros(1, 1, den.target_length))


class Pinned:
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


def test_maskers_bit_exact_with_reference_fixtures(golden_dir):
    from wavjepa_amd.masking import SpeechMasker, TimeInverseBlockMasker
    fx = dict(np.load(os.path.join(golden_dir, "masks.npz")))
    with Pinned(int(fx["as_base"])):
        c, t, v = TimeInverseBlockMasker(4, 0.65, 10, 0.25, 10, 0.1)(batch_size=8, n_times=200, in_channels=1)
    assert np.array_equal(c.numpy(), fx["as_ctx"]) and np.array_equal(t.numpy(), fx["as_tgt"]) and np.array_equal(v.numpy(), fx["as_vis"])
    with Pinned(int(fx["ls_base"])):
        c, t, v = SpeechMasker(4, 0.1, 10, 0.5, 5)(batch_size=8, n_times=200, in_channels=1)
    assert np.array_equal(c.numpy(), fx["ls_ctx"]) and np.array_equal(t.numpy(), fx["ls_tgt"]) and np.array_equal(v.numpy(), fx["ls_vis"])
    with Pinned(int(fx["as400_base"])):
        c, t, v = TimeInverseBlockMasker(4, 0.65, 10, 0.25, 10, 0.1)(batch_size=4, n_times=400, in_channels=1)
    assert np.array_equal(c.numpy(), fx["as400_ctx"]) and np.array_equal(t.numpy(), fx["as400_tgt"])


def test_masker_statistics():
    from wavjepa_amd.masking import TimeInverseBlockMasker
    c, t, v = TimeInverseBlockMasker(4, 0.65, 10, 0.25, 10, 0.1)(batch_size=64, n_times=200, in_channels=1)
    ctx = (~c).sum(-1).float()
    assert c.dtype == torch.bool and t.shape == (64, 4, 200) and v.shape == (64, 4, 200)
    assert 25 < float(ctx.mean()) < 55 and float(ctx.min()) >= 20          # survey: mean 38.7, cutoff 0.1 * 200
    assert 38 < float(t.sum(-1).float().mean()) < 50                       # survey: 45.6 targets per group
    assert not ((~c)[:, None] & t).any()


def test_positions_and_schedules_match_reference(golden_dir):
    from wavjepa_amd.jepa import cosine_schedule_with_warmup
    from wavjepa_amd.pos_embed import get_1d_sincos_pos_embed_from_grid
    fx = dict(np.load(os.path.join(golden_dir, "misc.npz")))
    for d in (768, 384, 64):
        tab = torch.from_numpy(get_1d_sincos_pos_embed_from_grid(d, np.arange(200, dtype=np.float64))).float().numpy()
        assert np.array_equal(tab[::13, ::17], fx[f"pos{d}_slice"]) and np.array_equal(tab[199], fx[f"pos{d}_row199"])
    m = small_model()
    for s, d in zip(fx["ema_steps"], fx["ema_decay"]):
        m.global_step = int(s)
        assert abs(m._get_ema_decay() - d) < 1e-15
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
    sch = cosine_schedule_with_warmup(opt, 100000, 375000)
    for s, l in zip(fx["lr_steps"], fx["lr_lambda"]):
        assert abs(sch.lr_lambdas[0](int(s)) - l) < 1e-15
    ext = m.extract_audio
    for L, n in zip(fx["patch_lens"], fx["patch_counts"]):
        assert ext.total_patches(int(L)) == int(n)
    assert m.target_length == 32159 and m.total_patches == 200


def test_state_dict_layout_matches_reference():
    from wavjepa_amd.extractors import ConvFeatureExtractor
    from wavjepa_amd.jepa import JEPA
    from wavjepa_amd.types import TransformerEncoderCFG, TransformerLayerCFG
    ext = ConvFeatureExtractor(conv_layers_spec=SPEC, in_channels=1)
    with torch.device("meta"):
        pass
    m = JEPA(feature_extractor=ext, transformer_encoder_cfg=TransformerEncoderCFG.create(), transformer_encoder_layers_cfg=TransformerLayerCFG.create(),
             transformer_decoder_cfg=TransformerEncoderCFG.create(), transformer_decoder_layers_cfg=TransformerLayerCFG.create(d_model=384),
             average_top_k_layers=8, process_audio_seconds=2.01, nr_samples_per_audio=8)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    ref = synth.jepa_shapes(conv_spec=SPEC, in_channels=1, d_enc=768, enc_layers=12, d_dec=384, dec_layers=12, n_tokens=200)
    assert shapes == ref and len(shapes) == 457
    assert sum(p.numel() for p in m.parameters() if p.requires_grad) == 111012864          # SURVEY measured values
    assert sum(v.numel() for v in m.state_dict().values()) == 196299264
    # reference init quirks: teacher == student; in_proj identical across layers, Linear weights re-drawn per layer
    sd = m.state_dict()
    assert torch.equal(sd["encoder.layers.3.linear1.weight"], sd["teacher_encoder.layers.3.linear1.weight"])
    assert torch.equal(sd["encoder.layers.0.self_attn.in_proj_weight"], sd["encoder.layers.5.self_attn.in_proj_weight"])
    assert not torch.equal(sd["encoder.layers.0.linear1.weight"], sd["encoder.layers.5.linear1.weight"])
    assert float(sd["encoder.layers.0.linear1.weight"].abs().max()) <= 2.0 and abs(float(sd["encoder.layers.0.linear1.weight"].std()) - 0.02) < 2e-3
    assert all(not p.requires_grad for p in m.teacher_encoder.parameters())


def test_flat_params_and_gradient_bucket_tiling():
    from wavjepa_amd.ddp import section_ranges
    from wavjepa_amd.params import FlatParams
    m = small_model()
    before = {k: v.clone() for k, v in m.state_dict().items()}
    flat = FlatParams(m, torch.device("cpu"))
    assert flat.owns(m)
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k]), k                      # re-pointing to flat storage keeps every value
    assert flat.n % 8 == 0 and all(s.offset % 8 == 0 for s in flat.slots)
    # parameters alias the flat buffers: a write through the buffer is visible in the module and vice versa
    flat.p32.zero_()
    assert float(m.encoder.layers[1].linear2.weight.abs().sum()) == 0.0
    m.mask_token.data.fill_(3.0)
    s = flat.by_name["mask_token"]
    assert float(flat.p32[s.offset]) == 3.0
    ranges = section_ranges(flat, enc_layers=3, enc_chunk=2)
    assert set(ranges) == {"dec", "enc:1", "enc:0", "front"}
    total = sum(hi - lo for v in ranges.values() for lo, hi in v)
    assert total == flat.n
    # student-encoder slice mirrors the teacher buffer (single-kernel EMA)
    assert flat.enc_numel == flat.tn
    # the optimiser's split (FusedAdamW.overlap_next_forward): what the front-end reads before the first transformer kernel / the rest;
    # together exactly [0, n), the encoder slice (what the EMA reads) entirely in `rest`
    front, rest = flat.front_and_rest_ranges()
    runs = sorted(front + rest)
    assert runs[0][0] == 0 and runs[-1][1] == flat.n and all(a[1] == b[0] for a, b in zip(runs, runs[1:]))
    assert all(lo % 8 == 0 and hi % 8 == 0 for lo, hi in runs)
    inside = lambda off, rs: any(lo <= off < hi for lo, hi in rs)
    for sl in flat.slots:
        is_front = sl.name.startswith(("mask_token", "extract_audio.", "feature_norms.", "post_extraction_mapper."))
        assert inside(sl.offset, front) == is_front and inside(sl.offset, rest) == (not is_front), sl.name
    assert inside(flat.enc_offset, rest) and inside(flat.enc_offset + flat.enc_numel - 1, rest)


def test_size_large_layout_and_gradient_buckets_at_24_layers(golden_dir):
    """size="large" (reference jepa.py:114-118): the constructor is handed the BASE layer configs and widens the student to ViT-Large.  Built on
    the meta device (600 M parameters: no storage needed for a layout check): state_dict names / shapes against the reference's own large model
    (tests/golden/large_forward.npz, written by make_golden.py `large`), the flat-parameter layout, and the all-reduce buckets of
    ddp.section_ranges tiling the flat gradient buffer at 24 encoder layers (8 chunks of 3)."""
    import types
    from wavjepa_amd.ddp import section_ranges
    from wavjepa_amd.extractors import ConvFeatureExtractor
    from wavjepa_amd.jepa import JEPA
    from wavjepa_amd.params import _layout
    from wavjepa_amd.types import TransformerEncoderCFG, TransformerLayerCFG
    with torch.device("meta"):
        ext = ConvFeatureExtractor(conv_layers_spec=SPEC, in_channels=1)
        m = JEPA(feature_extractor=ext, transformer_encoder_cfg=TransformerEncoderCFG.create(), transformer_encoder_layers_cfg=TransformerLayerCFG.create(),
                 transformer_decoder_cfg=TransformerEncoderCFG.create(), transformer_decoder_layers_cfg=TransformerLayerCFG.create(d_model=384),
                 average_top_k_layers=8, process_audio_seconds=2.01, nr_samples_per_audio=8, size="large")
    assert (m.encoder_embedding_dim, m.n_encoder_heads, m.encoder.num_layers, m.decoder_embedding_dim, m.decoder.num_layers) == (1024, 16, 24, 384, 12)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert shapes == synth.jepa_shapes(conv_spec=SPEC, in_channels=1, d_enc=1024, enc_layers=24, d_dec=384, dec_layers=12, n_tokens=200)
    fx = np.load(os.path.join(golden_dir, "large_forward.npz"))
    assert len(shapes) == int(fx["n_tensors"]) and sum(int(np.prod(s)) for s in shapes.values()) == int(fx["n_params_total"])
    assert sum(p.numel() for p in m.parameters() if p.requires_grad) == int(fx["n_params_trainable"])
    slots, n = _layout([(k, p) for k, p in m.named_parameters() if p.requires_grad])
    flat = types.SimpleNamespace(slots=slots, n=n)
    ranges = section_ranges(flat, enc_layers=24, enc_chunk=3)
    assert set(ranges) == {"dec", "front"} | {f"enc:{i}" for i in range(0, 24, 3)}
    assert sum(hi - lo for v in ranges.values() for lo, hi in v) == n
    by = {s.name: s for s in slots}
    lo, hi = ranges["enc:21"][0]            # the top chunk carries layers 21-23 and the final norm
    for name in ("encoder.layers.21.linear1.weight", "encoder.layers.23.linear2.bias", "encoder.norm.weight"):
        assert lo <= by[name].offset < hi, name
    assert not lo <= by["encoder.layers.20.linear2.bias"].offset < hi


def test_gemm_release_stream_is_a_host_call():
    """wj_gemm_release_stream: releasing a stream the library has never seen is a no-op that succeeds (no GPU needed: the counter-set
    table is host state)."""
    from wavjepa_amd import ops
    ops.gemm_release_stream(0)
    ops.gemm_release_stream(0x1234)


def test_workspace_query():
    """wj_workspace_bytes: the caller sizes every scratch buffer from the library (no compute, runs without a GPU)."""
    from wavjepa_amd import ops
    assert ops.workspace_bytes("wj_layernorm_bwd", D=768) == 1536 * 3 * 768 * 4
    assert ops.workspace_bytes("wj_attn_bwd", B=1024, H=12, hd=32) == 1024 * 3 * 384 * 4
    # conv0: folded sums + one partial record per (clip, chunk) -- stored and folded in order, no float atomics
    # (round 5: the chunk count follows the clip count -- about one round of the chip's 768 resident workgroups: 3 chunks at 256 clips,
    # 12 at 64 clips; round 4: always ceil(L_out / 1024) = 7)
    assert ops.workspace_bytes("wj_conv0_gn_gelu_fwd", N=256, C=512, C_in=1, k=10, L_out=6430) == (2 * 256 * 512 + 256 * 3 * (512 * 12 + 10)) * 4
    assert ops.workspace_bytes("wj_conv0_gn_gelu_fwd", N=64, C=512, C_in=1, k=10, L_out=6430) == (2 * 64 * 512 + 64 * 12 * (512 * 12 + 10)) * 4
    assert ops.workspace_bytes("wj_conv0_gn_gelu_bwd", N=256, C=512, C_in=1, k=10, L_out=6430) == 256 * (1 + 26) * 512 * 12 * 4
    assert ops.workspace_bytes("wj_conv0_gn_gelu_bwd", N=256, C=512, C_in=1, k=10, L_out=6430, max_rows=1300) == 256 * (1 + 6) * 512 * 12 * 4
    assert ops.workspace_bytes("wj_masked_mse", B=256, G=4, T=200) == (2 + 204800) * 4
    from wavjepa_amd import _abi
    lib = _abi.load()
    assert lib.wj_workspace_bytes(b"nope", b"x") == -1
    # wj_gemm_bf16 (round 5): scratch for the K-split pairs -- only for row-form WJ_EPI_BF16 problems of 33..128 output tiles with
    # N % 256 == 0, K % 256 == 0, K >= 1536: [tiles][2] flags padded to 4 KiB + two roles x four waves' accumulators per tile
    gemm = dict(lda=3072, ldb=3072, ldc=768, epilogue=ops.EPI_BF16)
    assert ops.workspace_bytes("wj_gemm_bf16", M=9945, N=768, K=3072, **gemm) == 4096 + 117 * 2 * 4 * 32 * 64 * 16
    assert ops.workspace_bytes("wj_gemm_bf16", M=9945, N=768, K=768, **gemm) == 0            # short K: the exchange costs more than it saves
    assert ops.workspace_bytes("wj_gemm_bf16", M=51200, N=768, K=3072, **gemm) == 0          # 600 tiles: the persistent schedule
    assert ops.workspace_bytes("wj_gemm_bf16", M=9945, N=768, K=3072, **dict(gemm, epilogue=ops.EPI_ADD_F32)) == 0
    assert ops.workspace_bytes("wj_gemm_bf16", M=9945, N=768, K=3072, b_trans=1, **gemm) == 0
    assert ops.workspace_bytes("wj_mask_scatter_fill_pos_bwd", B=256, T=200, D=384, G=4) == ops.scatter_fill_bwd_partial_rows(256, 200) * 384 * 4


def test_engine_pair_rule_matches_the_library():
    """engine._pair_pays (which dgrads take the row form because the library will run them as K-split pairs) must agree with the
    library's own eligibility rule (wj_workspace_bytes > 0) on every shape of a grid -- two copies of one rule, kept honest."""
    from wavjepa_amd import ops
    from wavjepa_amd.engine import JepaEngine

    class Probe:
        pair_split = True
        PAIR_TILES = JepaEngine.PAIR_TILES
    for M in (900, 4000, 8300, 9945, 12000, 16384, 32768, 33000):
        for N in (256, 384, 512, 768, 1024):
            for K in (256, 768, 1280, 1536, 2304, 3072):
                want = ops.workspace_bytes("wj_gemm_bf16", M=M, N=N, K=K, lda=K, ldb=K, ldc=N, epilogue=ops.EPI_BF16) > 0
                assert JepaEngine._pair_pays(Probe, M, N, K) == want, (M, N, K)


def test_conv_geometry_and_mask_plan():
    from wavjepa_amd.engine import conv_geometry, make_mask_plan
    L, P = conv_geometry(32159, SPEC)
    assert L == [6430, 3214, 1606, 802, 400, 200] and P == [6432, 3216, 1608, 804, 402, 201]
    L2, P2 = conv_geometry(64160, SPEC)
    assert L2[-1] == 400 and all(P2[i] == 2 * P2[i + 1] for i in range(5)) and all(p > l for p, l in zip(P2, L2))
    w2v = SPEC + [(512, 2, 2)]
    L3, P3 = conv_geometry(64320, w2v)
    assert L3[-1] == 200 and all(P3[i] == 2 * P3[i + 1] for i in range(6))
    ctx = torch.tensor([[True, False, False, True], [False, True, True, False]])
    tgt = torch.zeros(2, 2, 4, dtype=torch.bool)
    plan = make_mask_plan(ctx, tgt, tgt, torch.device("cpu"))
    assert plan.n_ctx == 4 and plan.keep.tolist() == [1, 2, 4, 7] and plan.inv.tolist() == [-1, 0, 1, -1, 2, -1, -1, 3]
    assert plan.vis_u8.shape == (4, 4)
    # ragged index lists: context rows per clip; visible predictor rows per (clip, group), packed in (b, g, t) order
    tgt[0, 0, 0] = tgt[0, 1, 3] = tgt[1, 1, 1] = True
    vis = ~((~ctx)[:, None] | tgt)
    plan = make_mask_plan(ctx, tgt, vis, torch.device("cpu"))
    assert plan.ragged_ok and plan.enc_off.tolist() == [0, 2, 4] and plan.max_enc == 2
    assert plan.dec_rows.tolist() == [0, 1, 2, 5, 6, 7, 8, 11, 12, 13, 15] and plan.n_dec == 11
    assert plan.dec_off.tolist() == [0, 3, 6, 8, 11] and plan.max_dec == 3
    assert plan.dec_map.tolist() == [0, 1, 2, -1, -1, 3, 4, 5, 6, -1, -1, 7, 8, 9, -1, 10]
    # target rows among the packed predictor rows (the last predictor layer keeps only these after its attention)
    assert plan.n_tgt == 3 and plan.tgt_rows.tolist() == [0, 5, 9] and plan.tgt_dense.tolist() == [0, 7, 13]
    assert plan.tgt_inv.tolist() == [0, -1, -1, -1, -1, 1, -1, -1, -1, 2, -1]
    vis[0, 0, 0] = True                                            # a key-masked target: not expressible in ragged form
    assert not make_mask_plan(ctx, tgt, vis, torch.device("cpu")).ragged_ok


def test_conv_active_rows_against_brute_force():
    """Row lists of the sparse conv backward: every row that can carry gradient is listed, the dgrad row list is the halo-grown
    set, and the dgrad outputs of layer l are exactly the listed rows of layer l-1 (interval arithmetic vs dense propagation)."""
    from wavjepa_amd.engine import conv_active_rows, conv_geometry
    L, P = conv_geometry(32159, SPEC)
    rng = np.random.default_rng(5)
    N = 6
    keep = np.zeros((N, L[-1]), bool)
    for n in range(N):
        for st in rng.integers(0, L[-1] - 12, size=rng.integers(1, 7)):
            keep[n, st:st + rng.integers(1, 12)] = True
    keep[0] = False                                   # a clip without any context row
    keep[1, :] = True                                 # ... and one that is all context
    keep[2, -1] = True                                # the last token of a clip
    out = conv_active_rows(keep, P, SPEC)
    cur = np.zeros((N, P[-1]), bool)
    cur[:, :L[-1]] = keep
    for l in range(len(SPEC) - 1, 0, -1):
        _, k, s = SPEC[l]
        act, ext = out[l]
        assert np.all(np.diff(act) > 0) and np.all(np.diff(ext) > 0)
        assert np.isin(np.flatnonzero(cur.reshape(-1)), act).all()          # superset of the truly active rows
        a2 = np.zeros(N * P[l], bool)
        a2[act] = True
        a2 = a2.reshape(N, P[l])
        grown = a2.copy()
        for h in range(1, -(-k // s)):
            grown[:, h:] |= a2[:, :-h]
        assert np.array_equal(np.flatnonzero(grown.reshape(-1)), ext)
        written = np.zeros((N, P[l - 1]), bool)
        e = np.flatnonzero(grown.reshape(-1))
        for rho in range(s):
            written[e // P[l], (e % P[l]) * s + rho] = True
        assert np.array_equal(np.flatnonzero(written.reshape(-1)), out[l - 1][0])
        assert int(grown[:, L[l]:].sum()) <= N * (-(-k // s) - 1)             # the halo stays inside the clip's padding rows
        nxt = np.zeros((N, P[l - 1]), bool)
        idx = np.nonzero(cur)
        for kk in range(k):
            nxt[idx[0], idx[1] * s + kk] = True
        cur = nxt
    act0, off0 = out[0]
    assert off0[0] == 0 and off0[-1] == act0.size and np.all(np.diff(off0) >= 0)
    assert np.array_equal(np.bincount(act0 // P[0], minlength=N), np.diff(off0))
    assert off0[1] == 0                                # the clip without context has no active rows anywhere


def test_config_loader_and_factories():
    import train
    from wavjepa_amd.config import load_config, parse_conv_spec
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs")
    cfg = load_config(root, ["masker=LibriSpeech", "trainer.steps=100", "optimizer.lr=1e-3", "extractor=wav2vec2"])
    assert cfg.masker.name == "speech-masker" and cfg.trainer.steps == 100 and cfg.optimizer.lr == 1e-3
    assert len(parse_conv_spec(cfg.extractor.conv_layers_spec)) == 7
    with pytest.raises(ValueError):
        parse_conv_spec("__import__('os').system('true')")
    cfg = load_config(root)
    mk = train.ComponentFactory.create_masker(cfg)                 # YAML spelling context_mask_prob is accepted
    assert mk.context_mask_prob == 0.65 and mk.context_mask_length == 10
    cfg.masker["context_prob"] = 0.5                               # ... and so is the spelling train.py reads upstream
    assert train.ComponentFactory.create_masker(cfg).context_mask_prob == 0.5
    cfg.extractor["name"] = "nope"
    with pytest.raises(ValueError):
        train.ComponentFactory.create_extractor(cfg)
    model, patches = train.build_model(load_config(root))
    assert patches == 200 and model.hparams.lr == 0.0004 and model.hparams.adam_weight_decay == 0.04


def test_hear_runtime_padding_arithmetic():
    from hear_api.runtime import calculate_padding_mask, get_timestamps, strip_compile_prefixes

    class M:
        device = torch.device("cpu")

    for n in (16000, 32159, 50000, 160000):
        unit = 32159
        pad = unit - (n % unit)
        total = n + pad
        mask, cut = calculate_padding_mask(pad, total, 16000, 200, 32159 // 16000, M(), 2)
        assert mask.shape[0] == 2 and mask.shape[1] == 200 * (total // unit)
        assert int(mask[0].sum()) == mask.shape[1] - cut and (cut == mask.shape[1] or bool(mask[0, cut]))
    ts = get_timestamps(16000, 2, 32000, torch.zeros(2, 100, 8))
    assert ts.shape == (2, 100) and abs(float(ts[0, 1]) - 20.0) < 1e-6
    sd = strip_compile_prefixes({"encoder._orig_mod.layers.0.linear1.weight": 1, "mask_token": 2})
    assert set(sd) == {"encoder.layers.0.linear1.weight", "mask_token"}


def test_run_identity_strings_equal_the_reference_output(golden_dir):
    """utils.get_identity_from_cfg(_denoise) (reference utils.py:1-43): the checkpoint directory of a configuration.  Expected strings
    = what the reference's own functions returned for these configurations (tests/golden/hear_helpers.npz, make_golden.py)."""
    import utils
    from wavjepa_amd.config import load_config
    fx = np.load(os.path.join(golden_dir, "hear_helpers.npz"))
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs")
    cfg = load_config(root, [])
    assert utils.get_identity_from_cfg(cfg) == str(fx["identity_base"])
    assert utils.get_identity_from_cfg(load_config(root, ["masker=LibriSpeech", "trainer.batch_size=16"])) == str(fx["identity_librispeech_bs16"])
    assert utils.get_identity_from_cfg_denoise(load_config(root, [], config_name="denoise")) == str(fx["identity_denoise"])
    assert utils.checkpoint_dir(cfg, "saved_models_jepa_new_masking", utils.get_identity_from_cfg(cfg)).endswith(
        "/saved_models_jepa_new_masking/Data=Synthetic/Extractor=wavjepa/InSeconds=2.01/BatchSize=32/NrSamples=8/NrGPUs=1/LR=0.0004/"
        "TargetProb=0.25/TargetLen=10/ContextProb=0.65/ContextLen=10/MinContextBlock=1/ContextRatio=0.1")


def test_hear_helpers_equal_the_reference_output(golden_dir):
    """hear_api.runtime.calculate_padding_mask / get_timestamps / normalize against what the reference's own functions returned
    (tests/golden/hear_helpers.npz): 228 clip lengths over the window set-ups in use (2.01 s / 200 steps, 4.02 s / 200 steps, ...),
    including exact multiples of the window and one frame either side."""
    from hear_api.runtime import calculate_padding_mask, get_timestamps, normalize
    fx = np.load(os.path.join(golden_dir, "hear_helpers.npz"))

    class M:
        device = torch.device("cpu")

    assert bool(fx["mask_is_trailing"].all())
    for (unit, steps, ps, n), cut, mlen, mtrue in zip(fx["cases"].tolist(), fx["cut"].tolist(), fx["mask_len"].tolist(), fx["mask_true"].tolist()):
        pad = unit - (n % unit)
        mask, got_cut = calculate_padding_mask(pad, n + pad, 16000, steps, ps, M(), 2)
        assert got_cut == cut and mask.shape == (2, mlen) and int(mask[0].sum()) == mtrue, (unit, steps, ps, n)
        assert bool(mask[0, mlen - mtrue:].all()) and torch.equal(mask[0], mask[1])
    assert np.array_equal(get_timestamps(16000, 2, 50000, torch.zeros(2, 137, 8)).numpy(), fx["ts_50000_137"])
    assert np.allclose(normalize(torch.from_numpy(fx["norm_in"])).numpy(), fx["norm_out"], rtol=0, atol=1e-6)
    # feature_helper.FeatureExtractor._wav2feature (reference feature_helper.py:27-84): -14 dBFS per clip, channel count fixed
    from hear_api.feature_helper import FeatureExtractor
    cases = [k for k in fx.files if k.startswith("feat_") and "_to_" in k]
    assert len(cases) == 11
    for k in cases:
        src, dst = k[5:].split("_to_")
        x = torch.zeros(1, 1, 500) if src == "silent" else torch.from_numpy(fx["feat_in_" + src])
        out = FeatureExtractor(in_channels=int(dst))._wav2feature(x).numpy()
        assert out.shape == fx[k].shape and np.allclose(out, fx[k], rtol=1e-6, atol=1e-7), k
    with pytest.raises(Exception):
        FeatureExtractor(in_channels=4)._wav2feature(torch.from_numpy(fx["feat_in_2"]))       # stereo -> 4 channels: undefined upstream too


def test_public_signatures_match_the_reference(golden_dir):
    """Drop-in surface: every parameter (name, default) of the reference's constructors / entry points on the path exists here
    (tests/golden/signatures_ref.json = inspect.signature of the reference's classes).  Additions, all keyword-only in use:
    JEPA(warmup_steps=), maskers(channel_major=), data modules(seed=, rank=, world_size=)."""
    import importlib
    import inspect
    import json
    ref = json.load(open(os.path.join(golden_dir, "signatures_ref.json")))
    where = {"JEPA": "wavjepa_amd.jepa", "Denoiser": "wavjepa_amd.denoiser", "ConvFeatureExtractor": "wavjepa_amd.extractors",
             "ConvChannelFeatureExtractor": "wavjepa_amd.extractors", "TimeInverseBlockMasker": "wavjepa_amd.masking",
             "SpeechMasker": "wavjepa_amd.masking", "RuntimeJEPA": "hear_api.runtime", "RuntimeNatJEPA": "hear_api.runtime_natjepa",
             "WebAudioDataModule": "wavjepa_amd.data_modules", "WebAudioDataModuleDenoiser": "wavjepa_amd.data_modules"}
    allowed_extra = {"warmup_steps", "channel_major", "seed", "rank", "world_size"}
    assert len(ref) == 15
    for name, params in ref.items():
        cls, fn = name.split(".")
        f = getattr(getattr(importlib.import_module(where[cls]), cls), fn)
        mine = {p.name: p.default for p in inspect.signature(f).parameters.values()}
        names = [p[0] for p in params]
        for pname, kind, default in params:
            if kind in ("VAR_POSITIONAL", "VAR_KEYWORD"):
                continue
            assert pname in mine, (name, pname)
            d = mine[pname]
            if default == "<required>":
                assert d is inspect.Parameter.empty, (name, pname)
            elif default != "<object>":
                assert (list(d) if isinstance(d, tuple) else d) == default, (name, pname, d, default)
        assert {k for k in mine if k not in names} <= allowed_extra, (name, set(mine) - set(names))
        positional = [p for p in names if p in mine]
        assert [k for k in mine if k in positional] == positional, (name, "parameter order")


def test_config_tree_carries_the_reference_schema_and_values(golden_dir):
    """configs/ against the reference's tree (tests/golden/configs_ref.json = its YAML files parsed, cluster paths dropped): every file
    and key exists here with the same value.  Deliberate differences, each listed: the default data group is the synthetic source
    (no corpus in the image), one GPU and no torch.compile by default, `trainer: denoise` in the reference's denoise.yaml names a file
    that does not exist there (`denoise_audioset` here), and `lr: 1e-4` is a YAML string upstream (a float here)."""
    import json
    import yaml
    ref = json.load(open(os.path.join(golden_dir, "configs_ref.json")))
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs")
    allowed = {("base.yaml", "defaults"), ("denoise.yaml", "defaults"), ("trainer/default_trainer.yaml", "num_gpus"),
               ("trainer/default_trainer.yaml", "compile_modules"), ("trainer/denoise_audioset.yaml", "num_gpus"),
               ("trainer/denoise_librispeech.yaml", "num_gpus"), ("optimizer/adamW_denoise.yaml", "lr")}
    seen = set()
    for rel_path, doc in ref.items():
        mine = yaml.safe_load(open(os.path.join(root, rel_path)))
        for k, v in doc.items():
            assert k in mine, (rel_path, k)
            if v == "<site path>":
                assert isinstance(mine[k], str) and mine[k]
            elif (rel_path, k) in allowed:
                seen.add((rel_path, k))
                if k == "lr":
                    assert float(mine[k]) == float(v)
            else:
                assert mine[k] == v, (rel_path, k, mine[k], v)
    assert seen == allowed


def test_hear_config_modules_follow_the_hear_contract():
    """hear_configs/*.py (reference hear_configs/WavJEPA.py:11-43, WavJEPA_w2v2.py:11-45): `load_model` builds the runtime without
    weights, window length and steps per window follow the conv spec (2.01 s -> 200 steps; 7-layer spec on 4.02 s -> 200 steps)."""
    import hear_configs.WavJEPA as base
    import hear_configs.WavJEPA_Nat as nat
    import hear_configs.WavJEPA_w2v2 as w2v2
    for mod, unit, steps, tokens in ((base, int(2.01 * 16000), 200, 200), (w2v2, int(4.02 * 16000), 200, 200), (nat, int(2.01 * 16000), 200, 400)):
        assert callable(mod.get_scene_embeddings) and callable(mod.get_timestamp_embeddings) and mod.SR == 16000      # 32159 / 64319 frames
        rt = mod.load_model()
        assert rt.unit_frames == unit and rt.output_steps == steps and rt.model.total_patches == tokens and rt.sample_rate == 16000
        assert rt.scene_embedding_size == rt.timestamp_embedding_size == 768
    assert type(nat.load_model()).__name__ == "RuntimeNatJEPA" and nat.load_model().in_channels == 2


def test_mask_plan_takes_group_count_from_the_masks_and_validates_shapes():
    """reference jepa.py:402-405: nr_targets = target_indices.shape[1]; a mismatch must raise, never index out of bounds."""
    import pytest
    from wavjepa_amd.engine import make_mask_plan
    from wavjepa_amd.masking import TimeInverseBlockMasker
    for G in (1, 3, 4):
        ctx, tgt, vis = TimeInverseBlockMasker(G, 0.65, 10, 0.25, 10, 0.1)(batch_size=2, n_times=200, in_channels=1)
        plan = make_mask_plan(ctx, tgt, vis, torch.device("cpu"))
        assert (plan.N, plan.G, plan.T) == (2, G, 200) and plan.vis_u8.shape == (2 * G, 200) and plan.ragged_ok
        assert plan.n_ctx == int((~ctx).sum()) and plan.n_dec == int((~vis).sum()) and plan.n_tgt == int(tgt.sum())
    with pytest.raises(ValueError):
        make_mask_plan(ctx[:1], tgt, vis, torch.device("cpu"))
    with pytest.raises(ValueError):
        make_mask_plan(ctx, tgt, vis[:, :2], torch.device("cpu"))
    with pytest.raises(ValueError):
        make_mask_plan(ctx, tgt[:, 0], vis, torch.device("cpu"))


def test_jepa_is_a_lightning_module_when_lightning_is_importable():
    """reference jepa.py:24 is a pl.LightningModule; so is wavjepa_amd.JEPA whenever pytorch_lightning can be imported (checked in
    a fresh interpreter with a stand-in that has the real class's read-only `global_step` / `device` / `hparams` and a `trainer`
    that raises while unattached).  Without Lightning it is an nn.Module driven by wavjepa_amd.trainer.Trainer."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from wavjepa_amd import jepa as J0
    assert not J0.HAS_LIGHTNING and issubclass(J0.JEPA, torch.nn.Module)      # this image ships no Lightning
    code = """
import sys, types
sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)
import lightning_stub
pl = lightning_stub.install()
from wavjepa_amd import jepa as J
from test_host_cpu import small_model
assert J.HAS_LIGHTNING and issubclass(J.JEPA, pl.LightningModule)
m = small_model(lr=3e-4, warmup_steps=7)
assert m.hparams.lr == 3e-4 and m.hparams.warmup_steps == 7 and m.hparams.average_top_k_layers == 2
assert "feature_extractor" not in m.hparams and m.global_step == 0
oc = m.configure_optimizers()                      # no Trainer attached yet: the schedule falls back to the stock horizon
assert set(oc) == {"optimizer", "lr_scheduler"} and oc["lr_scheduler"]["interval"] == "step"
m.trainer = types.SimpleNamespace(max_steps=50, global_step=5)
assert m.global_step == 5 and m._max_steps() == 50
assert abs(m._get_ema_decay() - (0.99999 - (0.99999 - 0.999) * (1 - 5 / 100000))) < 1e-12
assert len(m.state_dict()) == len(J.JEPA.state_dict(m)) > 50
print("LIGHTNING_OK")
""" % (root, os.path.join(root, "tests"), os.path.join(root, "tests", "golden"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "LIGHTNING_OK" in r.stdout, r.stderr[-3000:]


def test_integration_md_ctypes_stub_matches_the_library():
    """The reference-side binding printed in INTEGRATION.md section 2 must describe the struct the built library expects."""
    from wavjepa_amd import _abi

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    stub = text.split("```python")[2].split("```")[0]
    decl = stub[stub.index("class wj_ln_fwd_args"):stub.index("assert _lib.wj_struct_size")]
    ns = {"ctypes": ctypes}
    exec(decl, ns)
    mirror = ns["wj_ln_fwd_args"]
    assert ctypes.sizeof(mirror) == _abi.load().wj_struct_size(b"wj_ln_fwd_args")
    generated = _abi.STRUCTS["wj_ln_fwd_args"]
    assert [n for n, _ in mirror._fields_] == [n for n, _ in generated._fields_]
    assert f"ABI version {_abi.DEFINES['WJ_ABI_VERSION']}" in text


def test_every_environment_switch_is_documented():
    """INTEGRATION.md section 2 holds ONE table of every WJ_* / WAVJEPA_* environment switch the sources read, with its build (release /
    lab).  This test greps the sources: an undocumented getenv fails, a documented switch nobody reads fails, and a switch that csrc/
    reads through plain getenv (i.e. in the release library) must be listed as release -- result-changing diagnostics belong behind
    wj_lab_env_* (compiled to their defaults unless -DWJ_LAB)."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    sect = text[text.index("### Environment switches"):text.index("ABI version 16")]
    rows = [ln for ln in sect.splitlines() if ln.startswith("| `")]
    doc = {}
    for ln in rows:
        cells = [c.strip() for c in ln.strip("|").split("|")]
        for name in re.findall(r"`((?:WJ|WAVJEPA)_[A-Z0-9_]+)`", cells[0]):
            assert name not in doc, f"{name} is listed twice"
            doc[name] = cells[1]
    csrc_release, csrc_lab, py = set(), set(), set()
    for f in glob.glob(os.path.join(root, "wavjepa_amd", "csrc", "*")):
        src = open(f).read()
        src = re.sub(r"//[^\n]*", "", src)                      # (comments mention switches of other files)
        csrc_release |= set(re.findall(r'(?<![_a-z])getenv\("((?:WJ|WAVJEPA)_[A-Z0-9_]+)"\)', src))
        csrc_lab |= set(re.findall(r'wj_lab_env_(?:int|str)\("(WJ_[A-Z0-9_]+)"', src))
    files = glob.glob(os.path.join(root, "wavjepa_amd", "**", "*.py"), recursive=True) + glob.glob(os.path.join(root, "hear_api", "**", "*.py"), recursive=True)
    files += [os.path.join(root, f) for f in ("train.py", "denoise.py", "bench.py", "utils.py")]
    for f in files:
        for ln in open(f):
            if "environ" in ln:
                py |= set(re.findall(r"[\"']((?:WJ|WAVJEPA)_[A-Z0-9_]+)[\"']", ln))
    assert csrc_release and csrc_lab and py
    assert not (csrc_release & csrc_lab), csrc_release & csrc_lab
    read = csrc_release | csrc_lab | py
    assert read - set(doc) == set(), f"undocumented environment switches: {sorted(read - set(doc))}"
    assert set(doc) - read == set(), f"documented but never read: {sorted(set(doc) - read)}"
    for name in csrc_release | py:
        assert doc[name].startswith("release"), (name, doc[name])
    for name in csrc_lab - py:
        assert doc[name].startswith("lab"), (name, doc[name])
    # the diagnostics that change results are lab-only
    for name in ("WJ_PERSIST_DIAG_NOSTORE", "WJ_PERSIST_STAMPS", "WJ_PERSIST_ACTIVE", "WJ_GEMM_VARIANT"):
        assert name in csrc_lab and name not in csrc_release


def test_release_library_exports_compute_and_query_entries_only():
    """The release .so exports exactly what include/wavjepa_hip.h declares (+ wj_workspace_bytes): no diagnostic entry, no process-global
    setter; the laboratory .so adds exactly the entries of include/wavjepa_hip_lab.h."""
    import subprocess
    from wavjepa_amd import _abi

    def exported(path):
        out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
        return {ln.split()[-1] for ln in out.splitlines() if " T wj_" in ln}

    rel = exported(_abi.LIB_PATH)
    assert rel == set(_abi.FUNCTIONS) | {"wj_workspace_bytes"}, (sorted(rel - set(_abi.FUNCTIONS)), sorted(set(_abi.FUNCTIONS) - rel))
    for gone in ("wj_gemm_set_variant", "wj_gemm_set_persist_cus", "wj_debug_persist_stamps", "wj_collective_footprint"):
        assert gone not in rel
    lab = exported(_abi.LAB_LIB_PATH)
    assert lab == rel | set(_abi.LAB_FUNCTIONS) and set(_abi.LAB_FUNCTIONS) == {"wj_debug_persist_stamps", "wj_collective_footprint"}
    assert _abi.load_lab().wj_abi_version() == _abi.load().wj_abi_version() == _abi.DEFINES["WJ_ABI_VERSION"]


def test_compiled_kernels_have_no_mfma_result_read_across_a_branch(tmp_path):
    """hipcc was seen to leave ONE wait state between an MFMA and the first VALU read of its result when a branch lay between them
    (attention forward, round 2: run-dependent softmax sums).  tools/mfma_hazard_scan.py walks the gfx950 assembly of every source
    for that pattern; it must stay clean."""
    import subprocess
    import sys
    from concurrent.futures import ThreadPoolExecutor
    from wavjepa_amd import build as B

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = B._hipcc()

    def scan(name):
        asm = str(tmp_path / (name + ".s"))
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-S", "--cuda-device-only", "-o", asm,
                            os.path.join(root, "wavjepa_amd", "csrc", name + ".hip")], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "mfma_hazard_scan.py"), asm], capture_output=True, text=True)
        return name, r.returncode, r.stdout[-1500:]

    with ThreadPoolExecutor(max_workers=4) as ex:
        for name, rc, out in ex.map(scan, ["attention", "gemm", "gemm_persist", "norm", "conv0", "fp8"]):
            assert rc == 0, (name, out)


def test_persistent_gemm_assembly_keeps_the_pulled_tile_index_register_untouched():
    """csrc/gemm_persist.hip pulls its next tile with a returning atomic from inline asm (hipcc must not see the result register, or
    it waits for it with vmcnt(0) and drains the LDS-DMA ring).  tools/asm_checks.py proves on the gfx950 assembly of THIS build that
    nothing reads, copies or spills that register between the atomic and the ds_write that consumes it, and that no kernel of the file
    touches scratch."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "asm_checks.py")], capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert "pulls checked" in r.stdout
