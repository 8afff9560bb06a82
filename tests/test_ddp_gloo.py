"""world_size-2 CPU test (gloo) of the data-parallel gradient path: the bucketed all-reduce over the flat gradient
buffer averages every element across ranks, buckets tile the buffer, and the parameter broadcast makes ranks equal."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), GLOO_SOCKET_IFNAME="lo")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from tests.test_host_cpu import small_model
        from wavjepa_amd.ddp import FlatGradAllReducer
        from wavjepa_amd.params import FlatParams
        torch.manual_seed(100 + rank)                 # different init per rank on purpose
        m = small_model()
        m._flat = FlatParams(m, torch.device("cpu"))
        m._ensure_engine = lambda: None               # CPU test: no engine, only the flat buffers + collectives
        red = FlatGradAllReducer(m, enc_chunk=2)
        if rank == 1:                                 # state outside the flat buffers (the frozen position tables) is synced too
            m.pos_encoding_encoder.data.add_(1.0)
        red.broadcast_parameters()
        flat = m._flat
        ref = flat.p32.clone()
        dist.broadcast(ref, 0)
        same_params = bool(torch.equal(ref, flat.p32))
        tab = m.pos_encoding_encoder.data.clone()
        dist.broadcast(tab, 0)
        same_params = same_params and bool(torch.equal(tab, m.pos_encoding_encoder.data))
        g = torch.Generator().manual_seed(7 + rank)
        flat.g32.copy_(torch.randn(flat.n, generator=g))
        mine = flat.g32.clone()
        for tag in ("dec", "enc:1", "enc:0", "front"):      # the order the backward emits them
            red.hook(tag)
        red.wait()
        other = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(other, mine)
        want = sum(other) / world
        q.put((rank, same_params, float((flat.g32 - want).abs().max()), len(red.handles)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_bucketed_allreduce_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=100) for _ in range(world)]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    for rank, same, err, pending in res:
        assert same, f"rank {rank}: parameters differ after broadcast"
        assert err < 1e-6, f"rank {rank}: averaged gradient off by {err}"
        assert pending == 0


def _grad_worker(rank, world, port, q):
    """Each rank: its OWN micro-batch, gradients of its own loss (normalised by its LOCAL target count, reference
    jepa.py:359-362) written into the flat gradient buffer at the slots the engine uses, then the bucketed all-reduce."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), GLOO_SOCKET_IFNAME="lo")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        import numpy as np
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        for pth in (root, os.path.join(root, "tests", "golden")):
            sys.path.insert(0, pth)
        import synth
        from oracle import jepa_oracle as J
        from tests.test_host_cpu import SMALL_SPEC, small_model
        from wavjepa_amd.ddp import FlatGradAllReducer
        from wavjepa_amd.params import FlatParams
        torch.manual_seed(5)                          # same init on both ranks (DDP broadcast makes it so anyway)
        m = small_model()
        m._flat = FlatParams(m, torch.device("cpu"))
        m._ensure_engine = lambda: None
        red = FlatGradAllReducer(m, enc_chunk=2)
        red.broadcast_parameters()
        flat = m._flat
        fx = dict(np.load(os.path.join(root, "tests", "golden", "masks.npz")))

        def grads_of(r):
            P = {k: v.detach().clone() for k, v in m.state_dict().items()}
            names = J.trainable_names(P)
            for k in names:
                P[k].requires_grad_(True)
            sl = slice(2 * r, 2 * r + 2)              # rank r's clips: different masks, different target counts
            audio = torch.from_numpy(synth.synth_audio(2, 1, 32159, seed=50 + r))
            out = J.jepa_forward(P, audio, *(torch.from_numpy(fx[k][sl]) for k in ("as_ctx", "as_tgt", "as_vis")), mode="fp32",
                                 spec=SMALL_SPEC, enc_heads=2, dec_heads=2, top_k=2)
            out["loss"].backward()
            return {k: P[k].grad.detach() for k in names}, int(fx["as_tgt"][sl].sum())

        mine, n_tgt = grads_of(rank)
        flat.g32.zero_()
        for k, g in mine.items():
            s = flat.by_name[k]
            flat.g32[s.offset:s.offset + s.numel].copy_(g.reshape(-1))
        for tag in ("dec", "enc:1", "enc:0", "front"):      # 3 encoder layers in chunks of 2: layers {1, 2}, then {0}
            red.hook(tag)
        red.wait()
        both = [grads_of(r) for r in range(world)]
        worst = 0.0
        for k in mine:
            want = sum(b[0][k] for b in both) / world          # equal rank weights, NOT re-weighted by target counts
            s = flat.by_name[k]
            got = flat.g32[s.offset:s.offset + s.numel].view(s.shape)
            worst = max(worst, float((got - want).abs().max() / (want.abs().max() + 1e-30)))
        q.put((rank, worst, [b[1] for b in both]))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_ddp_gradient_equals_mean_of_per_rank_gradients_world2():
    """N ranks x micro-batch == the rank-mean of the per-rank gradients, every rank normalising by its LOCAL
    target count (reference jepa.py:359-362 + Lightning DDP, train.py:174-179); the flat-buffer slots and the bucket hooks
    carry every parameter's gradient."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=280) for _ in range(world)]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    for rank, worst, counts in res:
        assert counts[0] != counts[1], "the two micro-batches should have different target counts"
        assert worst < 1e-5, f"rank {rank}: averaged gradient deviates by {worst}"


def _denoiser_worker(rank, world, port, q):
    """Denoiser stage under data parallelism: a module without an EMA copy (empty teacher buffer) broadcasts its parameters and averages
    its whole flat gradient buffer in one collective (`FlatGradAllReducer.reduce_all`; the stage has no backward section hooks)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), GLOO_SOCKET_IFNAME="lo")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from wavjepa_amd.ddp import FlatGradAllReducer
        from wavjepa_amd.denoiser import Denoiser
        from wavjepa_amd.extractors import ConvFeatureExtractor
        from wavjepa_amd.params import FlatParams
        from wavjepa_amd.types import TransformerEncoderCFG, TransformerLayerCFG
        torch.manual_seed(200 + rank)
        m = Denoiser(ConvFeatureExtractor(conv_layers_spec=[(32, 10, 5)] + [(32, 3, 2)] * 4 + [(32, 2, 2)], in_channels=1),
                     TransformerLayerCFG.create(d_model=64, nhead=2), TransformerEncoderCFG.create(num_layers=2))
        m._flat = FlatParams(m, torch.device("cpu"))
        assert m._flat.tn == 0 and m._flat.owns(m)
        m._ensure_engine = lambda: None
        red = FlatGradAllReducer(m)
        red.broadcast_parameters()
        flat = m._flat
        ref = flat.p32.clone()
        dist.broadcast(ref, 0)
        same = bool(torch.equal(ref, flat.p32))
        flat.g32.copy_(torch.randn(flat.n, generator=torch.Generator().manual_seed(9 + rank)))
        mine = flat.g32.clone()
        red.reduce_all()
        red.wait()
        other = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(other, mine)
        q.put((rank, same, float((flat.g32 - sum(other) / world).abs().max())))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_denoiser_stage_allreduce_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_denoiser_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=100) for _ in range(world)]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    for rank, same, err in res:
        assert same and err < 1e-6, (rank, same, err)
