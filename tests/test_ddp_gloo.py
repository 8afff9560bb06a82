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
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
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
        red.broadcast_parameters()
        flat = m._flat
        ref = flat.p32.clone()
        dist.broadcast(ref, 0)
        same_params = bool(torch.equal(ref, flat.p32))
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
