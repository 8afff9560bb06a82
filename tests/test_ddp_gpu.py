"""Two ranks on ONE GPU (gloo carries the collectives; RCCL refuses two ranks on a device): the real engine backward fires the
section hooks, the bucketed averages run on device tensors while the rest of the backward computes, and the optimiser steps on
the averaged flat buffer.  What the 8-GPU launch relies on, minus the transport: reference train.py:174-179 (Lightning DDP:
parameter broadcast, gradient mean with equal rank weights, every rank normalising by its LOCAL target count)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, golden_dir, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), GLOO_SOCKET_IFNAME="lo")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        here = os.path.dirname(os.path.abspath(__file__))
        sys.path[:0] = [os.path.dirname(here), here]
        import synth
        from test_jepa_gpu import SMALL, build, dev
        from wavjepa_amd.trainer import StepRunner

        def model(shift):
            m, _ = build(SMALL, warmup_steps=2)
            if shift:                      # rank 1 starts elsewhere (student, EMA teacher AND the frozen position tables): the
                with torch.no_grad():      # broadcast has to bring all of it back
                    for p in m.parameters():
                        p.add_(0.01)
            m.trainer.max_steps = 20
            m.hparams["ema_decay"], m.hparams["ema_end_decay"], m.ema_end_step = 0.9, 0.99, 10
            return m

        m = model(rank == 1)
        run = StepRunner(m)                                    # parameter broadcast + section hooks (JEPA: bucketed all-reduces)
        assert run.reducer.active and run.sectioned and m._grads_ready_hook is not None
        flat = m._flat
        ref = flat.p32.clone()
        dist.broadcast(ref, 0)
        same_start = bool(torch.equal(ref, flat.p32))

        fx = dict(np.load(os.path.join(golden_dir, "masks.npz")))

        def batch(i):
            sl = slice(2 * ((2 * i + rank) % 3), 2 * ((2 * i + rank) % 3) + 2)          # ranks see different clips and masks
            audio = torch.from_numpy(synth.synth_audio(2, 1, 32159, seed=300 + 2 * i + rank)).to(torch.bfloat16).to(dev())
            return (audio,) + tuple(torch.from_numpy(fx[k][sl]) for k in ("as_ctx", "as_tgt", "as_vis"))

        # step 0 on a second, un-hooked replica of the broadcast weights: this rank's LOCAL gradient of the same batch
        twin = model(False)
        twin._ensure_engine()
        twin._flat.p32.copy_(flat.p32)
        twin._flat.t32.copy_(flat.t32)
        twin._student_bf16_fresh = twin._teacher_bf16_fresh = False
        twin.global_step = 0
        out = twin.training_step(batch(0), 0)
        out["loss"].backward()
        local = twin._flat.g32.clone()
        n_tgt = int(batch(0)[2].sum())
        both = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(both, local)
        want = sum(both) / world

        # the hooked model: forward, backward (hooks average the buckets as they complete), then look at the flat buffer
        m.global_step = 0
        out = m.training_step(batch(0), 0)
        out["loss"].backward()
        n_handles = len(run.reducer.handles)
        run.reducer.wait()
        torch.cuda.synchronize()
        dev_grad = float((flat.g32 - want).norm() / (want.norm() + 1e-30))
        from wavjepa_amd.ddp import section_ranges
        per = {}
        for tag, rs in section_ranges(flat, m.encoder.num_layers, run.reducer.enc_chunk).items():
            for lo, hi in rs:
                per[f"{tag}[{lo}:{hi}]"] = (float((flat.g32[lo:hi] - want[lo:hi]).norm() / (want[lo:hi].norm() + 1e-30)),
                                            float((flat.g32[lo:hi] - local[lo:hi]).norm() / (local[lo:hi].norm() + 1e-30)))
        print("rank", rank, "per-section (vs mean, vs local):", per, flush=True)
        not_local = float((flat.g32 - local).norm() / (local.norm() + 1e-30))
        run.optimizer.step()
        run.scheduler.step()
        m.global_step = 1
        for i in (1, 2):                                        # two more whole steps through the same path
            out = m.training_step(batch(i), i)
            out["loss"].backward()
            run.reducer.wait()
            run.optimizer.step()
            run.scheduler.step()
            m.global_step = i + 1
        torch.cuda.synchronize()
        sums = torch.stack([flat.p32.double().sum(), flat.p32.double().abs().sum(), flat.t32.double().sum()]).cpu()
        allsums = [torch.empty_like(sums) for _ in range(world)]
        dist.all_gather(allsums, sums)
        q.put((rank, same_start, n_handles, dev_grad, not_local, n_tgt, [s.tolist() for s in allsums], float(out["loss"].detach())))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_on_one_gpu_average_gradients_and_stay_identical(golden_dir):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, golden_dir, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=560) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    counts = {}
    for rank, same_start, n_handles, dev_grad, not_local, n_tgt, allsums, loss in res:
        counts[rank] = n_tgt
        assert same_start, f"rank {rank}: parameters differ from rank 0 after the broadcast"
        assert n_handles >= 3, "the backward should have launched one collective per section (dec, enc:*, front)"
        # two backward passes of the same batch differ by the fp32 summation order of the split-K weight gradients only
        assert dev_grad < 1e-6, f"rank {rank}: averaged gradient deviates from the mean of the local gradients by {dev_grad}"
        assert not_local > 1e-2, "the ranks' local gradients should differ (different clips): the average must not equal the local one"
        assert allsums[0] == allsums[1], f"replicas diverged after three steps: {allsums}"
        assert np.isfinite(loss)
    assert counts[0] != counts[1], "the two ranks should hold different target counts (local normalisation is part of the contract)"


@pytest.mark.timeout(900)
def test_train_py_two_ranks_share_the_gpu_over_gloo(tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 2 train.py trainer.num_gpus=2 ...` end to end (reference train.py:174-179,
    strategy="ddp"): per-rank synthetic sources, Trainer.fit with the bucketed averages, rank 0 alone logs and writes checkpoints."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "train.py"), "trainer.num_gpus=2", "trainer.steps=4", "trainer.batch_size=2",
           "data.samples_per_audio=2", "trainer.log_every_n_steps=2", f"save_dir={tmp_path / 'runs'}"]
    from tests import launch
    rc, out, err = launch.run(cmd, cwd=root, env=dict(os.environ, WJ_DIST_BACKEND="gloo"), timeout=300)
    assert rc == 0, (out[-1500:], err[-4000:])
    steps = [ln for ln in out.splitlines() if ln.startswith("step ")]
    assert [ln.split()[1] for ln in steps] == ["2", "4"], out[-1500:]                              # one logger (rank 0), not two
    assert all(np.isfinite(float(ln.split("loss")[1].split()[0])) for ln in steps)
    assert "Effective Batch Size is: 8" in out                                                     # 2 sources x 2 crops x 2 ranks
    ckpts = list((tmp_path / "runs").rglob("last.ckpt"))
    assert len(ckpts) == 1 and "NrGPUs=2" in str(ckpts[0])



def test_train_py_config1_tiny_model_batch_of_four(tmp_path):
    """BASELINE config 1's shapes through the launcher: `train.py trainer.size=tiny` = 2-layer student d = 128, 2-layer predictor
    d = 64, a batch of 4 synthetic 2 s clips (1 source x 4 crops).  BASELINE quotes that configuration on the CPU as a plumbing check;
    this package has no CPU compute path by design (the product fails loudly without the HIP library and a GPU), so the same launch
    runs on the GPU.  A few optimisation steps with a 2-step warm-up: finite, and the loss moves."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "train.py"), "trainer.size=tiny", "trainer.batch_size=1", "data.samples_per_audio=4",
           "trainer.steps=6", "trainer.warmup_steps=2", "trainer.log_every_n_steps=1", f"save_dir={tmp_path / 'runs'}"]
    from tests import launch
    rc, out, err = launch.run(cmd, cwd=root, timeout=300)
    assert rc == 0, (out[-1500:], err[-4000:])
    assert "Effective Batch Size is: 4" in out
    losses = [float(ln.split("loss")[1].split()[0]) for ln in out.splitlines() if ln.startswith("step ")]
    assert len(losses) == 6 and all(np.isfinite(losses)) and losses[-1] != losses[0], out[-1500:]
    import torch
    ck = torch.load(next((tmp_path / "runs").rglob("last.ckpt")), map_location="cpu", weights_only=False)
    sd = ck["state_dict"]
    assert sd["encoder.layers.1.linear1.weight"].shape == (512, 128) and "encoder.layers.2.linear1.weight" not in sd
    assert sd["decoder.layers.1.self_attn.in_proj_weight"].shape == (192, 64) and sd["post_extraction_mapper.weight"].shape == (128, 512)


@pytest.mark.timeout(900)
def test_train_py_size_large_three_steps(tmp_path):
    """`train.py trainer.size=large` (reference jepa.py:114-118): the ViT-Large student (d = 1024, 16 x 64 heads, 24 layers, feed-forward
    4096) through the launcher for three optimisation steps -- GEMM K = 1024 / 4096, LayerNorm D = 1024, the 24-layer arena, flat-parameter
    layout and checkpoint.  Finite, the loss moves, the checkpoint carries the large layout."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "train.py"), "trainer.size=large", "trainer.batch_size=2", "data.samples_per_audio=4",
           "trainer.steps=3", "trainer.warmup_steps=1", "trainer.log_every_n_steps=1", f"save_dir={tmp_path / 'runs'}"]
    from tests import launch
    rc, out, err = launch.run(cmd, cwd=root, timeout=600)
    assert rc == 0, (out[-1500:], err[-4000:])
    losses = [float(ln.split("loss")[1].split()[0]) for ln in out.splitlines() if ln.startswith("step ")]
    assert len(losses) == 3 and all(np.isfinite(losses)) and losses[-1] != losses[0], out[-1500:]
    import torch
    sd = torch.load(next((tmp_path / "runs").rglob("last.ckpt")), map_location="cpu", weights_only=False)["state_dict"]
    assert sd["encoder.layers.23.linear1.weight"].shape == (4096, 1024) and "encoder.layers.24.linear1.weight" not in sd
    assert sd["teacher_encoder.layers.23.self_attn.in_proj_weight"].shape == (3072, 1024)
    assert sd["decoder.layers.11.linear1.weight"].shape == (1536, 384) and sd["post_extraction_mapper.weight"].shape == (1024, 512)
    assert sd["pos_encoding_encoder"].shape == (1, 200, 1024) and all(torch.isfinite(v).all() for v in sd.values() if v.is_floating_point())


@pytest.mark.timeout(900)
def test_train_py_config4_nat_scenes_two_ranks_over_gloo(tmp_path):
    """BASELINE config 4 through the launcher: `train.py extractor=wavjepa_nat data=nat_synthetic masker=AudioSet_nat` under
    torch.distributed.run with two ranks -- binaural scenes generated on the device inside the step (source RIR + 2 noise RIRs, SNR
    mix), one mono conv stack per ear (ConvChannelFeatureExtractor), 2 x 200 tokens per clip, channel-based masks in the extractor's
    channel-major order, bucketed gradient averages.  The heart-beat lines of both ranks must show every start-up stage."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "train.py"), "extractor=wavjepa_nat", "data=nat_synthetic",
           "masker=AudioSet_nat", "trainer.num_gpus=2", "trainer.steps=3", "trainer.batch_size=2", "data.samples_per_audio=2",
           "data.source_seconds=4.0", "trainer.log_every_n_steps=1", f"save_dir={tmp_path / 'runs'}"]
    from tests import launch
    rc, out, err = launch.run(cmd, cwd=root, env=dict(os.environ, WJ_DIST_BACKEND="gloo"), timeout=300)
    assert rc == 0, (out[-1500:], err[-4000:])
    steps = [ln for ln in out.splitlines() if ln.startswith("step ")]
    assert [ln.split()[1] for ln in steps] == ["1", "2", "3"], out[-1500:]
    assert all(np.isfinite(float(ln.split("loss")[1].split()[0])) for ln in steps)
    ckpts = list((tmp_path / "runs").rglob("last.ckpt"))
    assert len(ckpts) == 1 and "Extractor=wavjepa-nat" in str(ckpts[0]) and "Data=NatSynthetic" in str(ckpts[0])
    import torch
    sd = torch.load(ckpts[0], map_location="cpu", weights_only=False)["state_dict"]
    assert sd["pos_encoding_encoder"].shape[1] == 400                                   # 2 channels x 200 tokens
    assert any(k.startswith("extract_audio.") for k in sd)
    for r in (0, 1):
        for stage in ("process group up", "parameters broadcast from rank 0", "first optimisation step"):
            assert any(f"[rank {r}/2" in ln and stage in ln for ln in err.splitlines()), (r, stage, err[-3000:])
