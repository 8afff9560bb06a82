"""Minimal trainer with the slice of pytorch_lightning.Trainer's behaviour the reference relies on
(reference train.py:160-180,244): one process per GPU, bf16 compute (the kernels' native flow), gradient-norm clipping
at `gradient_clip_val`, per-step LR scheduler, EMA inside training_step, periodic checkpoints with the
Lightning checkpoint keys (`state_dict`, `hyper_parameters`, `global_step`)."""
from __future__ import annotations

import datetime
import os
import sys
import time
from typing import Any, Dict, Iterable, Optional

import torch
import torch.distributed as dist

from .ddp import FlatGradAllReducer


def heartbeat(msg: str) -> None:
    """One line per rank on stderr at every stage of a multi-rank start (rendezvous, broadcast, first collective): when an N-rank
    launch stalls, the last line of each rank says where.  WJ_HEARTBEAT=0 silences it."""
    if os.environ.get("WJ_HEARTBEAT", "1") != "0" and "RANK" in os.environ:
        print(f"[rank {os.environ.get('RANK')}/{os.environ.get('WORLD_SIZE')} pid {os.getpid()} +{time.perf_counter() - _T0:.1f}s] {msg}",
              file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def init_distributed() -> tuple:
    """(rank, local device index, world).  RCCL ("nccl" backend on ROCm) when launched by torch.distributed.run.

    WJ_DIST_BACKEND=gloo is a development aid: RCCL refuses two ranks on one device, gloo does not, so a 1-GPU box can run the
    N-rank launch end to end (ranks then share the visible GPUs round-robin; the compute path is unchanged)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ      # started by torch.distributed.run (any world size)
    backend = os.environ.get("WJ_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local %= max(torch.cuda.device_count(), 1)
    if launched and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        torch.cuda.set_device(local)
        # a collective that does not complete inside the bound aborts the process (non-zero exit) instead of hanging the launch:
        # WJ_DIST_TIMEOUT_S, default 300 s (the first RCCL collective builds its rings / loads kernels: tens of seconds at most)
        timeout = datetime.timedelta(seconds=float(os.environ.get("WJ_DIST_TIMEOUT_S", "300")))
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
        heartbeat(f"rendezvous at {os.environ['MASTER_ADDR']}:{os.environ['MASTER_PORT']} (backend {backend}, device {local})")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local), timeout=timeout)
        else:
            if os.environ["MASTER_ADDR"] in ("127.0.0.1", "localhost"):
                # gloo binds to the interface the HOSTNAME resolves to; container hostnames often do not resolve (or do after a resolver
                # time-out): a one-node run stays on the loopback interface
                os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
            dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=timeout)
        heartbeat("process group up")
    elif torch.cuda.is_available():
        torch.cuda.set_device(local)
    return rank, local, world


DP_PERSIST_CUS = 28     # resident persistent-GEMM workgroups per XCD in a data-parallel run: 4 CUs per XCD (32 in all) stay free


def init_persist_cus(world: int) -> int:
    """A persistent GEMM workgroup owns its CU (the whole register file, 150 KB of LDS): while one is resident on every CU an RCCL
    channel kernel cannot start.  With more than one rank the persistent kernels therefore leave DP_PERSIST_CUS..32 CUs per XCD to the
    gradient all-reduce that runs beside the backward (WJ_PERSIST_CUS set by the user wins).  Returns the value in effect."""
    from . import ops
    if world > 1 and "WJ_PERSIST_CUS" not in os.environ:
        ops.gemm_set_persist_cus(DP_PERSIST_CUS)
    return ops.gemm_set_persist_cus(0)


class StepRunner:
    """One optimisation step in the reference's order."""

    def __init__(self, model, gradient_clip_val: float = 5.0, enc_chunk: int = 3):
        self.model = model
        model._ensure_engine()
        self.reducer = FlatGradAllReducer(model, enc_chunk=enc_chunk)
        self.reducer.broadcast_parameters()
        if self.reducer.active:
            torch.cuda.synchronize()
            heartbeat("parameters broadcast from rank 0")
        init_persist_cus(dist.get_world_size() if dist.is_initialized() else 1)
        self._first_step_done = False
        self.sectioned = hasattr(model, "_grads_ready_hook")      # JEPA: bucketed all-reduces launched from the backward's hooks
        if self.sectioned:
            model._grads_ready_hook = self.reducer.hook if (self.reducer.active or self.reducer.emulate) else None
        oc = model.configure_optimizers()
        self.optimizer = oc["optimizer"]
        self.scheduler = oc["lr_scheduler"]["scheduler"]
        self.optimizer.max_grad_norm = float(gradient_clip_val or 0.0)
        if self.sectioned and hasattr(self.optimizer, "overlap_next_forward"):
            # (JEPA only: its engine's forward knows where to wait.)  This loop reads parameters only through the engine / state_dict, which wait
            self.optimizer.overlap_next_forward = True
            self.optimizer.fuse_zero_grad = True        # (this loop never reads p.grad behind step())

    def step(self, raw_batch, batch_idx: int) -> Dict[str, Any]:
        m = self.model
        batch = m.on_after_batch_transfer(raw_batch, 0)     # crops + normalise + bf16 on the device
        out = m.training_step(batch, batch_idx)             # forward + EMA of the teacher
        out["loss"].backward()                              # engine backward; buckets all-reduce as they complete
        if not self.sectioned:
            self.reducer.reduce_all()
        self.reducer.wait()
        self.optimizer.step()                               # fused global-norm clip + AdamW over the flat buffers
        self.scheduler.step()
        m.global_step += 1
        if not self._first_step_done:
            self._first_step_done = True
            if self.reducer.active:
                torch.cuda.synchronize()
                heartbeat("first optimisation step (all gradient buckets reduced) done")
        return out


class Trainer:
    def __init__(self, accelerator: str = "gpu", max_steps: int = 375000, max_epochs: int = -1, precision: str = "bf16-mixed",
                 devices: int = 1, gradient_clip_val: float = 5.0, gradient_clip_algorithm: str = "norm", strategy: str = "auto",
                 log_every_n_steps: int = 50, default_root_dir: Optional[str] = None, checkpoint_every_n_steps: int = 25000,
                 **unused):
        if accelerator not in ("gpu", "cuda", "auto"):
            raise ValueError("wavjepa_amd trains on MI355X GPUs only (accelerator='gpu'); there is no CPU path")
        if precision not in ("bf16-mixed", "bf16"):
            raise ValueError("the HIP path computes in bf16 with fp32 master weights (precision='bf16-mixed')")
        if gradient_clip_algorithm != "norm":
            raise ValueError("only norm clipping is implemented")
        self.max_steps = int(max_steps)
        self.devices = int(devices)
        self.gradient_clip_val = gradient_clip_val
        self.log_every_n_steps = log_every_n_steps
        self.root = default_root_dir
        self.ckpt_every = checkpoint_every_n_steps
        self.rank, self.local_rank, self.world = init_distributed()
        self.logged: Dict[str, Any] = {}

    def log_dict(self, data: Dict[str, Any], **kw) -> None:
        self.logged = data          # kept on the device: no per-step host sync / scalar all-reduce

    def save_checkpoint(self, model, runner: StepRunner, path: str, versioned: bool = False) -> None:
        """`versioned`: never overwrite (Lightning's enable_version_counter): last.ckpt, last-v1.ckpt, ... -- the run directory is
        the reference's run-identity directory, where an earlier run's (or a reference run's) last.ckpt may already sit."""
        if self.rank != 0:
            return
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        if versioned and os.path.exists(path):
            stem, ext = os.path.splitext(path)
            v = 1
            while os.path.exists(f"{stem}-v{v}{ext}"):
                v += 1
            path = f"{stem}-v{v}{ext}"
        torch.save({"state_dict": {k: v.detach().cpu() for k, v in model.state_dict().items()},
                    "hyper_parameters": dict(model.hparams), "global_step": model.global_step,
                    "optimizer": runner.optimizer.state_dict(), "lr_scheduler": runner.scheduler.state_dict()}, path)

    def fit(self, model, datamodule=None, train_dataloaders: Optional[Iterable] = None, ckpt_path: Optional[str] = None):
        dev = torch.device("cuda", self.local_rank)
        model.to(dev)
        model.trainer = self
        model.train()
        runner = StepRunner(model, self.gradient_clip_val)
        if ckpt_path:
            ck = torch.load(ckpt_path, map_location="cpu", weights_only=False)
            model.load_state_dict(ck["state_dict"])
            model.global_step = int(ck.get("global_step", 0))
            if "optimizer" in ck:
                runner.optimizer.load_state_dict(ck["optimizer"])
            if "lr_scheduler" in ck:
                runner.scheduler.load_state_dict(ck["lr_scheduler"])
        loader = train_dataloaders if train_dataloaders is not None else datamodule.train_dataloader()
        t0 = time.time()
        for batch in loader:
            if model.global_step >= self.max_steps:
                break
            out = runner.step(batch, model.global_step)
            gs = model.global_step
            if self.log_every_n_steps and gs % self.log_every_n_steps == 0:
                lt = out["loss"].detach().float().clone()
                if dist.is_initialized() and self.world > 1:
                    dist.all_reduce(lt, op=dist.ReduceOp.AVG)   # the reference logs with sync_dist=True (jepa.py:328): the rank mean
            if self.log_every_n_steps and gs % self.log_every_n_steps == 0 and self.rank == 0:
                loss = float(lt)                      # the only host sync, every n steps
                dt = time.time() - t0
                ema = f"  ema {model._get_ema_decay():.6f}" if hasattr(model, "_get_ema_decay") else ""
                print(f"step {gs}  loss {loss:.5f}  lr {runner.scheduler.get_last_lr()[0]:.3e}{ema}  "
                      f"{dt / self.log_every_n_steps * 1000:.1f} ms/step", flush=True)
                t0 = time.time()
            if self.root and self.ckpt_every and gs % self.ckpt_every == 0:
                self.save_checkpoint(model, runner, os.path.join(self.root, f"step={gs}.ckpt"), versioned=ckpt_path is None)
        if self.root:
            self.save_checkpoint(model, runner, os.path.join(self.root, "last.ckpt"), versioned=ckpt_path is None)
        eng = getattr(model, "_engine", None)
        if eng is not None and hasattr(eng, "wait_optimizer"):
            # the last step's update of the transformer stacks may still run on the side stream (overlap_next_forward): whatever the caller
            # does with the model next (export, .cpu(), a submodule's state_dict) is ordered behind it on the compute stream -- no host sync
            eng.wait_optimizer()
        if self.rank == 0 and self.log_every_n_steps:
            print(f"done: {model.global_step} steps, peak HBM {torch.cuda.max_memory_allocated(dev) / 2**30:.1f} GiB allocated, "
                  f"{torch.cuda.max_memory_reserved(dev) / 2**30:.1f} GiB reserved", flush=True)
        return runner
