#!/usr/bin/env python3
"""Headline benchmark: WavJEPA pre-training clips/s on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
      N > 1 is launched by the driver as  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

One "step" = the whole optimisation step on every rank's own 256 synthetic clips (32 white-noise 10 s sources x 8
random 2.01 s crops @16 kHz): crop + normalise, conv encoder, student ViT, predictor, EMA-teacher targets, masked MSE,
EMA update, full backward, gradient all-reduce (N > 1), global-norm clip, AdamW.  WavJEPA-base (12 x d768 student and
teacher, 12 x d384 predictor, 512-ch conv stack), bf16 compute with fp32 master weights, random-init weights.

Prints ONE JSON line (rank 0).  Besides the contract fields it carries
  roofline     the dominant kernel of the step (the bf16 MFMA GEMM family), ALGORITHMIC flops / measured duration, both
               taken live from an instrumented step of this very run (HIP events on the launch stream around every
               C-ABI call), against the dense bf16 MFMA peak of MI355X (2.5 PFLOP/s);
  cpu_baseline the CPU oracle (a port of the reference's algorithm, `oracle/`) timed on this host's cores on a bounded
               sample (N=4 clips; base and tiny models; fp32 and the bf16-autocast flow; thread count picked by a 2-step probe;
               median of 5 timed steps; 60 s budget) -- a reported baseline, not the target;
  calibration  two fixed launches from the library (an MFMA-bound GEMM, an HBM-bound stream) before the warm-up and after the timed
               region: how fast THIS box is, the committed reference box's figures, and `ms_per_step_at_reference_box`;
  dense_ms_per_step  the same step with the reference's dense key-masked shapes, timed in this run (5 steps);
  replicas_equal     (N > 1) every rank ends the run with bit-identical parameters;
  allreduce          (N > 1, or one rank under torch.distributed.run) how long the gradient buckets had to travel while the
                     backward still ran, and how long the optimiser then still waited for them (HIP events, median over the
                     timed steps).
Everything after the timed region (dense-shape block, replica check, instrumented step, CPU baseline) is optional: a failure
there is reported on stderr and as a null field; the JSON line is printed regardless.  A watchdog thread ends the process with
exit code 3 (after printing where every stage stood) if the whole run exceeds WJ_BENCH_LIMIT_S seconds (default 900): a stalled
collective must not hang the launch.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

CONV_SPEC = [(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512, 2, 2)]
MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense; /opt/skills/guides/MI355X_MICROARCH.md (AMD's 5 PF figure is 2:1 sparse)
MFMA_FP8_PEAK_TFLOPS = 5000.0    # dense, block-scaled (MX) e4m3: the scaled MFMA form is the one that runs at 2x the bf16 rate
WORKLOADS = {   # name: (seconds per clip, fp8 forward GEMMs, binaural scene front-end)
    "2s-bf16": (2.01, False, False),    # BASELINE config 2 / 3: the headline metric
    "4s-bf16": (4.01, False, False),    # 400 tokens, bf16
    "4s-fp8": (4.01, True, False),      # BASELINE config 5: 400 tokens, MX fp8 forward GEMMs
    "2s-nat": (2.01, False, True),      # BASELINE config 4: scene augmentation + 2-channel front-end (2 x 200 tokens)
}
STEP_GFLOP_PER_CLIP = 283.7      # dense algorithmic FLOPs of one step per clip (fwd 118.2 + bwd 165.5)


CALIBRATION_REFERENCE = os.path.join(ROOT, "profiles", "r06_calibration_reference.json")


def _sclk_mhz():
    """The shader clock level sysfs marks as current (pp_dpm_sclk), or None.  A label only: under an MFMA-dense loop the in-kernel clock
    reads up to ~10 % below it (MI355X_MICROARCH.md, DVFS give-back item 6)."""
    import glob
    for path in sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")):
        try:
            for ln in open(path):
                if "*" in ln:
                    return int(ln.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
        except (OSError, ValueError, IndexError):
            continue
    return None


class Calibration:
    """Two fixed launches from the library that say how fast THIS box is (the pool's boxes differ by ~4 % in the clock they hold under
    load, which is the size of a round's gain): the persistent bf16 GEMM on an 8192 x 8192 x 1024 problem whose operands stay in the
    Infinity Cache (MFMA-bound: 137.4 GFLOP per launch) and wj_ema_update over 2^28 floats (HBM-bound: 12 bytes per element = 3.22 GB per
    launch).  Run before the warm-up and after the timed region, never inside it; each figure is the mean over a burst of back-to-back
    launches between ONE pair of HIP timing events (40 GEMMs ~ 6 ms behind 100 untimed ones, 10 EMAs ~ 6 ms)."""
    M, N, K = 8192, 8192, 1024
    EMA_N = 1 << 28

    def __init__(self, device):
        from wavjepa_amd import ops
        self.ops, self.dev = ops, device
        g = torch.Generator(device=device).manual_seed(1234)
        bf = torch.bfloat16
        self.A = torch.randn(self.M, self.K, device=device, generator=g).to(bf)
        self.B = (torch.randn(self.N, self.K, device=device, generator=g) * 0.05).to(bf)
        self.C = torch.empty(self.M, self.N, dtype=bf, device=device)
        self.s = torch.randn(self.EMA_N, device=device, generator=g)
        self.t = torch.randn(self.EMA_N, device=device, generator=g)

    def _burst(self, fn, warm: int, reps: int) -> float:
        """milliseconds per launch"""
        ops = self.ops
        stream = torch.cuda.current_stream().cuda_stream
        for _ in range(warm):
            fn()
        e0, e1 = ops.TimingEvent(), ops.TimingEvent()
        e0.record(stream)
        for _ in range(reps):
            fn()
        e1.record(stream)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    def measure(self) -> dict:
        ops = self.ops
        # (100 untimed launches first: ~15 ms, the time the chip takes to settle on the clock it holds under a matrix load -- a burst
        # timed from an idle chip read 999 TFLOP/s against 1178 for the same launches behind the timed region)
        gemm_ms = self._burst(lambda: ops.gemm(self.A, self.B, self.C, M=self.M, N=self.N, K=self.K, lda=self.K, ldb=self.K, ldc=self.N), 100, 40)
        ema_ms = self._burst(lambda: ops.ema_update(self.s, self.t, self.EMA_N, 0.999), 2, 10)
        return dict(mfma_tflops=round(2.0 * self.M * self.N * self.K / (gemm_ms * 1e-3) / 1e12, 1),
                    hbm_tbps=round(12.0 * self.EMA_N / (ema_ms * 1e-3) / 1e12, 3), sclk_mhz=_sclk_mhz())


def run_calibration(device) -> dict:
    """Allocate the calibration operands (2.3 GB), measure, free them again (the step's peak-memory figure must not see them)."""
    c = Calibration(device)
    try:
        return c.measure()
    finally:
        del c


def calibration_summary(before, after, mfma_weight):
    """`calibration` object of the bench line: both measurements, their mean, the committed reference box's figures and this box's speed
    relative to it -- speed = w x (mfma here / mfma there) + (1 - w) x (hbm here / hbm there), w = the GEMM share of the step's kernel time
    (instrumented step of this run; 0.68 when that leg did not run)."""
    if not before and not after:
        return None
    got = [c for c in (before, after) if c]
    here = dict(mfma_tflops=round(sum(c["mfma_tflops"] for c in got) / len(got), 1), hbm_tbps=round(sum(c["hbm_tbps"] for c in got) / len(got), 3))
    out = dict(before=before, after=after, mean=here, mfma_weight=round(mfma_weight, 3),
               workloads="persistent bf16 GEMM 8192x8192x1024 (40 launches back to back); wj_ema_update on 2^28 floats (10 launches, 12 B per element)")
    ref = None
    if os.path.exists(CALIBRATION_REFERENCE):
        try:
            ref = json.load(open(CALIBRATION_REFERENCE))
        except (OSError, ValueError):
            ref = None
    if ref:
        speed = mfma_weight * here["mfma_tflops"] / ref["mfma_tflops"] + (1.0 - mfma_weight) * here["hbm_tbps"] / ref["hbm_tbps"]
        out["reference_box"] = dict(mfma_tflops=ref["mfma_tflops"], hbm_tbps=ref["hbm_tbps"], source="profiles/" + os.path.basename(CALIBRATION_REFERENCE),
                                    note=ref.get("note"))
        out["speed_vs_reference_box"] = round(speed, 4)
        out["mfma_vs_reference_box"] = round(here["mfma_tflops"] / ref["mfma_tflops"], 4)
    return out


def build_model(device, seed: int, seconds: float = 2.01, nat: bool = False):
    from wavjepa_amd.extractors import ConvChannelFeatureExtractor, ConvFeatureExtractor
    from wavjepa_amd.jepa import JEPA
    from wavjepa_amd.types import TransformerEncoderCFG, TransformerLayerCFG
    torch.manual_seed(seed)
    if nat:      # WavJEPA-Nat: each ear through its own mono conv stack, tokens of both channels in one sequence
        ext = ConvChannelFeatureExtractor(conv_layers_spec=CONV_SPEC, in_channels=2, share_weights_over_channels=False)
    else:
        ext = ConvFeatureExtractor(conv_layers_spec=CONV_SPEC, in_channels=1)
    model = JEPA(feature_extractor=ext, transformer_encoder_cfg=TransformerEncoderCFG.create(),
                 transformer_encoder_layers_cfg=TransformerLayerCFG.create(), transformer_decoder_cfg=TransformerEncoderCFG.create(),
                 transformer_decoder_layers_cfg=TransformerLayerCFG.create(d_model=384), lr=4e-4, adam_betas=(0.9, 0.98),
                 adam_weight_decay=0.04, resample_sr=16000, process_audio_seconds=seconds, nr_samples_per_audio=8,
                 average_top_k_layers=8, size="base")
    return model.to(device)


def gemm_kernel_name(f) -> str:
    epi = {0: "BF16", 1: "BIAS_GELU2", 2: "MUL_GELU_GRAD", 3: "ADD_F32", 4: "ATOMIC_F32", 5: "CONV_GELU", 6: "BIAS_GELU", 7: "MUL_GELU_GRAD_Z", 8: "BF16_ADD_POS"}[f["epilogue"]]
    return f"gemm_kernel<{'T' if f['a_trans'] else 'N'}{'T' if f['b_trans'] else 'N'},{epi}>"


PMC_FILES = ["r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"]


def pmc_traffic_path():
    for f in PMC_FILES:
        if os.path.exists(os.path.join(ROOT, "profiles", f)):
            return os.path.join(ROOT, "profiles", f)
    return None


def pmc_traffic_for(name: str):
    """HBM bytes per launch of a GEMM class from the newest committed PMC summary (profiles/rNN_pmc_traffic.json, collected with
    tools/pmc_traffic.py on this same command in two separate rocprofv3 --pmc passes): launch-weighted mean over the tile variants
    of the class, or None.  NOT measured by this run: the line says so in `traffic_source`."""
    path = pmc_traffic_path()
    if path is None:
        return None
    m = __import__("re").match(r"gemm_kernel<([NT])([NT]),(\w+)>", name)
    if not m or not os.path.exists(path):
        return None
    epi = {"BF16": 0, "BIAS_GELU2": 1, "BIAS_GELU": 1, "MUL_GELU_GRAD": 2, "ADD_F32": 3, "ATOMIC_F32": 4, "CONV_GELU": 5, "MUL_GELU_GRAD_Z": 7, "BF16_ADD_POS": 8}[m.group(3)]
    prefix = f"gemm3_kernel<{'true' if m.group(1) == 'T' else 'false'}, {'true' if m.group(2) == 'T' else 'false'}, {epi},"
    pn = 6 if m.group(3) == "BIAS_GELU" else epi                                      # the persistent variant of the forward shapes
    persist = (f"gemm_persist_kernel<{pn}>", f"gemm_persist_kernel<{pn},")
    with open(path) as fh:
        kernels = json.load(fh)["kernels"]
    hits = [v for k, v in kernels.items() if k.startswith(prefix) or (m.group(1) == "N" and m.group(2) == "N" and k.startswith(persist))]
    n = sum(v["launches"] for v in hits)
    return int(sum(v["hbm_bytes_per_launch"] * v["launches"] for v in hits) / n) if n else None


def gemm_class_patterns(name: str):
    """Kernel-name patterns (rocprofv3 summary) of a GEMM class of the instrumented step, or None."""
    import re
    m = re.match(r"gemm_kernel<([NT])([NT]),(\w+)>", name)
    if not m:
        return None
    epi = {"BF16": 0, "BIAS_GELU2": 1, "BIAS_GELU": 1, "MUL_GELU_GRAD": 2, "ADD_F32": 3, "ATOMIC_F32": 4, "CONV_GELU": 5, "MUL_GELU_GRAD_Z": 7, "BF16_ADD_POS": 8}[m.group(3)]
    pn = 6 if m.group(3) == "BIAS_GELU" else epi
    pats = [f"gemm3_kernel<{'true' if m.group(1) == 'T' else 'false'}, {'true' if m.group(2) == 'T' else 'false'}, {epi},"]
    if m.group(1) == "N" and m.group(2) == "N":
        pats += [f"gemm_persist_kernel<{pn}>", f"gemm_persist_kernel<{pn},"]
    return pats


def committed_rocprof_serial(name: str, workload: str, clips_per_gpu: int):
    """The class's rate in the COMMITTED `rocprofv3 --kernel-trace --stats` summary of this command with the side stream serialised
    (profiles/r06_serial_meta.json -- or the previous round's --, written by tools/serial_meta.py next to the CSV it describes: that run's own algorithmic flops of the
    class over that run's own kernel time, with its commit and configuration).  Returned only when the committed run had this run's
    workload and clips per GPU -- a cross-check from another box, labelled as such, never a judged metric."""
    path = next((p for p in (os.path.join(ROOT, "profiles", f) for f in ("r06_serial_meta.json", "r05_serial_meta.json")) if os.path.exists(p)), None)
    if path is None:
        return None
    meta = json.load(open(path))
    c = meta.get("classes", {}).get(name)
    if c is None or meta.get("workload") != workload or int(meta.get("clips_per_gpu", -1)) != int(clips_per_gpu):
        return None
    return dict(frac=c["frac"], achieved_tflops=c["tflops"], kernel_ms_per_step=c["kernel_ms_per_step"], gflop_per_step=c["gflop_per_step"],
                source=f"profiles/{meta['csv']} at commit {meta.get('commit', '?')} ({meta.get('steps')} steps, {meta.get('clips_per_gpu')} clips/GPU): "
                       "committed summary of this command, NOT collected by this run")


def profile_one_step(runner, source, step_idx: int):
    """Run one extra step with every C-ABI call bracketed by HIP events; returns per-kernel-class totals."""
    from wavjepa_amd import ops
    eng = runner.model._engine
    side = eng.use_side
    eng.use_side = False            # serialise the side-stream work so that every kernel is timed alone
    ops.PROFILE = []
    try:
        runner.step(source.next_batch(), step_idx)
        torch.cuda.synchronize()
        recs = ops.PROFILE
        # what the bracket itself costs: the same two events around a one-wave kernel that returns at once
        ops.PROFILE = []
        for _ in range(64):
            ops.spin(0)
        torch.cuda.synchronize()
        empty = sorted(e0.elapsed_time(e1) for _, _, e0, e1 in ops.PROFILE)
        profile_one_step.bracket_us = round(empty[len(empty) // 2] * 1e3, 2)
    finally:
        ops.PROFILE = None
        eng.use_side = side
    classes, shapes = {}, {}
    for fn, f, e0, e1 in recs:
        ms = e0.elapsed_time(e1)
        if fn == "wj_wgrad_grouped":
            name, flops, nbytes = "wgrad_grouped<TT,ATOMIC_F32>", f["flops"], f["bytes"]
        elif fn == "wj_gemm_mxfp8":
            name, flops = f"gemm_mxfp8<NN,{ {0: 'BF16', 1: 'BIAS_GELU2', 6: 'BIAS_GELU'}[f['epilogue']] }>", 2.0 * f["M"] * f["N"] * f["K"]
            nbytes = 1.03 * f["K"] * (f["M"] + f["N"]) + 2.0 * f["M"] * f["N"] * (2 if f["epilogue"] == 1 else 1)
            key = f"{name} M={f['M']} N={f['N']} K={f['K']}"
            sh = shapes.setdefault(key, dict(ms=0.0, flops=0.0, launches=0))
            sh["ms"] += ms; sh["flops"] += flops; sh["launches"] += 1
        elif fn == "wj_gemm_bf16":
            name, flops = gemm_kernel_name(f), 2.0 * f["M"] * f["N"] * f["K"]
            ep = f["epilogue"]
            outb = 4 if ep in (3, 4) else 2
            nbytes = 2.0 * f["K"] * (f["M"] + f["N"]) + f["M"] * f["N"] * (outb * (2 if ep in (1, 5) else 1) + (outb if ep in (2, 3, 7) else 0) + (4 if ep == 8 else 0))
            key = f"{name} M={f['M']} N={f['N']} K={f['K']}" + (" gather" if f.get("rowmap") else "")
            sh = shapes.setdefault(key, dict(ms=0.0, flops=0.0, launches=0))
            sh["ms"] += ms; sh["flops"] += flops; sh["launches"] += 1
        else:
            name, flops, nbytes = fn, 0.0, 0.0
        c = classes.setdefault(name, dict(ms=0.0, flops=0.0, launches=0, bytes=0.0))
        c["ms"] += ms
        c["flops"] += flops
        c["bytes"] += nbytes
        c["launches"] += 1
    profile_one_step.shapes = shapes
    return classes


def _cpu_model() -> str:
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _cpu_baseline_child(threads, n_clips: int, configs) -> None:
    """Child process: time the oracle's train step for each (model, mode) and print one JSON line per finished configuration (the parent
    enforces the wall-clock bound by killing this process).  `threads` = a list of candidate thread counts: the FIRST configuration is
    probed at each (ascending, two steps each, stopping as soon as doubling the threads no longer pays 10 %) and everything is then
    measured at the fastest; one `probe` line says what was tried."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import synth
    from oracle import jepa_oracle as J
    from oracle import masking_oracle as M
    cands = sorted(set(int(t) for t in (threads if isinstance(threads, (list, tuple)) else [threads])))
    torch.set_num_threads(cands[0])
    tiny_spec = [(64, 10, 5)] + [(64, 3, 2)] * 4 + [(64, 2, 2)]
    models = {"tiny": dict(spec=tiny_spec, d_enc=128, h_enc=4, l_enc=2, d_dec=64, h_dec=4, l_dec=2, top_k=2),
              "base": dict(spec=CONV_SPEC, d_enc=768, h_enc=12, l_enc=12, d_dec=384, h_dec=12, l_dec=12, top_k=8)}
    rng = np.random.default_rng(0)
    ctx, tgt, vis = M.time_inverse_block_masks(n_clips, 200, 1, new_rng=lambda: np.random.default_rng(rng.integers(1 << 31)))
    audio = torch.randn(n_clips, 1, 32159, generator=torch.Generator().manual_seed(0))
    masks = (torch.from_numpy(ctx), torch.from_numpy(tgt), torch.from_numpy(vis))
    best_threads = cands[0]
    for ci, (name, mode) in enumerate(configs):
        c = models[name]
        shapes = synth.jepa_shapes(conv_spec=c["spec"], in_channels=1, d_enc=c["d_enc"], enc_layers=c["l_enc"], d_dec=c["d_dec"],
                                   dec_layers=c["l_dec"], n_tokens=200)
        P = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, seed=7).items()}
        P["pos_encoding_encoder"] = J.sincos_positions(c["d_enc"], 200)
        P["pos_encoding_decoder"] = J.sincos_positions(c["d_dec"], 200)
        batch = (audio if mode == "fp32" else audio.to(torch.bfloat16),) + masks
        kw = dict(mode=mode, spec=c["spec"], enc_heads=c["h_enc"], dec_heads=c["h_dec"], top_k=c["top_k"])
        state, times, step = {}, [], 0
        if ci == 0 and len(cands) > 1:
            probe, prev = {}, None
            for t in cands:
                torch.set_num_threads(t)
                dts = []
                for _ in range(2):
                    t0 = time.perf_counter()
                    J.train_step(P, state, step, batch, **kw)
                    step += 1
                    dts.append(time.perf_counter() - t0)
                probe[t] = min(dts)
                print(json.dumps(dict(probe={str(k): round(v * 1000, 1) for k, v in probe.items()}, config=f"{name}_{mode}")), flush=True)
                if prev is not None and probe[t] > 0.9 * prev:
                    break
                prev = probe[t]
            best_threads = min(probe, key=probe.get)
        torch.set_num_threads(best_threads)
        for i in range(8):                            # 3 warm-up + 5 timed; a line after every timed step (the latest one counts)
            t0 = time.perf_counter()
            J.train_step(P, state, step, batch, **kw)
            step += 1
            dt = time.perf_counter() - t0
            warm_needed = 1 if (ci == 0 and len(cands) > 1) else 3       # the probe's steps were this configuration's warm-up
            if i >= warm_needed or dt > 4.0:          # slow configurations count from their first step on
                times.append(dt)
                med = sorted(times)[len(times) // 2]
                print(json.dumps(dict(config=f"{name}_{mode}", clips_per_s=round(n_clips / med, 3), ms_per_step=round(med * 1000, 1),
                                      warmup=i + 1 - len(times), timed=len(times), threads=best_threads)), flush=True)
                if len(times) >= 5:
                    break
        del P, state


def cpu_baseline(n_clips: int = 4, budget_s: float = 60.0):
    """The oracle's full train step on the host cores: a port of the reference algorithm (kind='port'), N = 4 clips, base
    (12-layer d=768) and tiny (2-layer d=128) models, fp32 and the bf16-autocast flow, median of 5 timed steps each -- in a CHILD process
    that is killed at the wall-clock bound (`budget_s`).  The thread count is PROBED, not assumed: two steps of the base fp32 configuration at
    16, 32, 64, ... up to every hardware thread, ascending, stopping when doubling no longer pays 10 % (on the pool's 256-thread hosts the
    oracle is fastest at 32-64 threads and an all-core step takes longer than the whole budget: rounds 1-5 spent 38 s of every bench run
    learning that).  `value` is the BASE fp32 rate at the probed thread count; everything measured is listed in `detail`."""
    import subprocess
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    order = [("base", "fp32"), ("tiny", "fp32"), ("base", "bf16"), ("tiny", "bf16")]
    cands = sorted({t for t in (8, 16, 32, 64, 128, 256, avail) if t <= avail} | ({avail} if avail < 8 else set()))
    detail, probe, t_start = {}, {}, time.perf_counter()
    spec = json.dumps(dict(threads=cands, n_clips=n_clips, configs=order))
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="", OMP_NUM_THREADS=str(max(cands)))
    proc = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", spec], stdout=subprocess.PIPE,
                            stderr=subprocess.DEVNULL, text=True, env=env, cwd=ROOT)
    try:
        out, _ = proc.communicate(timeout=budget_s)
    except subprocess.TimeoutExpired:
        proc.kill()
        out, _ = proc.communicate()
    for ln in (out or "").splitlines():
        if ln.startswith("{"):
            r = json.loads(ln)
            if "probe" in r:
                probe = r["probe"]
                continue
            detail[f"{r.pop('config')}@{r['threads']}t"] = r
    base = {k: v for k, v in detail.items() if k.startswith("base_fp32@")}
    if not base:
        return dict(value=None, unit="clips/s", cores=None, host_cores=avail, kind="port", cpu=_cpu_model(),
                    sample="no configuration finished inside the bound", thread_probe_ms_per_step=probe, detail=detail)
    key, head = max(base.items(), key=lambda kv: kv[1]["clips_per_s"])
    # cores = the threads the reported value was measured with (the probe's pick); host_cores = hardware threads this process may use
    return dict(value=head["clips_per_s"], unit="clips/s", cores=head["threads"], host_cores=avail, kind="port", cpu=_cpu_model(),
                sample=f"oracle train step (fwd+EMA+bwd+clip+AdamW), N={n_clips} clips of 2.01 s; value = WavJEPA-base fp32 on {head['threads']} of "
                       f"{avail} host threads (thread count picked by a 2-step probe over {list(probe) or cands}), median of {head['timed']} timed "
                       f"steps ({head['ms_per_step']} ms/step); {time.perf_counter() - t_start:.0f} s of wall clock in all (bounded at {budget_s:.0f} s)",
                thread_probe_ms_per_step=probe, detail=detail)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--clips-per-gpu", type=int, default=256)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="2s-bf16",
                    help="2s-bf16 = the headline metric (BASELINE config 2/3); 4s-fp8 = BASELINE config 5 (400 tokens, MX fp8 forward GEMMs); "
                         "2s-nat = BASELINE config 4 (device-side scene augmentation + 2-channel front-end)")
    ap.add_argument("--dense-steps", type=int, default=5, help="extra timed steps with the dense (non-ragged) shapes; 0 = skip")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-calibration", action="store_true", help="skip the box-speed calibration launches around the timed region")
    ap.add_argument("--seed", type=int, default=None,
                    help="pin the masks (the maskers draw from OS entropy, as upstream) and the crop starts / shuffles: two runs with the same "
                         "seed see the same batches (A/B comparisons of a loss; the default run stays unpinned)")
    ap.add_argument("--emulate-allreduce", action="store_true",
                    help="one GPU: at every gradient-bucket hook a copy kernel confined to the CUs a data-parallel run leaves free moves 2 x the "
                         "bucket's bytes on a communication stream (the HBM + CU footprint of reduce-scatter + all-gather); an EMULATION, "
                         "reported as `allreduce_emulated`, never part of a scaling record")
    args = ap.parse_args()

    from wavjepa_amd.data import SyntheticAudioSource
    from wavjepa_amd.masking import TimeInverseBlockMasker
    from wavjepa_amd.trainer import StepRunner, init_distributed

    stage = {"name": "start", "t": time.perf_counter()}

    def watchdog():
        import threading
        limit = float(os.environ.get("WJ_BENCH_LIMIT_S", "900"))

        def run():
            t_start = time.perf_counter()
            while time.perf_counter() - t_start < limit:
                time.sleep(2.0)
            print(f"[bench rank {os.environ.get('RANK', '0')}] exceeded {limit:.0f} s; last stage entered: {stage['name']} "
                  f"({time.perf_counter() - stage['t']:.0f} s ago) -- exiting with code 3", file=sys.stderr, flush=True)
            os._exit(3)
        threading.Thread(target=run, daemon=True).start()

    watchdog()
    if args.emulate_allreduce:
        os.environ["WJ_EMULATE_ALLREDUCE"] = "1"
    rank, local, world = init_distributed()
    if args.emulate_allreduce and world != 1:
        raise SystemExit("--emulate-allreduce rehearses the collective's footprint on ONE GPU; with N > 1 ranks the real all-reduce runs")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    device = torch.device("cuda", local)
    S = 8
    if args.clips_per_gpu % S:
        raise SystemExit("--clips-per-gpu must be a multiple of 8 (8 crops per source audio)")
    seconds, fp8, nat = WORKLOADS[args.workload]
    model = build_model(device, seed=42, seconds=seconds, nat=nat)  # same init on every rank (+ broadcast from rank 0 in StepRunner)
    model._ensure_engine().fp8 = fp8
    model.trainer.max_steps = 375000
    masker = TimeInverseBlockMasker(target_masks_per_context=4, context_mask_prob=0.65, context_mask_length=10, target_prob=0.25,
                                    target_length=10, ratio_cutoff=0.1, channel_based_masking=nat, channel_major=nat)   # configs/masker/AudioSet.yaml
    src_kw = dict(batch_size=args.clips_per_gpu // S, samples_per_audio=S, n_tokens=model.total_patches, seed=42 + rank, n_mask_sets=64,
                  device=device)                                                      # masks pre-generated for 64 steps and cycled
    import contextlib
    import numpy as np

    @contextlib.contextmanager
    def pinned_masks(seed):
        """Inside: the k-th np.random.default_rng() call of the maskers returns default_rng(seed + k) (--seed)."""
        if seed is None:
            yield
            return
        orig, k = np.random.default_rng, [0]

        def pinned(_seed=None):
            k[0] += 1
            return orig(seed + 1000 * rank + k[0])
        np.random.default_rng = pinned
        try:
            yield
        finally:
            np.random.default_rng = orig

    with pinned_masks(args.seed):
        if nat:
            from wavjepa_amd.data import NatSceneSource
            source = NatSceneSource(masker, **src_kw)          # scene generation runs inside the timed step (next_batch)
        else:
            source = SyntheticAudioSource(masker, **src_kw)
    if args.seed is not None:            # crop starts (device generator) and the crop shuffle (CPU generator) of on_after_batch_transfer
        torch.manual_seed(args.seed + rank)
        torch.cuda.manual_seed(args.seed + rank)
    runner = StepRunner(model, gradient_clip_val=5.0)
    # Data-parallel runs leave CUs to RCCL: a persistent GEMM workgroup owns its CU's whole register file and 150 KB of its LDS, so a
    # channel kernel of the gradient all-reduce can only start on a CU the persistent kernel does not occupy.  28 of 32 workgroups per
    # XCD leave 32 CUs free (WJ_PERSIST_CUS overrides; StepRunner applies the same default through trainer.init_persist_cus).
    from wavjepa_amd import ops as _ops
    persist_cus = _ops.gemm_set_persist_cus(0)

    def sync():
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
            torch.cuda.synchronize()

    t_begin = time.perf_counter()

    def note(msg):
        stage["name"], stage["t"] = msg, time.perf_counter()
        if rank == 0:
            print(f"[bench +{time.perf_counter() - t_begin:.1f}s] {msg}", file=sys.stderr, flush=True)

    coll = {"broken": False}

    def optional(what, fn, collective=False):
        """Legs after the timed region: never lose the JSON line over them.  A leg that contains collectives (`collective`) and
        fails on this rank leaves the other ranks inside a collective this rank will never join; issuing FURTHER collectives from
        here would pair them with the wrong ones.  So after such a failure this rank issues no more collectives at all (the legs
        below are skipped, the process group is not torn down collectively); the other ranks leave theirs through the process-group
        timeout (WJ_DIST_TIMEOUT_S) and then do the same."""
        if collective and coll["broken"]:
            print(f"[bench rank {rank}] optional leg '{what}' skipped: an earlier collective leg failed on this rank", file=sys.stderr, flush=True)
            return None
        try:
            return fn()
        except BaseException as e:                      # noqa: BLE001  (SystemExit of a check included: reported, not fatal)
            if isinstance(e, KeyboardInterrupt):
                raise
            if collective and dist.is_initialized() and world > 1:
                coll["broken"] = True
            print(f"[bench rank {rank}] optional leg '{what}' failed: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
            return None

    step_idx = 0
    calib_before = calib_after = None
    if not args.no_calibration:
        note("calibration launches (before)")
        calib_before = optional("calibration (before)", lambda: run_calibration(device))
        torch.cuda.reset_peak_memory_stats(device)
    note("warm-up")
    for _ in range(args.warmup):
        runner.step(source.next_batch(), step_idx)
        step_idx += 1
    sync()
    note("timed region")
    runner.reducer.timing = runner.reducer.active or runner.reducer.emulate
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = runner.step(source.next_batch(), step_idx)
        step_idx += 1
    sync()
    elapsed = time.perf_counter() - t0
    runner.reducer.timing = False
    allreduce = runner.reducer.timing_summary()
    peak_timed_gb = torch.cuda.max_memory_allocated(device) / 1e9      # before the optional dense-shape block grows the arena
    if not args.no_calibration:
        note("calibration launches (after)")
        calib_after = optional("calibration (after)", lambda: run_calibration(device))
    if dist.is_initialized():
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    loss = float(out["loss"].detach())
    if not (loss == loss):
        raise SystemExit("loss is NaN")

    note(f"timed region done: {elapsed / args.steps * 1000:.1f} ms/step; side-stream probe (pair / single wait per candidate): "
         f"{getattr(model._engine, '_stream_probe', None)}")
    # the same step with the reference's dense key-masked shapes (every token of the student / predictor computed), timed in the
    # same run so that the dense-shape rate is measured here too, not only reported
    def dense_block():
        nonlocal step_idx
        model._engine.ragged = False
        try:
            for _ in range(2):
                runner.step(source.next_batch(), step_idx); step_idx += 1
            sync()
            td = time.perf_counter()
            for _ in range(args.dense_steps):
                runner.step(source.next_batch(), step_idx); step_idx += 1
            sync()
            ms = (time.perf_counter() - td) / args.dense_steps * 1000
            if dist.is_initialized():
                t = torch.tensor([ms], dtype=torch.float64, device=device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                ms = float(t)
            return ms
        finally:
            model._engine.ragged = True

    dense_ms = None
    # one GPU only: the dense-shape rate is a property of the kernels, not of the scaling run, and every extra leg with collectives
    # in an N-rank run is one more way to lose ranks after the metric has been measured
    if args.dense_steps > 0 and model._engine.ragged and world == 1:
        note("dense-shape block")
        dense_ms = optional("dense-shape block", dense_block, collective=True)
        if dense_ms is not None:
            note(f"dense-shape block done: {dense_ms:.1f} ms/step")
    # data-parallel self-check: after the same number of identical optimiser steps every rank must hold the same parameters
    flat = model._flat
    checksum = torch.stack([flat.p32.double().sum(), flat.p32.double().abs().sum(), flat.t32.double().abs().sum()])
    replicas_equal = None
    diverged = None

    def replica_check():
        gathered = [torch.empty_like(checksum) for _ in range(world)]
        dist.all_gather(gathered, checksum)
        return gathered

    if dist.is_initialized():
        note("replica check")
        gathered = optional("replica check", replica_check, collective=True)
        if gathered is not None:
            replicas_equal = all(torch.equal(gathered[0], g) for g in gathered)
            if not replicas_equal:
                diverged = f"data-parallel replicas diverged: parameter checksums per rank {[g.tolist() for g in gathered]}"
    roofline = None
    classes = {}
    executed_gflop = None
    if not args.no_profile:
        note("instrumented step")
        classes = optional("instrumented step", lambda: profile_one_step(runner, source, step_idx), collective=True) or {}   # (a full step: its gradient all-reduce is a collective)
        gemms = {k: v for k, v in classes.items() if v["flops"] > 0}
        executed_gflop = sum(v["flops"] for v in gemms.values()) / 1e9      # GEMM flops one step actually executes
        if gemms:
            name, c = max(gemms.items(), key=lambda kv: kv[1]["ms"])
            if fp8:                    # config 5: the roofline of the fp8 GEMM family, against the dense fp8 MFMA peak
                f8 = {k: v for k, v in gemms.items() if k.startswith("gemm_mxfp8")}
                name = max(f8.items(), key=lambda kv: kv[1]["ms"])[0]
                c = dict(ms=sum(v["ms"] for v in f8.values()), flops=sum(v["flops"] for v in f8.values()),
                         launches=sum(v["launches"] for v in f8.values()), bytes=sum(v["bytes"] for v in f8.values()))
                name = "gemm_mxfp8<NN,*> (all epilogues; largest: " + name + ")"
            peak = MFMA_FP8_PEAK_TFLOPS if fp8 else MFMA_BF16_PEAK_TFLOPS
            # A launch is timed between two HIP events on its stream.  The same bracket around a kernel that returns at once measures
            # `event_bracket_us` (median of 64, this run): the event packets' own queue time, which a kernel-trace duration (rocprofv3) does
            # not contain.  `achieved` is computed from the launch durations WITH that constant taken off -- so that it is the figure the
            # committed rocprofv3 summary's average kernel duration gives (round 5: 0.339 in-run against 0.374 committed, a 5.9-us bracket
            # on a 89-us launch being most of the gap); `achieved_with_event_bracket` is the uncorrected quotient.
            br_ms = (getattr(profile_one_step, "bracket_us", 0.0) or 0.0) * 1e-3
            empty_kernel_ms = 0.0015                           # what rocprofv3 reports for the empty kernel itself (~1.5 us): stays in
            corr = max(0.0, br_ms - empty_kernel_ms)
            k_ms = lambda v: max(v["ms"] - corr * v["launches"], 0.25 * v["ms"])
            all_ms = sum(k_ms(v) for v in gemms.values())
            all_fl = sum(v["flops"] for v in gemms.values())
            ach_raw = c["flops"] / (c["ms"] * 1e-3) / 1e12
            ach = c["flops"] / (k_ms(c) * 1e-3) / 1e12
            roofline = dict(bound="mfma", kernel=name, achieved=round(ach, 1), peak=peak, unit="TFLOP/s",
                            frac=round(ach / peak, 4),
                            achieved_with_event_bracket=round(ach_raw, 1), frac_with_event_bracket=round(ach_raw / peak, 4),
                            # the same class by KERNEL time in the committed rocprofv3 summary (its own flops over its own kernel time; only when
                            # that run had this run's configuration; a cross-check, not collected by this run)
                            committed_rocprof_serial=None if fp8 else optional("rocprof summary", lambda: committed_rocprof_serial(name, args.workload, args.clips_per_gpu)),
                            traffic=None if fp8 else optional("pmc summary", lambda: pmc_traffic_for(name)),
                            traffic_source=None if fp8 or pmc_traffic_path() is None else
                            f"profiles/{os.path.basename(pmc_traffic_path())} (rocprofv3 --pmc passes of this command, committed; not collected by this run)",
                            launches_per_step=c["launches"],
                            avg_launch_ms=round(k_ms(c) / c["launches"], 4),
                            avg_launch_ms_with_event_bracket=round(c["ms"] / c["launches"], 4),
                            # the bracket around a kernel that returns at once (median of 64); `achieved` has (this - 1.5 us) taken off per launch
                            event_bracket_us=getattr(profile_one_step, "bracket_us", None),
                            gflop_per_launch=round(c["flops"] / c["launches"] / 1e9, 2),
                            algorithmic_bytes_per_launch=int(c["bytes"] / c["launches"]),   # operands once + outputs (+ addends)
                            all_gemm_achieved=round(all_fl / (all_ms * 1e-3) / 1e12, 1),
                            gemm_share_of_step=round(all_ms / sum(k_ms(v) for v in classes.values()), 3))

    if rank == 0:
        clips = args.clips_per_gpu * world * args.steps
        value = clips / elapsed
        line = {
            "metric": "jepa_pretrain_clips_per_sec", "value": round(value, 1), "unit": "clips/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1000, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "fp8-e4m3 (MX) forward GEMMs, bf16 elsewhere" if fp8 else "bf16",
            "data": "synthetic",
            "config": {"workload": f"WavJEPA-base JEPA pre-training step, {seconds} s @16 kHz white-noise clips ({model.target_length} samples -> "
                                   f"{model.total_patches} tokens), {args.clips_per_gpu} clips per GPU ({args.clips_per_gpu // 8} sources x 8 crops), "
                                   "AudioSet masker, random-init weights" + (", MX fp8 forward GEMMs (BASELINE config 5)" if fp8 else "")
                                   + (", binaural scenes generated on the device inside the step (source RIR + 2 noise RIRs of 0.5 s, segmental-SNR mix) "
                                      "and a 2-channel ConvChannelFeatureExtractor front-end (BASELINE config 4)" if nat else ""),
                       "workload_name": args.workload,
                       "global_batch": args.clips_per_gpu * world, "seq_len": model.total_patches, "parallelism": f"dp{world}",
                       # student / predictor run on their visible tokens only unless WJ_RAGGED=0 (same loss and gradients:
                       # the dropped rows are key-masked and carry zero loss weight on the reference)
                       "token_execution": "ragged" if model._engine.ragged else "dense",
                       # resident workgroups per XCD of the persistent GEMM (32 = every CU; N > 1 leaves CUs to the RCCL kernels)
                       "persistent_gemm_workgroups_per_xcd": persist_cus,
                       # who carries the gradient buckets: torch.distributed (backend nccl = RCCL) or, WJ_RCCL_DIRECT=1, the library's
                       # own RCCL binding (include/wavjepa_hip.h wj_rccl_bucket_allreduce_*); None without a process group
                       "gradient_transport": None if not runner.reducer.active else ("wj_rccl_bucket_allreduce" if runner.reducer.direct else "torch.distributed"),
                       "step_gflop_per_clip_dense": STEP_GFLOP_PER_CLIP,
                       "step_gemm_gflop_per_clip_executed": None if executed_gflop is None else round(executed_gflop / args.clips_per_gpu, 1)},
            # executed GEMM flops (instrumented step) over the measured step time; NOT the dense-shape flop count
            "model_tflops_per_gpu": None if executed_gflop is None else round(executed_gflop / (elapsed / args.steps) / 1000, 1),
            # dense model FLOPs exactly as the reference computes them (283.7 GFLOP per clip and step), independent
            # of the rows the ragged execution does not compute -- the rate a dense-shape implementation would need to keep up
            # -- NOT a utilisation figure: most of those FLOPs are not executed (ragged execution)
            "dense_equivalent_rate_tflops": round(value / world * STEP_GFLOP_PER_CLIP / 1000, 1) if seconds < 3 and not nat else None,
            # the same step computed with the reference's dense shapes (WJ_RAGGED=0 equivalent), this run
            "dense_ms_per_step": None if dense_ms is None else round(dense_ms, 2),
            "dense_clips_per_s": None if dense_ms is None else round(args.clips_per_gpu * world / (dense_ms / 1000), 1),
            "allreduce": None if runner.reducer.emulate else allreduce,
            # --emulate-allreduce (one GPU): NOT a collective -- a paced in-place copy kernel on the CUs a data-parallel run keeps free
            # stands in for the all-reduce's CU + HBM share; ms_per_step of this line is then the step WITH that load beside the backward
            "allreduce_emulated": None if not runner.reducer.emulate else dict(
                allreduce, emulation=True, workgroups=runner.reducer._emu_wgs, assumed_busbw_gbps=runner.reducer._emu_gbps or None,
                assumed_world=runner.reducer._emu_world,
                note="one GPU; a copy kernel reads and rewrites every gradient bucket twice on a communication stream; nothing crosses xGMI"),
            "replicas_equal": replicas_equal, "param_checksum": [float(v) for v in checksum.tolist()],
            "final_loss": round(loss, 5),
            "peak_hbm_gb": round(peak_timed_gb, 1),                     # warm-up + timed region (the arena follows the ragged row counts)
            "peak_hbm_gb_after_dense_block": round(torch.cuda.max_memory_allocated(device) / 1e9, 1),
            "roofline": roofline,
        }
        cal = optional("calibration summary", lambda: calibration_summary(calib_before, calib_after, (roofline or {}).get("gemm_share_of_step") or 0.68))
        line["calibration"] = cal
        # the step time this run would have shown on the committed reference box: measured x (this box's speed / the reference box's)
        line["ms_per_step_at_reference_box"] = (round(line["ms_per_step"] * cal["speed_vs_reference_box"], 2)
                                                if cal and cal.get("speed_vs_reference_box") else None)
        if roofline and cal and cal.get("mfma_vs_reference_box"):
            roofline["frac_at_reference_box"] = round(roofline["frac"] / cal["mfma_vs_reference_box"], 4)
        if world == 1 and not args.no_cpu_baseline:
            note("instrumented step done; timing the CPU oracle")
            line["cpu_baseline"] = optional("cpu baseline", cpu_baseline)
            note("cpu baseline done")
        print(json.dumps(line), flush=True)
        if classes:
            top = sorted(classes.items(), key=lambda kv: -kv[1]["ms"])[:25]
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "bench_kernel_classes.json"), "w") as fh:
                json.dump({k: dict(ms=round(v["ms"], 3), launches=v["launches"],
                                   tflops=round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) if v["flops"] else None) for k, v in top}, fh, indent=1)
            shapes = sorted(getattr(profile_one_step, "shapes", {}).items(), key=lambda kv: -kv[1]["ms"])
            with open(os.path.join(ROOT, "gpurun_out", "bench_gemm_shapes.json"), "w") as fh:
                json.dump({k: dict(ms=round(v["ms"], 3), launches=v["launches"], us_per_launch=round(v["ms"] / v["launches"] * 1e3, 1),
                                   tflops=round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1)) for k, v in shapes}, fh, indent=1)
    if dist.is_initialized():
        optional("final barrier", lambda: (dist.barrier(), dist.destroy_process_group()), collective=True)
    if diverged:
        raise SystemExit(diverged)          # after the line: the record exists, the exit code says the run is invalid


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "--cpu-baseline-child":
        spec = json.loads(sys.argv[2])
        _cpu_baseline_child(spec["threads"], spec["n_clips"], [tuple(c) for c in spec["configs"]])
    else:
        main()
