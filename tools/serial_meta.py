#!/usr/bin/env python3
"""profiles/r05_serial_meta.json: what the committed serial rocprofv3 summary was measured on, and each GEMM class's rate IN that run --
its own algorithmic flops (the instrumented step of the same gpurun call, bench_gemm_shapes.json) over its own kernel time per step
(run_kernel_stats.csv).  bench.py quotes it (roofline.committed_rocprof_serial) only when a run has the same configuration.

    python3 tools/serial_meta.py <run_kernel_stats.csv> <gemm_shapes.json> --csv-name r05_bench_kernel_stats_serial.csv \
        --commit $(git rev-parse --short HEAD) --clips 256 --workload 2s-bf16 > profiles/r05_serial_meta.json
"""
import argparse
import csv
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("stats")
    ap.add_argument("shapes")
    ap.add_argument("--csv-name", required=True)
    ap.add_argument("--commit", default="?")
    ap.add_argument("--clips", type=int, default=256)
    ap.add_argument("--workload", default="2s-bf16")
    a = ap.parse_args()
    import bench
    rows = list(csv.DictReader(open(a.stats, newline="")))
    steps = max([int(r["Calls"]) for r in rows if "adamw_kernel" in r["Name"]] or [1])
    flops = {}
    for key, v in json.load(open(a.shapes)).items():
        m = re.match(r"(gemm_kernel<[NT][NT],\w+>) M=(\d+) N=(\d+) K=(\d+)", key)
        if m:
            flops[m.group(1)] = flops.get(m.group(1), 0.0) + 2.0 * int(m.group(2)) * int(m.group(3)) * int(m.group(4)) * v["launches"]
    classes = {}
    for name, fl in flops.items():
        pats = bench.gemm_class_patterns(name)
        ns = sum(float(r["TotalDurationNs"]) for r in rows if any(p in r["Name"] for p in pats))
        if ns <= 0:
            continue
        ms = ns / 1e6 / steps
        tf = fl / (ms * 1e-3) / 1e12
        classes[name] = dict(gflop_per_step=round(fl / 1e9, 1), kernel_ms_per_step=round(ms, 3), tflops=round(tf, 1),
                             frac=round(tf / bench.MFMA_BF16_PEAK_TFLOPS, 4))
    json.dump(dict(csv=a.csv_name, commit=a.commit, workload=a.workload, clips_per_gpu=a.clips, steps=steps, classes=classes), sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
