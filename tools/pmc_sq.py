#!/usr/bin/env python3
"""Summarise SQ counters of a rocprofv3 --pmc run per kernel: python tools/pmc_sq.py <dir> [<dir> ...]"""
import csv, glob, os, re, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
FILTER = [a[2:] for a in sys.argv[1:] if a.startswith("k=")] or ["gemm3"]      # k=<substring> selects kernels (default: the GEMM)
for d in [a for a in sys.argv[1:] if not a.startswith("k=")]:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
            k = re.sub(r"\(.*", "", k).replace("void ", "") + f" grid={r['Grid_Size']}"
            a = acc[k][r["Counter_Name"]]
            a[0] += 1; a[1] += float(r["Counter_Value"])
for k, cs in acc.items():
    if not any(t in k for t in FILTER):
        continue
    print(k)
    w = cs.get("SQ_WAVE_CYCLES", [1, 1.0])
    wc = w[1] / w[0]
    for c, (n, v) in sorted(cs.items()):
        print(f"    {c:32s} {v / n:16.0f}   {v / n / wc:7.3f} of WAVE_CYCLES")
