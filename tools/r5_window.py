#!/usr/bin/env python3
"""Every kernel of one training step between two points of its timeline (ms from the step's first kernel), per HIP queue:
   python3 tools/r5_window.py <kernel_trace.csv> <from_ms> <to_ms>"""
import csv
import re
import sys

rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r.get("Queue_Id", 0) or 0)))
rows.sort()
ends = [i for i, r in enumerate(rows) if "adamw" in r[2]]
seg = rows[ends[-2] + 1: ends[-1] + 1]
t0 = seg[0][0]
lo, hi = float(sys.argv[2]), float(sys.argv[3])
for a, b, n, q in seg:
    if lo <= (a - t0) / 1e6 <= hi:
        n = re.sub(r"\(anonymous namespace\)::|void ", "", n).split("(")[0][:60]
        print(f"q{q} {(a - t0) / 1e6:8.3f} -> {(b - t0) / 1e6:8.3f} ms ({(b - a) / 1e3:7.1f} us)  {n}")
