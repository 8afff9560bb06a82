#!/usr/bin/env python3
"""Timeline summary of a rocprofv3 --kernel-trace CSV: per training step (delimited by the wj adamw kernel), the span,
the time at least one kernel is running (busy), the idle remainder, and the per-kernel busy share.

    python tools/trace_gaps.py gpurun_out/prof/<pid>_kernel_trace.csv [--steps 3]
"""
import csv
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"void\s+(\w+)<(.*)>\(", name)
    if m:
        return f"{m.group(1)}<{m.group(2)[:40]}>"
    return name.split("(")[0][:60]


def main():
    path = sys.argv[1]
    rows = []
    with open(path) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r.get("Queue_Id", 0) or 0)))
    rows.sort()
    ends = [i for i, r in enumerate(rows) if "adamw" in r[2]]
    if len(ends) < 3:
        print("fewer than 3 optimiser steps in the trace")
        return
    # steady-state steps: between consecutive adamw launches, skipping the first
    for s in range(max(1, len(ends) - 4), len(ends)):
        seg = rows[ends[s - 1] + 1: ends[s] + 1]
        t0, t1 = seg[0][0], max(r[1] for r in seg)
        busy, cur_s, cur_e = 0, None, None
        for a, b, _, _ in seg:
            if cur_e is None or a > cur_e:
                if cur_e is not None:
                    busy += cur_e - cur_s
                cur_s, cur_e = a, b
            else:
                cur_e = max(cur_e, b)
        busy += cur_e - cur_s
        ksum = sum(b - a for a, b, _, _ in seg)
        gaps = sorted(((seg[i + 1][0] - max(r[1] for r in seg[:i + 1][-8:]), short(seg[i][2]), short(seg[i + 1][2])) for i in range(len(seg) - 1)),
                      reverse=True)[:6]
        print(f"step {s}: span {(t1 - t0) / 1e6:.2f} ms  busy {busy / 1e6:.2f} ms  idle {(t1 - t0 - busy) / 1e6:.2f} ms  "
              f"sum of kernels {ksum / 1e6:.2f} ms  launches {len(seg)}  queues {sorted(set(r[3] for r in seg))}")
        for g, a, b in gaps:
            if g > 0:
                print(f"    gap {g / 1e3:8.1f} us  after {a}  before {b}")
    per = defaultdict(float)
    seg = rows[ends[-2] + 1: ends[-1] + 1]
    for a, b, n, _ in seg:
        per[short(n)] += (b - a) / 1e6
    for k, v in sorted(per.items(), key=lambda kv: -kv[1])[:25]:
        print(f"  {v:8.3f} ms  {k}")


if __name__ == "__main__":
    main()
