#!/usr/bin/env python3
"""Two-stream picture of one training step from a rocprofv3 --kernel-trace CSV: per HIP queue the busy time, how long each queue runs
ALONE (the other one empty), and a coarse timeline (one line per 0.5 ms: the kernel that holds most of the slot on each queue).

    python tools/trace_streams.py <kernel_trace.csv> [slot_ms]
"""
import csv
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"void\s+(\w+)<(.*)>\(", name)
    if m:
        return f"{m.group(1)}<{m.group(2)[:18]}>"
    return name.split("(")[0][:34]


def union(iv):
    iv = sorted(iv)
    out = []
    for a, b in iv:
        if out and a <= out[-1][1]:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return out


def length(iv):
    return sum(b - a for a, b in iv)


def intersect(x, y):
    i = j = 0
    out = []
    while i < len(x) and j < len(y):
        a, b = max(x[i][0], y[j][0]), min(x[i][1], y[j][1])
        if a < b:
            out.append([a, b])
        if x[i][1] < y[j][1]:
            i += 1
        else:
            j += 1
    return out


def main():
    path = sys.argv[1]
    slot = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 0.5e6
    rows = []
    with open(path) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r.get("Queue_Id", 0) or 0)))
    rows.sort()
    ends = [i for i, r in enumerate(rows) if "adamw" in r[2]]
    seg = rows[ends[-2] + 1: ends[-1] + 1]
    t0, t1 = seg[0][0], max(r[1] for r in seg)
    queues = sorted(set(r[3] for r in seg))
    per_q = {q: union([(a, b) for a, b, _, qq in seg if qq == q]) for q in queues}
    print(f"step span {(t1 - t0) / 1e6:.2f} ms; queues {queues}")
    for q in queues:
        others = union([iv for qq in queues if qq != q for iv in per_q[qq]])
        both = intersect(per_q[q], others)
        print(f"  queue {q}: busy {length(per_q[q]) / 1e6:6.2f} ms, alone {(length(per_q[q]) - length(both)) / 1e6:6.2f} ms, "
              f"kernel time {sum(b - a for a, b, _, qq in seg if qq == q) / 1e6:6.2f} ms, launches {sum(1 for r in seg if r[3] == q)}")
    n = int((t1 - t0) / slot) + 1
    for k in range(n):
        a0, b0 = t0 + k * slot, t0 + (k + 1) * slot
        cells = []
        for q in queues:
            share = defaultdict(float)
            for a, b, name, qq in seg:
                if qq == q and a < b0 and b > a0:
                    share[short(name)] += min(b, b0) - max(a, a0)
            if share:
                nm, v = max(share.items(), key=lambda kv: kv[1])
                cells.append(f"{nm:34s} {sum(share.values()) / slot * 100:4.0f}%")
            else:
                cells.append(f"{'-':34s}    0%")
        print(f"  {k * slot / 1e6:5.1f} ms | " + " | ".join(cells))


if __name__ == "__main__":
    main()
