#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950).

    cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_fetch -o runc --output-format csv -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_write -o runc --output-format csv -- python3 bench.py ...
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_pmc_traffic.json

Corrections as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes: both counters are in KiB; on gfx950 FETCH_SIZE
reports half of the bytes of wide coalesced reads, so  hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024  per launch.
Only this library's kernels are kept (torch fill / RNG kernels of the harness are dropped)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "")
    m = re.match(r"(?:void\s+)?([\w:]+(?:<[^(]*>)?)\(", name)
    return (m.group(1) if m else name)[:90]


def collect(directory: str, counter: str):
    out = defaultdict(lambda: [0, 0.0])
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] != counter:
                    continue
                k = short(r["Kernel_Name"])
                out[k][0] += 1
                out[k][1] += float(r["Counter_Value"])
    return out


def main():
    fetch_dir, write_dir, dst = sys.argv[1:4]
    fetch, write = collect(fetch_dir, "FETCH_SIZE"), collect(write_dir, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        if k.startswith(("at::", "__amd", "void at::")) or "at::native" in k:
            continue
        nf, f = fetch.get(k, [0, 0.0])
        nw, w = write.get(k, [0, 0.0])
        n = max(nf, nw)
        if n == 0:
            continue
        fk, wk = (f / nf if nf else 0.0), (w / nw if nw else 0.0)
        kernels[k] = dict(launches=n, fetch_kib_raw=round(fk, 1), write_kib=round(wk, 1),
                          hbm_bytes_per_launch=int((2 * fk + wk) * 1024))
    note = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only), averaged per launch; "
            "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE half-count correction, MI355X_MICROARCH.md)")
    with open(dst, "w") as fh:
        json.dump(dict(note=note, kernels=kernels), fh, indent=1)
    for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:25]:
        print(f"{k:70s} x{v['launches']:5d}  {v['hbm_bytes_per_launch'] / 1e6:10.1f} MB/launch")


if __name__ == "__main__":
    main()
