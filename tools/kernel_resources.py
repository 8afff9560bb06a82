#!/usr/bin/env python3
"""Registers / LDS / scratch of every gfx950 kernel of a csrc/*.hip source (hipcc -S --cuda-device-only, no GPU needed):
    python tools/kernel_resources.py norm.hip [misc.hip ...]
What decides co-residency: a persistent GEMM workgroup holds 2 x 224 VGPRs per SIMD and 150 KB of LDS, so a kernel of the other stream
shares its CU only at <= 64 VGPRs and <= ~10 KB of LDS."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def resources(src: str):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-S", "--cuda-device-only",
                        src, "-o", out] + sys.argv[1:0], check=True, capture_output=True)
        asm = open(out).read()
    rows = []
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", asm, re.S):
        name, body = m.group(1), m.group(2)
        get = lambda k: (re.search(rf"\.amdhsa_{k} (\S+)", body) or [None, "?"])[1]
        rows.append((name, get("next_free_vgpr"), get("accum_offset"), get("next_free_sgpr"), get("group_segment_fixed_size"), get("private_segment_fixed_size")))
    return rows


if __name__ == "__main__":
    for f in sys.argv[1:]:
        src = f if os.path.exists(f) else os.path.join(ROOT, "wavjepa_amd", "csrc", f)
        print(f"== {os.path.basename(src)}")
        for name, vg, acc, sg, lds, scr in resources(src):
            short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            short = re.sub(r"\(anonymous namespace\)::", "", short)[:90]
            print(f"  vgpr {vg:>4}  sgpr {sg:>4}  lds {lds:>7}  scratch {scr:>4}  {short}")
