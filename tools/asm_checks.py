#!/usr/bin/env python3
"""Static checks on the gfx950 assembly of csrc/gemm_persist.hip (runs in the CPU suite; hipcc cross-compiles without a GPU).

The persistent GEMM pulls its next tile index with a RETURNING atomic written in inline asm, so that hipcc's waitcnt pass does not
see the result register (a tracked result is waited for with vmcnt(0), which drains the LDS-DMA ring).  The price: nothing in the
compiler stops it from copying or spilling that register before the value has landed.  This script proves, per build, that between
every `global_atomic_add vN, ...` and the `ds_write_b32 ..., vN` that consumes it no other instruction names vN, that the
kernels do not touch scratch inside the K loop (a scratch access is a vmcnt(0) as well), and that the main loop has no vmcnt(0).
"""
from __future__ import annotations

import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "wavjepa_amd", "csrc", "gemm_persist.hip")


def assembly(src: str = SRC) -> str:
    hipcc = "/opt/rocm/bin/hipcc"
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-S", "--cuda-device-only", src, "-o", out],
                       check=True, capture_output=True)
        return open(out).read()


def kernels(asm: str):
    cur, name = None, None
    for line in asm.splitlines():
        m = re.match(r"^(_ZN\S*gemm_persist_kernel\S*):", line)
        if m:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            cur.append(line)
            if "s_endpgm" in line:
                yield name, cur
                cur = None


def uses(line: str, reg: int) -> bool:
    code = line.split(";")[0]
    if re.search(rf"\bv{reg}\b", code):
        return True
    for m in re.finditer(r"v\[(\d+):(\d+)\]", code):
        if int(m.group(1)) <= reg <= int(m.group(2)):
            return True
    return False


def check(asm: str):
    problems, n_pulls = [], 0
    for name, lines in kernels(asm):
        i = 0
        while i < len(lines):
            m = re.search(r"global_atomic_add v(\d+),", lines[i])
            if not m:
                i += 1
                continue
            n_pulls += 1
            reg = int(m.group(1))
            j = i + 1
            found = False
            while j < len(lines):
                if re.search(rf"ds_write_b32 v\d+, v{reg}\b", lines[j]):
                    found = True
                    break
                if re.search(r"global_atomic_add v", lines[j]):
                    # the loop's pull follows the prologue's in program text: the prologue's consumer must have come first
                    break
                if uses(lines[j], reg) and not lines[j].strip().startswith(";"):
                    problems.append(f"{name}: line {j}: `{lines[j].strip()}` touches v{reg} between the pull and its consumer")
                j += 1
            if not found:
                problems.append(f"{name}: pull into v{reg} at line {i} has no ds_write_b32 consumer before the next pull")
            i += 1
        body = "\n".join(lines)
        if re.search(r"scratch_(load|store)", body):
            # allowed only outside the K loop: report where
            for k, l in enumerate(lines):
                if re.search(r"scratch_(load|store)", l):
                    problems.append(f"{name}: scratch access at line {k}: `{l.strip()}`")
    return n_pulls, problems


if __name__ == "__main__":
    n, probs = check(assembly(sys.argv[1] if len(sys.argv) > 1 else SRC))
    print(f"{n} pulls checked")
    for p in probs:
        print("PROBLEM:", p)
    sys.exit(1 if probs or n == 0 else 0)
