#!/usr/bin/env python3
"""Static screen of hipcc's gfx950 assembly for one hazard the compiler's recogniser was seen to miss: a VALU (non-MFMA) read
of an MFMA result fewer than `need` wait states after the MFMA when a BRANCH lies between the two (round 2: a wave-uniform
branch after the score MFMAs of attn_fwd left one s_nop before v_max3_f32 and the kernel returned run-dependent sums).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o /tmp/k.s wavjepa_amd/csrc/attention.hip
    python tools/mfma_hazard_scan.py /tmp/k.s

Follows fall-through and taken paths from every v_mfma, counting one wait state per instruction (+n for s_nop n), stops a path at
`need` states, at a wait on memory counters (s_waitcnt with a pending operation is far longer than the hazard window) or at the
first reader."""
import re
import sys

need = 8          # 16x16x32 bf16 = 8 passes: hipcc itself leaves s_nop 7 (8 wait states) in straight-line code
src = open(sys.argv[1]).read().split("\n")
ins, labels = [], {}
for ln, text in enumerate(src, 1):
    t = text.strip()
    if not t or t.startswith(";") or t.startswith("//"):
        continue
    m = re.match(r"^([.\w$]+):", t)
    if m:
        labels[m.group(1)] = len(ins)
        continue
    if t.startswith("."):
        continue
    ins.append((ln, t.split(";")[0].strip()))


def regs(tok):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


flagged = 0
for idx, (ln, t) in enumerate(ins):
    if not t.startswith("v_mfma"):
        continue
    ops = t.split(None, 1)[1].split(",")
    dst = regs(ops[0])
    stack, seen = [(idx + 1, 0, False)], set()
    while stack:
        j, states, branched = stack.pop()
        while j < len(ins) and states < need:
            if (j, branched) in seen:
                break
            seen.add((j, branched))
            l2, u = ins[j]
            op = u.split()[0]
            if op.startswith("s_waitcnt") or op == "s_barrier" or op.startswith("s_endpgm"):
                break
            if op in ("s_branch",):
                tgt = u.split()[1]
                j, branched = labels.get(tgt, len(ins)), True
                states += 1
                continue
            if op.startswith("s_cbranch"):
                tgt = u.split()[1]
                if tgt in labels:
                    stack.append((labels[tgt], states + 1, True))
                states += 1
                j += 1
                continue
            if op == "s_nop":
                states += int(u.split()[1]) + 1
                j += 1
                continue
            if op.startswith("v_") and not op.startswith("v_mfma"):
                srcs = u.split(None, 1)[1].split(",")[1:] if "," in u else []
                if branched and regs(",".join(srcs)) & dst:
                    print(f"line {ln}: {t}\n   -> line {l2} ({states} wait states, across a branch): {u}")
                    flagged += 1
                    states = need
                    break
                if regs(",".join(srcs)) & dst:
                    break          # straight-line reader: hipcc's recogniser covers it
                if regs(u.split(None, 1)[1].split(",")[0]) & dst:
                    break          # overwritten
            states += 1
            j += 1
print("flagged:", flagged)
sys.exit(1 if flagged else 0)
