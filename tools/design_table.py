#!/usr/bin/env python3
"""DESIGN.md section 4's table from the committed profile of the round: every GEMM shape class of one training step -> the schedule
that runs it -> launches x time -> rate -> fraction of the dense bf16 MFMA peak, plus the non-GEMM kernel classes with their HBM rates.

    python3 tools/design_table.py profiles/r06_bench_gemm_shapes.json profiles/r06_serial_sum.txt > /tmp/table.md

The GEMM rows come from the instrumented step of bench.py (HIP events around every launch, streams serialised); the schedule column
restates the selection rules of csrc/gemm.hip (pick_variant, pair_shape) and wavjepa_amd/engine.py (_row_form_pays, _pair_pays)."""
import json
import re
import sys

PEAK = 2500.0


def schedule(kind, epi, M, N, K, gather):
    tiles = -(-M // 256) * (-(-N // 256))
    if gather:
        return "gather form, 256x128 (dgrad) / 256x256 4-stage (wgrad), one tile per workgroup"
    if kind == "TT":
        return "split-K, 256x256 4-stage ring, fp32 atomics (ungrouped)"
    if kind == "NT":
        return "col-form B, 256x128, 2 workgroups/CU"
    items = -(-M // 256) * (N // 256 + (1 if N % 256 else 0))
    if K % 128 == 0 and items >= 256 and epi in ("BF16", "BIAS_GELU", "BIAS_GELU2", "CONV_GELU", "MUL_GELU_GRAD") and (N % 256 == 0 or N % 256 >= 128):
        return "persistent eight-phase 256x256x64" + (" + half-width items" if N % 256 == 128 else "")
    if K % 128 == 0 and N % 256 == 0:
        return "eight-phase 256x256x64, one tile per workgroup"     # (K-split pairs: off by default since round 6)
    return "256x128, 2 workgroups/CU"


def who(M, N, K, epi, gather=False):
    if gather:
        return "conv backward, active rows"
    t = {51200: "teacher", 823296: "conv L1", 411648: "conv L2", 205824: "conv L3", 102912: "conv L4", 51456: "conv L5"}.get(M)
    if t:
        return t
    return "predictor" if M > 30000 else "student"


def main():
    shapes = json.load(open(sys.argv[1]))
    rows = []
    for key, v in shapes.items():
        m = re.match(r"gemm_kernel<([NT])([NT]),(\w+)> M=(\d+) N=(\d+) K=(\d+)( gather)?", key)
        if not m:
            continue
        kind, epi = m.group(1) + m.group(2), m.group(3)
        M, N, K = int(m.group(4)), int(m.group(5)), int(m.group(6))
        rows.append((v["ms"], kind, epi, M, N, K, bool(m.group(7)), v))
    rows.sort(key=lambda r: -r[0])
    print("| GEMM shape class (who) | schedule | launches x us | TFLOP/s | of peak | bound |")
    print("|---|---|---|---|---|---|")
    shown = 0.0
    for ms, kind, epi, M, N, K, gather, v in rows:
        if ms < 0.2:
            continue
        shown += ms
        intensity = K / (2.0 if epi in ("BIAS_GELU2", "MUL_GELU_GRAD", "CONV_GELU") else 1.0)        # flops per output byte ~ 2K / (2 or 4)
        few = -(-M // 256) * -(-N // 256) < 200 and kind != "TT"
        if epi == "BIAS_GELU2" and K <= 512:
            bound = "epilogue VALU (erf-GELU and GELU': ~20 issue slots per output, 13 of an item's 23 us) + 2 bf16 stores per output"
        elif epi == "MUL_GELU_GRAD" and K <= 512:
            bound = "epilogue (a gelu' tile read + a product tile written per item, column sums) on 6 K tiles per item"
        elif K <= 512:
            bound = "operand delivery / epilogue (K = 384: 6 K tiles per item)"
        elif epi in ("BIAS_GELU", "BIAS_GELU2", "CONV_GELU") and not few:
            bound = "MFMA, then epilogue VALU with the matrix pipe idle (erf-GELU: 7 of an item's 27-30 us)"
        else:
            bound = "MFMA + tile count (fills < 256 CUs)" if few else "MFMA"
        print(f"| `{kind},{epi}` {M}x{N}x{K}{' gather' if gather else ''} ({who(M, N, K, epi, gather)}) | {schedule(kind, epi, M, N, K, gather)} | "
              f"{v['launches']} x {v['us_per_launch']:.0f} | {v['tflops']:.0f} | {v['tflops'] / PEAK:.2f} | {bound} |")
    rest = sum(r[0] for r in rows) - shown
    print(f"| {sum(1 for r in rows if r[0] < 0.2)} smaller shape classes (mappers, conv tails, bottom-layer `ADD_F32` dgrads) | as above | {rest:.2f} ms in all | | | |")
    if len(sys.argv) > 2:
        print()
        print("| kernel class (serial kernel-time sum) | ms/step | launches |")
        print("|---|---|---|")
        for ln in open(sys.argv[2]):
            m = re.match(r"^(\S.*?)\s{2,}(\d+\.\d+)\s+(\d+\.\d)\s*$", ln.rstrip())
            if m and not ln.startswith("#") and not ln.startswith("class"):
                print(f"| {m.group(1)} | {m.group(2)} | {m.group(3)} |")
            if ln.startswith("# kernels"):
                break


if __name__ == "__main__":
    main()
