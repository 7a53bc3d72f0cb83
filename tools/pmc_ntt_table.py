#!/usr/bin/env python3
"""Summarise tools/pmc_ntt.sh's output (gpurun_out/pmc_ntt) into one table per NTT kernel: shipped library vs the
MI_NTT_NO_ARITH build (same loads, LDS round trips, barriers and stores; no field arithmetic).
usage: tools/pmc_ntt_table.py gpurun_out/pmc_ntt > profiles/r03_pmc_ntt.txt"""
import collections
import csv
import os
import sys

root = sys.argv[1]
KERNELS = ["k_ntt_pass<8, false, 5, true>", "k_ntt_pass<8, true, 5, true>", "k_lde_mid<7, 1, 5>"]
ROWS96 = {KERNELS[0]: (1 << 24, 2.0), KERNELS[1]: (1 << 23, 2.0), KERNELS[2]: (1 << 23, 3.0)}   # rows in, volumes of (rows_in x cols x 8 B) moved


def short(name):
    for k in KERNELS:
        if k in name:
            return k
    return None


def stats(variant):
    out = collections.defaultdict(dict)
    p = os.path.join(root, f"{variant}_trace", f"{variant}_trace_kernel_stats.csv")
    for r in csv.DictReader(open(p)):
        k = short(r["Name"])
        if k:
            out[k]["calls"] = int(r["Calls"])
            out[k]["avg_ms"] = float(r["AverageNs"]) / 1e6
    for pas in ("sq1", "sq2", "tcc1", "tcc2"):
        p = os.path.join(root, f"{variant}_{pas}", f"{variant}_{pas}_counter_collection.csv")
        if not os.path.exists(p):
            continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(p)):
            k = short(r["Kernel_Name"])
            if k and int(r["Grid_Size"]) >= 50331648:           # the 96-column chunks only (the last chunk of 665 = 6 x 96 + 89 is narrower)
                acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in acc.items():
            out[k][c] = sum(v) / len(v)
    return out


A, B = stats("arith"), stats("noarith")
print("# LDE diagnosis, round 3: bench.py default workload (2^23 x 665 -> 2^24, 96-column chunks), 1 x MI355X, rocprofv3 PMC passes")
print("# (tools/pmc_ntt.sh); shipped library vs MI_NTT_NO_ARITH build.  Counters are per launch (average over the full-width chunks);")
print("# SQ_* in quad-cycles summed over all waves / SIMDs, GRBM_GUI_ACTIVE summed over the 8 XCDs, FETCH_SIZE/WRITE_SIZE in KB as rocprofv3")
print("# reports them (FETCH_SIZE doubled per MI355X_MICROARCH.md: 128-byte requests tallied at 64 B).")
for k in KERNELS:
    a, b = A[k], B[k]
    rows, vols = ROWS96[k]
    elems = rows * 96
    gb = elems * 8 * vols / 1e9
    print(f"\n== {k}   ({rows} input rows x 96 columns per launch; {gb:.2f} GB moved by design)")
    print(f"{'':34s}{'arith':>16s}{'no-arith':>16s}")
    def row(label, fa, fb, fmt="{:16.3f}"):
        print(f"{label:34s}" + fmt.format(fa) + fmt.format(fb))
    row("launch ms (kernel trace)", a["avg_ms"], b["avg_ms"])
    row("GB/s moved by design", gb / a["avg_ms"] * 1e3, gb / b["avg_ms"] * 1e3, "{:16.0f}")
    for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVES"):
        if c in a:
            row(c, a[c], b.get(c, float("nan")), "{:16.4g}")
    if "SQ_INSTS_VALU" in a:
        row("VALU instr / element", a["SQ_INSTS_VALU"] * 64 / elems, b["SQ_INSTS_VALU"] * 64 / elems, "{:16.1f}")
    wc = "SQ_WAVE_CYCLES"
    if wc in a:
        for c, lab in (("SQ_ACTIVE_INST_ANY", "wave: issuing (any)"), ("SQ_ACTIVE_INST_VALU", "wave: issuing VALU"), ("SQ_WAIT_INST_ANY", "wave: issue-stalled"),
                       ("SQ_WAIT_ANY", "wave: parked (waitcnt/barrier)")):
            row(lab + " / wave cycles", a[c] / a[wc], b[c] / b[wc])
        # SIMD-level VALU occupancy: quad-cycles with a VALU instruction issuing / quad-cycles the 1024 SIMDs were up
        for lab, s in (("arith", a), ("noarith", b)):
            s["simd_quads"] = 1024 * s["GRBM_GUI_ACTIVE"] / 8 / 4
        row("VALU issue slots taken (1024 SIMDs)", a["SQ_ACTIVE_INST_VALU"] / a["simd_quads"], b["SQ_ACTIVE_INST_VALU"] / b["simd_quads"])
        row("launch ms if VALU slots were 100 %", a["avg_ms"] * a["SQ_ACTIVE_INST_VALU"] / a["simd_quads"], b["avg_ms"] * b["SQ_ACTIVE_INST_VALU"] / b["simd_quads"])
        row("waves resident per SIMD (avg)", a[wc] / a["simd_quads"], b[wc] / b["simd_quads"], "{:16.2f}")
    for c, lab in (("SQ_WAIT_INST_LDS", "wave: stalled on LDS issue"), ("SQ_ACTIVE_INST_LDS", "wave: issuing LDS"), ("SQ_ACTIVE_INST_VMEM", "wave: issuing VMEM")):
        if c in a and wc in a:
            row(lab + " / wave cycles", a[c] / a[wc], b[c] / b[wc], "{:16.4f}")
    if "SQ_LDS_BANK_CONFLICT" in a:
        row("SQ_LDS_BANK_CONFLICT (cycles)", a["SQ_LDS_BANK_CONFLICT"], b["SQ_LDS_BANK_CONFLICT"], "{:16.4g}")
    if "FETCH_SIZE" in a:
        row("HBM read GB (2 x FETCH_SIZE)", 2 * a["FETCH_SIZE"] * 1024 / 1e9, 2 * b["FETCH_SIZE"] * 1024 / 1e9)
    if "WRITE_SIZE" in a:
        row("HBM write GB (WRITE_SIZE)", a["WRITE_SIZE"] * 1024 / 1e9, b["WRITE_SIZE"] * 1024 / 1e9)
    if "TCC_EA0_RDREQ_sum" in a:
        row("TCC_EA0_RDREQ_sum", a["TCC_EA0_RDREQ_sum"], b["TCC_EA0_RDREQ_sum"], "{:16.4g}")
tot = lambda S: 14 * S[KERNELS[0]]["avg_ms"] + 14 * S[KERNELS[1]]["avg_ms"] + 7 * S[KERNELS[2]]["avg_ms"]
print(f"\nLDE per step (14 forward + 14 inverse passes + 7 fused middle passes, at the full-width chunk's time): arith {tot(A):.1f} ms, no-arith {tot(B):.1f} ms")
if "SQ_ACTIVE_INST_VALU" in A[KERNELS[0]]:
    fl = sum(n * A[k]["avg_ms"] * A[k]["SQ_ACTIVE_INST_VALU"] / A[k]["simd_quads"] for k, n in zip(KERNELS, (14, 14, 7)))
    print(f"VALU-issue floor at the current instruction count (every SIMD issuing a VALU instruction every quad-cycle): {fl:.1f} ms")
