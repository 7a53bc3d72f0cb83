#!/usr/bin/env python3
"""Summarise tools/pmc_ntt_classes.sh: VALU instructions per element and pass of the three NTT kernels, by class.
usage: tools/pmc_ntt_classes_table.py gpurun_out/pmc_ntt_classes profiles/r05_ntt_valu_breakdown   (writes .txt and .json)"""
import collections
import csv
import json
import os
import sys

root, out = sys.argv[1], sys.argv[2]
KERNELS = ["k_ntt_pass<8, false, 5, true>", "k_ntt_pass<8, true, 5, true>", "k_lde_mid<7, 1, 5>"]
ELEMS = {KERNELS[0]: (1 << 24) * 96, KERNELS[1]: (1 << 23) * 96, KERNELS[2]: (1 << 23) * 96}   # elements a launch reads (full-width 96-column chunks)
PASSES = {KERNELS[0]: 1, KERNELS[1]: 1, KERNELS[2]: 2}                                        # radix steps a launch performs on them (the fused kernel: INTT-last + NTT-first)
BUILDS = ["shipped", "noarith", "NOCANON", "NOADDSUB", "NOPOW2", "NOMULW"]
# tools/valu_floor.py's model per element and radix-256 pass: 4 butterflies x (3 + 3), 2.1 shift twiddles x 4, 2 twiddle multiplies x (4 + 5), no data movement
FLOOR = {"butterfly add / sub": 24.0, "shift twiddles (2^e)": 8.4, "twiddle multiplies": 18.0, "canonical form": 0.0, "movement, addressing, LDS, stores": 0.0}


def short(name):
    for k in KERNELS:
        if k in name:
            return k
    return None


def read(build):
    res = {}
    p = os.path.join(root, build + "_sq", build + "_sq_counter_collection.csv")
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        k = short(r["Kernel_Name"])
        if k and int(r["Grid_Size"]) >= 50331648 // (2 if k != KERNELS[0] else 1):   # the 96-column chunks only
            acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in acc.items():
        res.setdefault(k, {})[c] = max(v)          # (the widest launches of a kernel: max filters the narrower last chunk)
    p = os.path.join(root, build + "_trace", build + "_trace_kernel_stats.csv")
    if os.path.exists(p):
        for r in csv.DictReader(open(p)):
            k = short(r["Name"])
            if k:
                res.setdefault(k, {})["avg_ms"] = float(r["AverageNs"]) / 1e6
    return res


D = {b: read(b) for b in BUILDS}
J = {"source": "tools/pmc_ntt_classes.sh: rocprofv3 --pmc SQ_INSTS_VALU of bench.py's default workload, shipped build and five builds with one class of the arithmetic compiled out", "kernels": {}}
lines = ["# NTT VALU instructions per element and radix step, by class (round 5; tools/pmc_ntt_classes.sh).  A class's count = shipped build's",
         "# SQ_INSTS_VALU minus the build's with that class compiled out (wrong results, same data flow); 'movement' = the no-arith build;",
         "# 'unattributed' = what the four class builds and the no-arith build do not add up to (the compiler schedules a build without a",
         "# class differently).  floor: tools/valu_floor.py's model.  Times: kernel trace of the same builds."]
for k in KERNELS:
    s = D["shipped"][k]
    per = lambda b: D[b][k]["SQ_INSTS_VALU"] * 64.0 / ELEMS[k] / PASSES[k]
    total = per("shipped")
    cls = collections.OrderedDict()
    cls["butterfly add / sub"] = total - per("NOADDSUB")
    cls["shift twiddles (2^e)"] = total - per("NOPOW2")
    cls["twiddle multiplies"] = total - per("NOMULW")
    cls["canonical form"] = total - per("NOCANON")
    cls["movement, addressing, LDS, stores"] = per("noarith")
    un = total - sum(cls.values())
    lines += ["", "== %s   (%d radix step%s per launch)" % (k, PASSES[k], "s" if PASSES[k] > 1 else ""),
              "%-40s %10s %10s %10s" % ("class", "instr/elem", "floor", "excess")]
    for name, v in cls.items():
        lines.append("%-40s %10.1f %10.1f %10.1f" % (name, v, FLOOR[name], v - FLOOR[name]))
    lines.append("%-40s %10.1f" % ("unattributed", un))
    lines.append("%-40s %10.1f %10.1f %10.1f" % ("total (measured, shipped build)", total, sum(FLOOR.values()), total - sum(FLOOR.values())))
    lines.append("launch ms: " + ", ".join("%s %.3f" % (b, D[b][k].get("avg_ms", float("nan"))) for b in BUILDS))
    J["kernels"][k] = {"valu_per_element_per_radix_step": total, "classes": {n: round(v, 2) for n, v in cls.items()}, "unattributed": round(un, 2),
                       "floor": FLOOR, "launch_ms": {b: D[b][k].get("avg_ms") for b in BUILDS}}
open(out + ".txt", "w").write("\n".join(lines) + "\n")
json.dump(J, open(out + ".json", "w"), indent=1)
print("\n".join(lines))
