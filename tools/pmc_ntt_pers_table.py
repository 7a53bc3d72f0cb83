#!/usr/bin/env python3
"""tools/pmc_ntt_pers.sh's output -> a table: the radix-256 passes of bench.py's LDE, default kernels (MI_NTT_PERSISTENT=0) beside the
persistent double-buffered form (=1).   usage: tools/pmc_ntt_pers_table.py gpurun_out/pmc_ntt_pers > profiles/r04_pmc_ntt_persistent.txt"""
import collections, csv, glob, os, sys
root = sys.argv[1]


def find(d, suffix):
    m = glob.glob(os.path.join(root, d, "**", "*" + suffix), recursive=True)
    return m[0] if m else None


def stats(p):
    out = collections.defaultdict(lambda: collections.defaultdict(float))
    f = find("p%d_trace" % p, "kernel_stats.csv")
    for r in csv.DictReader(open(f)):
        if "k_ntt_pass" in r["Name"] or "k_lde_mid" in r["Name"]:
            k = r["Name"].split("(")[0].replace("void ", "")
            out[k]["calls"] = int(r["Calls"]); out[k]["avg_ms"] = float(r["AverageNs"]) / 1e6; out[k]["total_ms"] = float(r["TotalDurationNs"]) / 1e6
    f = find("p%d_sq1" % p, "counter_collection.csv")
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_ntt_pass" in r["Kernel_Name"] or "k_lde_mid" in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in acc.items():
        out[k][c] = sum(v) / len(v)
    return out


print("# The persistent double-buffered radix-256 pass (csrc/ntt.hip k_ntt_pass_pers: one 512-thread workgroup per CU, two 64 KB tiles, the next tile")
print("# requested by LDS-DMA loads while the current one is computed) against the default passes, bench.py default workload, 1 x MI355X, rocprofv3.")
print("# Per launch, averaged over ALL launches of a kernel name (the narrower last chunk included); SQ_* in quad-cycles over all waves.")
for p in (0, 1):
    S = stats(p)
    print("\n== MI_NTT_PERSISTENT=%d" % p)
    tot = 0.0
    for k, s in sorted(S.items()):
        quads = 1024 * s["GRBM_GUI_ACTIVE"] / 8 / 4 if s["GRBM_GUI_ACTIVE"] else float("nan")
        tot += s["total_ms"]
        print("%-44s calls %3d  avg %7.3f ms  total %8.2f ms | VALU issue slots taken %.3f | wave parked %.3f issue-stalled %.3f issuing VALU %.3f | waves/SIMD %.2f | SQ_INSTS_VALU %.4g"
              % (k, s["calls"], s["avg_ms"], s["total_ms"], s["SQ_ACTIVE_INST_VALU"] / quads, s["SQ_WAIT_ANY"] / max(s["SQ_WAVE_CYCLES"], 1), s["SQ_WAIT_INST_ANY"] / max(s["SQ_WAVE_CYCLES"], 1),
                 s["SQ_ACTIVE_INST_VALU"] / max(s["SQ_WAVE_CYCLES"], 1), s["SQ_WAVE_CYCLES"] / quads, s["SQ_INSTS_VALU"]))
    print("NTT kernels, whole run (3 steps): %.1f ms" % tot)
