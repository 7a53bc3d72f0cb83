#!/usr/bin/env python3
"""tools/pmc_chunk.sh's output -> a table per constraint step (profiles/r04_pmc_chelpers.txt): for the generated kernels (chelpers_chunk),
the operand copies (k_chp_transpose) and the linear kernel of ONE proof at zkEVM size: launches, VALU instructions per row, how a wave's
cycles split (issuing VALU / stalled behind other waves' issue / parked on a wait counter), HBM bytes (FETCH_SIZE doubled: gfx950 tallies
128-byte requests at 64 B; KB of 1024 B).  Steps are told apart by dispatch order (the leaf-hash launches of the commits lie between them).
usage: tools/pmc_chunk_table.py gpurun_out/pmc_chunk"""
import collections, csv, glob, os, sys
root = sys.argv[1]
STEPS = ["step2prev", "step3prev", "step3", "step42ns", "step52ns"]
ROWS = {"step2prev": 1 << 23, "step3prev": 1 << 23, "step3": 1 << 23, "step42ns": 1 << 24, "step52ns": 1 << 24}


def load(d):
    f = glob.glob(os.path.join(root, d, "**", "*counter_collection.csv"), recursive=True)[0]
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        k = int(r["Dispatch_Id"])
        e = disp.setdefault(k, {"name": r["Kernel_Name"], "c": {}})
        e["c"][r["Counter_Name"]] = e["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    out, step, seen = [], -1, False
    for k in sorted(disp):
        n = disp[k]["name"]
        kind = "chunk" if n.startswith("chelpers_chunk") else "copy" if "k_chp_transpose" in n else "linear" if "k_chp_linear" in n else None
        if kind is None:
            if any(x in n for x in ("k_linear_hash_rows", "k_ntt_pass", "k_evmap", "h1h2", "z_blocks")):
                seen = False
            continue
        if kind == "copy" and step < 0 and not any(o[1] == "chunk" for o in out):
            pass
        if not seen:
            # a new run of constraint kernels begins; runs without generated kernels (the witness's tile-major write) are dropped later
            step += 1
            seen = True
        out.append((step, kind, disp[k]["c"]))
    # keep the runs that contain generated or linear kernels, renumbered
    runs = sorted({s for s, kind, _ in out if kind in ("chunk", "linear")})
    return [(runs.index(s), kind, c) for s, kind, c in out if s in runs]


sq, fe, wr = load("sq"), load("fetch"), load("write")
print("%-10s %-7s %8s %14s %9s %9s %9s %12s %12s" % ("step", "kernel", "launches", "VALU per row", "issuing", "stalled", "parked", "HBM read GB", "HBM write GB"))
for si, st in enumerate(STEPS):
    for kind in ("chunk", "copy", "linear"):
        a = [c for s, k, c in sq if s == si and k == kind]
        if not a:
            continue
        tot = collections.Counter()
        for c in a:
            tot.update(c)
        rd = 2 * 1024 * sum(c.get("FETCH_SIZE", 0) for s, k, c in fe if s == si and k == kind)
        wb = 1024 * sum(c.get("WRITE_SIZE", 0) for s, k, c in wr if s == si and k == kind)
        wc = tot["SQ_WAVE_CYCLES"] or 1.0
        print("%-10s %-7s %8d %14.0f %9.3f %9.3f %9.3f %12.1f %12.1f" % (st, kind, len(a), tot["SQ_INSTS_VALU"] * 64.0 / ROWS[st], tot["SQ_ACTIVE_INST_VALU"] / wc,
                                                                      tot["SQ_WAIT_INST_ANY"] / wc, tot["SQ_WAIT_ANY"] / wc, rd / 1e9, wb / 1e9))
