#!/usr/bin/env python3
"""tools/pmc_leaf.sh's output -> profiles/r04_pmc_leaf.json (same fields as profiles/r03_pmc_leaf.json, which bench.py reads): per-launch
averages of k_linear_hash_rows_lines at the bench workload.  FETCH_SIZE is doubled (gfx950 tallies 128-byte requests at 64 B:
MI355X_MICROARCH.md), WRITE_SIZE as reported; both in KB of 1024 B.  The issue mix / class rates are the static ISA analysis and
microbenchmarks of round 2 (the kernel's code is unchanged), copied from the r03 file.
usage: tools/pmc_leaf_json.py gpurun_out/pmc_leaf profiles/r03_pmc_leaf.json > profiles/r04_pmc_leaf.json"""
import collections, csv, glob, json, os, sys
root, prev = sys.argv[1], json.load(open(sys.argv[2]))
K = "k_linear_hash_rows_lines"


def counters(d):
    f = glob.glob(os.path.join(root, d, "**", "*counter_collection.csv"), recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if K in r["Kernel_Name"] and int(r["Grid_Size"]) >= 1 << 22:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


sq, n = counters("sq")
fe, _ = counters("fetch")
wr, _ = counters("write")
rows, ncols = 1 << 24, 665
perms_waves = rows * ((ncols + 7) // 8) / 64.0
alg = 8.0 * rows * ncols + 32.0 * rows
rd, wrb = 2 * fe["FETCH_SIZE"] * 1024, wr["WRITE_SIZE"] * 1024
avg_ms = None
f = glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True)
for r in csv.DictReader(open(f[0])) if f else []:
    if K in r["Name"]:
        avg_ms = float(r["AverageNs"]) / 1e6
old = prev["kernels"][K]
wc = sq["SQ_WAVE_CYCLES"]
out = {"_how": "round 4 capture: tools/pmc_leaf.sh (three separate rocprofv3 --pmc passes of `python3 bench.py --steps 2 --warmup 1 --no-verify --no-cpu-baseline --no-genproof --pcie-steps 0` "
               "+ a kernel trace); per-launch averages of k_linear_hash_rows_lines<2> (%d launches each); FETCH_SIZE doubled (gfx950: 128-byte requests tallied at 64 B, MI355X_MICROARCH.md), "
               "WRITE_SIZE as reported; KB = 1024 B.  issue_mix / issue_rates: the static ISA analysis and microbenchmarks of round 2 (the kernel's code is unchanged)." % n.get("SQ_INSTS_VALU", 0),
       "config": "2^23 x 665 trace, LDE to 2^24, Merkle tree (bench.py default), 1 x MI355X, round 4, poseidon variant 2",
       "kernels": {K: {"hbm_bytes_per_launch": rd + wrb, "algorithmic_bytes_per_launch": alg, "ratio": (rd + wrb) / alg, "avg_launch_ms_kernel_trace": avg_ms,
                       "sq_insts_valu_per_launch": sq["SQ_INSTS_VALU"], "sq_active_inst_valu": sq["SQ_ACTIVE_INST_VALU"], "sq_active_inst_any": sq["SQ_ACTIVE_INST_ANY"],
                       "sq_wait_inst_any": sq["SQ_WAIT_INST_ANY"], "sq_wait_any": sq["SQ_WAIT_ANY"], "sq_wave_cycles": wc, "sq_busy_cycles": sq["SQ_BUSY_CYCLES"],
                       "grbm_gui_active_sum_over_8_xcds": sq["GRBM_GUI_ACTIVE"], "valu_instructions_per_permutation": sq["SQ_INSTS_VALU"] / perms_waves,
                       "wave_time_split": {"issuing": sq["SQ_ACTIVE_INST_ANY"] / wc, "issuing_valu": sq["SQ_ACTIVE_INST_VALU"] / wc,
                                           "issue_stalled_behind_other_waves": sq["SQ_WAIT_INST_ANY"] / wc, "parked_waitcnt_or_nop": sq["SQ_WAIT_ANY"] / wc},
                       "effective_clock_ghz": (sq["GRBM_GUI_ACTIVE"] / 8) / (avg_ms * 1e-3) / 1e9 if avg_ms else old.get("effective_clock_ghz"),
                       "issue_mix": old["issue_mix"], "issue_rates_wave_instr_per_s": old["issue_rates_wave_instr_per_s"],
                       "hbm_read_bytes_2x_FETCH_SIZE": rd, "hbm_write_bytes_WRITE_SIZE": wrb}}}
print(json.dumps(out, indent=1))
