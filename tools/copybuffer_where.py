#!/usr/bin/env python3
"""Where do the `__amd_rocclr_copyBuffer` launches of a bench_starks.py run fall?  Reads the kernel TRACE of
`rocprofv3 --kernel-trace --stats -- python3 bench_starks.py --proofs 2 --check-rows 0` and places every copyBuffer launch against the
proofs: a proof spans from its first `k_linear_hash_rows_lines` launch (stage 1's first absorbed chunk) to its last FRI kernel; what
falls before the first proof is SETUP (bench_starks.py building its synthetic witness on the device and copying it to pageable host
memory through the runtime's staging buffers; the constants' upload in the Starks constructor).
usage: tools/copybuffer_where.py <dir with *_kernel_trace.csv> [out.json]"""
import csv
import glob
import json
import os
import sys

d = sys.argv[1]
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[0]
rows = list(csv.DictReader(open(f)))
key_s = "Start_Timestamp" if "Start_Timestamp" in rows[0] else "Start"
key_e = "End_Timestamp" if "End_Timestamp" in rows[0] else "End"
ev = sorted(((int(r[key_s]), int(r[key_e]), r["Kernel_Name"]) for r in rows))
t0 = ev[0][0]
leaf = [e for e in ev if "k_linear_hash_rows_lines" in e[2]]
fold = [e for e in ev if "k_fri_fold" in e[2]]
# proofs: a gap of more than 50 ms between leaf launches that follows an FRI fold starts a new proof
proofs = []
cur = None
for s, e, n in ev:
    if "k_linear_hash_rows_lines" in n and (cur is None or cur.get("fri_seen")):
        if cur:
            proofs.append(cur)
        cur = {"start": s, "end": e, "fri_seen": False}
    if cur:
        cur["end"] = max(cur["end"], e)
        if "k_fri_fold" in n:
            cur["fri_seen"] = True
if cur:
    proofs.append(cur)
cb = [e for e in ev if "copyBuffer" in e[2]]
inside = [0] * len(proofs)
ms_inside = [0.0] * len(proofs)
before, ms_before, between, ms_between = 0, 0.0, 0, 0.0
for s, e, n in cb:
    hit = False
    for i, p in enumerate(proofs):
        if p["start"] <= s <= p["end"]:
            inside[i] += 1
            ms_inside[i] += (e - s) / 1e6
            hit = True
    if not hit:
        if not proofs or s < proofs[0]["start"]:
            before += 1
            ms_before += (e - s) / 1e6
        else:
            between += 1
            ms_between += (e - s) / 1e6
out = {"trace": os.path.relpath(f), "copyBuffer_launches": len(cb), "copyBuffer_ms_total": sum((e - s) / 1e6 for s, e, n in cb),
       "proofs_found": len(proofs), "proof_ms": [(p["end"] - p["start"]) / 1e6 for p in proofs],
       "before_the_first_proof": {"launches": before, "ms": ms_before},
       "between_or_after_proofs": {"launches": between, "ms": ms_between},
       "inside_proofs": [{"launches": a, "ms": b} for a, b in zip(inside, ms_inside)]}
print(json.dumps(out, indent=1))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
