#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel (name, grid) average counter value.
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB-like units of 1 KB; see MI355X_MICROARCH.md (HBM)."""
import csv, collections, sys
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    agg[(r["Kernel_Name"].split("(")[0], r["Grid_Size"], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k in sorted(agg, key=lambda k: -sum(agg[k]))[: int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    v = agg[k]
    print(f"{k[0]} grid={k[1]} {k[2]}: calls={len(v)} avg={sum(v)/len(v):.1f} KB  (= {sum(v)/len(v)*1024/1e9:.3f} GB)")
