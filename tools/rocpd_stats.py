#!/usr/bin/env python3
"""rocprofv3's default output on this image is a rocpd SQLite database (<name>_results.db); this turns its kernel dispatches into the
`--kernel-trace --stats` summary the earlier rounds committed as CSV: Name, Calls, TotalDurationNs, AverageNs, Percentage, MinNs, MaxNs,
StdDev -- one row per kernel, longest total first.   usage: rocpd_stats.py results.db [out.csv]"""
import csv
import sqlite3
import statistics
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, duration from kernels").fetchall()
    by = {}
    for name, dur in rows:
        by.setdefault(name, []).append(float(dur))
    total = sum(sum(v) for v in by.values()) or 1.0
    out = csv.writer(open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout, quoting=csv.QUOTE_NONNUMERIC)
    out.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for name, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        out.writerow([name, len(v), int(sum(v)), round(sum(v) / len(v), 6), round(100.0 * sum(v) / total, 4), int(min(v)), int(max(v)),
                      round(statistics.pstdev(v), 6) if len(v) > 1 else 0.0])


if __name__ == "__main__":
    main()
