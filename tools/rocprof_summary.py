#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats kernel_stats.csv (names can be kilobytes long) into
a small CSV: name (truncated), calls, avg/min/max us, total ms, percentage."""
import csv
import sys

src, dst = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(src)))
with open(dst, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "AverageUs", "MinUs", "MaxUs", "TotalMs", "Percentage"])
    for r in rows:
        name = r["Name"]
        if len(name) > 110:
            name = name[:107] + "..."
        w.writerow([name, r["Calls"], f"{float(r['AverageNs']) / 1e3:.3f}", f"{float(r['MinNs']) / 1e3:.3f}",
                    f"{float(r['MaxNs']) / 1e3:.3f}", f"{float(r['TotalDurationNs']) / 1e6:.3f}", r["Percentage"]])
