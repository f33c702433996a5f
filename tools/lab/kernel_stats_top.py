#!/usr/bin/env python3
"""Lab: top rows of a rocprofv3 *_kernel_stats.csv.   python tools/lab/kernel_stats_top.py <dir-or-file> [n]"""
import csv, glob, os, sys
p = sys.argv[1]
f = p if p.endswith(".csv") else glob.glob(os.path.join(p, "**", "*kernel_stats.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{f}: {len(rows)} kernels, total {tot / 1e6:.2f} ms")
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    print(f"{r['Name'][:100]:100s} calls {r['Calls']:>5s} avg {float(r['AverageNs']) / 1e3:8.1f} us {r['Percentage']:>6s} %")
