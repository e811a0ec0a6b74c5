#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats CSV (per-kernel calls / avg / share)."""
import csv, glob, sys
d = sys.argv[1]
f = (glob.glob(d + '/*/*_kernel_stats.csv') + glob.glob(d + '/*_kernel_stats.csv'))[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("total GPU time %.1f ms" % (tot / 1e6))
for r in rows:
    n = r['Name'].replace('blh::', '').replace('void ', '').split('(')[0]
    if float(r['TotalDurationNs']) / tot < 0.002: continue
    print("%-60s calls %5s avg %8.1f us  %5.1f%%" % (n[:60], r['Calls'], float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
