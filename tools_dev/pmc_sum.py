#!/usr/bin/env python3
"""Developer tool: mean per launch of every PMC counter per kernel (summed over the chip) from a
rocprofv3 --pmc output directory.  usage: pmc_sum.py <dir> [substring of the kernel name ...]"""
import collections
import csv
import glob
import sys


def main():
    d, pats = sys.argv[1], sys.argv[2:]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if pats and not any(p in n for p in pats):
                continue
            n = n.replace("void ", "").replace("(anonymous namespace)::", "").replace("blh::", "").split("(")[0][:90]
            acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for n in sorted(acc):
        print(n)
        for c in sorted(acc[n]):
            v = acc[n][c]
            print("   %-28s %14.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))


if __name__ == "__main__":
    main()
