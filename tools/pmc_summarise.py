#!/usr/bin/env python3
"""Mean of each PMC counter over the dispatches of kernels matching a substring.  usage: pmc_summarise.py <dir> <substr>"""
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print(f"{k:32s} mean {sum(v) / len(v):16.1f}   (n={len(v)})")
