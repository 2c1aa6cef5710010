#!/usr/bin/env python3
"""Sums rocprofv3 --pmc CSV rows per (kernel, counter) and prints the per-launch mean. usage: pmc_sum.py <dir>"""
import csv, glob, os, sys, collections
acc = collections.defaultdict(lambda: [0.0, set()])
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-60:]
        a = acc[(k, r["Counter_Name"])]
        a[0] += float(r["Counter_Value"]); a[1].add(r["Dispatch_Id"])
for (k, c), (v, d) in sorted(acc.items()):
    if "render" in k or "trace" in k:
        print("%-62s %-34s launches %3d  per launch %.4g" % (k, c, len(d), v / max(len(d), 1)))
