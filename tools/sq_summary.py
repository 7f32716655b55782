#!/usr/bin/env python
"""Sum rocprofv3 --pmc SQ_* counters per kernel: sq_summary.py <counter_collection.csv>... <kernel substring>"""
import csv, sys, collections
kernel = sys.argv[-1]
acc, cnt = collections.defaultdict(float), collections.defaultdict(int)
for path in sys.argv[1:-1]:
    for r in csv.DictReader(open(path)):
        if kernel in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
for k in sorted(acc):
    print(f"{k:28s} launches {cnt[k]:3d}  mean/launch {acc[k] / cnt[k]:16.0f}")
