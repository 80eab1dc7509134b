#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel name, mean of each counter."""
import csv, sys, collections
path = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
with open(path) as fh:
    for row in csv.DictReader(fh):
        name = row.get("Kernel_Name", "")
        if flt and flt not in name: continue
        acc[name[:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name, cs in acc.items():
    print(name)
    for c, v in sorted(cs.items()):
        print("   %-28s n=%-5d mean=%.6g" % (c, len(v), sum(v) / len(v)))
