#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: mean counter value per kernel (last dispatches)."""
import csv, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for k, d in acc.items():
    if flt and flt not in k:
        continue
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
