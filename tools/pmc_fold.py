#!/usr/bin/env python3
"""Fold a rocprofv3 --pmc counter_collection.csv: mean counter value per launch for kernels matching a substring.
usage: pmc_fold.py <dir> [substring]"""
import collections, csv, glob, sys
d, sub = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k)
    for c, vals in sorted(v.items()):
        print(f"    {c:28s} {sum(vals) / len(vals):16.1f}  (n={len(vals)})")
