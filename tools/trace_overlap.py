#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel-trace CSV: how much of the time two kernel families overlap."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    n = r["Kernel_Name"]
    fam = "gemm" if "gemm" in n else ("fill" if "fill" in n.lower() else None)
    if fam:
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), fam))
ev.sort()
tot = {"gemm": 0, "fill": 0}
for s, e, f in ev:
    tot[f] += e - s
# overlap: sweep
ov = 0
g = [(s, e) for s, e, f in ev if f == "gemm"]
fl = [(s, e) for s, e, f in ev if f == "fill"]
j = 0
for s, e in g:
    for s2, e2 in fl:
        if e2 <= s or s2 >= e:
            continue
        ov += min(e, e2) - max(s, s2)
print({k: v / 1e3 for k, v in tot.items()}, "overlap_us", ov / 1e3, "n", len(g), len(fl))
