#!/usr/bin/env python3
"""Top kernels of a rocprofv3 --kernel-trace --stats run (csv or sqlite output), per executed step.
usage: kstats.py <dir-or-file> <steps-executed> [top]"""
import csv, glob, os, sqlite3, sys
path, steps = sys.argv[1], float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
rows = []
cs = glob.glob(os.path.join(path, "**", "*kernel_stats.csv"), recursive=True) if os.path.isdir(path) else [path]
if cs and cs[0].endswith(".csv"):
    for r in csv.DictReader(open(cs[0])):
        rows.append((r["Name"], int(r["Calls"]), float(r["TotalDurationNs"])))
else:
    dbs = glob.glob(os.path.join(path, "**", "*.db"), recursive=True)
    db = sqlite3.connect(dbs[0]); cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]; sym = [t for t in tabs if "kernel_symbol" in t][0]
    for r in cur.execute(f"select s.kernel_name, count(*), sum(d.end-d.start) from {kd} d join {sym} s on d.kernel_id=s.id group by s.kernel_name"):
        rows.append(r)
rows.sort(key=lambda r: -r[2])
tot = sum(r[2] for r in rows)
print(f"{len(rows)} kernels, {sum(r[1] for r in rows) / steps:.0f} launches per step, {tot / steps / 1e3:.0f} us of kernel time per step")
for n, c, t in rows[:top]:
    print(f"  {n[:84]:84s} {c / steps:7.1f}/step  avg {t / c / 1e3:7.1f} us  {t / steps / 1e3:8.1f} us/step {100 * t / tot:5.1f}%")
