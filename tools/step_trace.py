#!/usr/bin/env python3
"""One steady-state training step out of a rocprofv3 kernel trace, launch by launch.

usage: step_trace.py <dir with *kernel_trace.csv> [--step N] [--small US]
Steps are delimited by the LAST adamw_kernel launch of a step.  Prints the launches of step N (default: the middle
one) whose duration is below --small microseconds in order, with the name of the launch before them (to find out who
issues the fills / copies), and a per-name table for the whole step incl. the idle gaps between launches."""
import csv, glob, os, sys
from collections import defaultdict


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "")
    for a, b in (("void ", ""), ("at::native::", "at::")):
        n = n.replace(a, b)
    return n[:70]


def main():
    d = sys.argv[1]
    step_n = int(sys.argv[sys.argv.index("--step") + 1]) if "--step" in sys.argv else None
    small = float(sys.argv[sys.argv.index("--small") + 1]) if "--small" in sys.argv else 12.0
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    ends = [i for i, r in enumerate(rows) if "adamw" in r[2] and (i + 1 == len(rows) or "adamw" not in rows[i + 1][2])]
    if len(ends) < 3:
        print("fewer than 3 steps in the trace")
        return
    k = step_n if step_n is not None else len(ends) // 2
    lo, hi = ends[k - 1] + 1, ends[k] + 1
    step = rows[lo:hi]
    t0 = step[0][0]
    span = (step[-1][1] - t0) / 1e3
    busy = sum(e - s for s, e, _ in step) / 1e3
    print(f"step {k}: {len(step)} launches, span {span:.1f} us, kernel time {busy:.1f} us, gaps {span - busy:.1f} us")
    tab = defaultdict(lambda: [0, 0.0])
    for s, e, n in step:
        tab[short(n)][0] += 1
        tab[short(n)][1] += (e - s) / 1e3
    print("\nper kernel (this step):")
    for n, (c, t) in sorted(tab.items(), key=lambda kv: -kv[1][1]):
        print(f"  {t:9.1f} us  {c:4d} x {t / c:8.1f}  {n}")
    print(f"\nlaunches shorter than {small} us, in order (offset us, duration us, name  <- previous launch):")
    for i, (s, e, n) in enumerate(step):
        if (e - s) / 1e3 < small:
            prev = short(step[i - 1][2]) if i else "-"
            print(f"  {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:6.1f}  {short(n):70s} <- {prev[:40]}")


if __name__ == "__main__":
    main()
