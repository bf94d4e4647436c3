#!/usr/bin/env python3
"""Kernel trace of a (graph-replayed) step: busy time, gaps between consecutive kernels, top kernels by time.
usage: trace_gaps.py <rocprofv3 output dir> [last_n_steps_marker_kernel_substring]"""
import csv, glob, sys, collections
d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
mark = sys.argv[2] if len(sys.argv) > 2 else "adamw"
idx = [i for i, r in enumerate(rows) if mark in r[2]]
# steps end at the last adamw launch of each step: take the window between the 3rd-last and last group boundaries
ends = [i for k, i in enumerate(idx) if k + 1 == len(idx) or idx[k + 1] - i > 5]
a, b = ends[-3] + 1, ends[-1] + 1
win = rows[a:b]
nsteps = 2
wall = (win[-1][1] - win[0][0]) / 1e3 / nsteps
busy = sum(e - s for s, e, _ in win) / 1e3 / nsteps
gaps = [win[i + 1][0] - win[i][1] for i in range(len(win) - 1)]
print(f"steps={nsteps} kernels/step={len(win) / nsteps:.0f} wall/step={wall:.1f} us busy/step={busy:.1f} us "
      f"idle/step={wall - busy:.1f} us  mean gap={sum(gaps) / len(gaps) / 1e3:.2f} us  gaps>3us: {sum(g > 3000 for g in gaps) / nsteps:.0f}/step")
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n in win:
    agg[n[:90]][0] += e - s; agg[n[:90]][1] += 1
for n, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:28]:
    print(f"{t / 1e3 / nsteps:8.1f} us/step  n={c / nsteps:5.1f}  avg={t / c / 1e3:6.1f} us  {n}")
