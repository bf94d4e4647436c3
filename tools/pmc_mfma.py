#!/usr/bin/env python3
"""Fold a rocprofv3 SQ/GRBM PMC pass (over `bench.py --steps 2 --warmup 1`, or over tools/gemm_bench.py) into per-kernel
MFMA utilisation and wave-state shares of the GEMM kernels.  usage: pmc_mfma.py <pmc_dir> <out.txt>
  MFMA util      = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8
                   (rocprofv3 sums GRBM_GUI_ACTIVE over the 8 XCDs, MI355X_MICROARCH.md)
  wave states    = SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY as shares of SQ_WAVE_CYCLES"""
import csv, glob, os, sys
d, out = sys.argv[1:3]
acc = {}
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemm" not in k:
            continue
        a = acc.setdefault(k, {})
        c = a.setdefault(r["Counter_Name"], [0.0, 0])
        c[0] += float(r["Counter_Value"]); c[1] += 1
with open(out, "w") as o:
    o.write(__doc__.split("usage")[0].strip() + "\n\n")
    o.write(f"{'kernel':70s} {'launches':>8s} {'MFMA util':>10s} {'wait_any':>9s} {'wait_inst':>10s} {'active':>7s} {'LDS conflict/active':>20s}\n")
    for k, a in sorted(acc.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", [0, 1])[0]):
        g = lambda n: a.get(n, [0.0, 1])[0] / max(1, a.get(n, [0.0, 1])[1])
        cyc = g("GRBM_GUI_ACTIVE") / 8.0
        if cyc <= 0:
            continue
        util = g("SQ_VALU_MFMA_BUSY_CYCLES") / (1024.0 * cyc)
        wc = max(1.0, g("SQ_WAVE_CYCLES"))
        lds = g("SQ_LDS_BANK_CONFLICT") / max(1.0, g("SQ_LDS_IDX_ACTIVE"))
        n = a.get("GRBM_GUI_ACTIVE", [0, 0])[1]
        o.write(f"{k[:68]:70s} {n:8d} {util:10.3f} {g('SQ_WAIT_ANY') / wc:9.3f} {g('SQ_WAIT_INST_ANY') / wc:10.3f} "
                f"{g('SQ_ACTIVE_INST_ANY') / wc:7.3f} {lds:20.3f}\n")
print(open(out).read())
