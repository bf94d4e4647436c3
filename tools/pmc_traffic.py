#!/usr/bin/env python3
"""Fold two rocprofv3 PMC passes (--pmc FETCH_SIZE and --pmc WRITE_SIZE, counter_collection CSVs)
into per-kernel HBM traffic per launch.  gfx950: FETCH_SIZE counts wide coalesced reads at half
their size (MI355X_MICROARCH.md), so bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB.

The JSON is stamped with the fingerprint of csrc/ (the same one bench.py computes): bench.py only reports the
traffic figure as current while the kernel sources are the ones the passes were taken on.

usage: pmc_traffic.py <fetch_dir> <write_dir> <out.json> <out.txt> [config] [dtype] [steps_profiled]
(steps_profiled is only the fallback: the number of executed steps is read off the cross-entropy kernel's launch count)

The JSON also lists EVERY kernel of the profiled run (bytes per launch, launches per step), from which bench.py
computes `step_hbm`: the HBM bytes one training step moves, against the 8 TB/s peak."""
import csv, glob, hashlib, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha16():
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "focused-attention-vit_amd", "csrc", "*"))):
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]

FAMILY = {  # bench.py GEMM family key -> substring(s) identifying the kernel instantiation
    "bf16_KM_obf16": ("gemm_bf16_p4_kernelILb1ELb0EDF16b", "gemm_bf16_p4_kernel<true, false, __bf16>"),
    "bf16_KK_obf16": ("gemm_bf16_p4_kernelILb1ELb1EDF16b", "gemm_bf16_p4_kernel<true, true, __bf16>"),
    "bf16_KK_of32": ("gemm_bf16_p4_kernel<true, true, float>", "gemm_bf16_p4_kernelILb1ELb1EfE"),
    "bf16_MM_of32_grouped": ("gemm_bf16_p4_grouped_tn_kernel",),
}


def load(d, counter):
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"]
            a = acc.setdefault(k, [0.0, 0])
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    return acc


def main():
    fd, wd, oj, ot = sys.argv[1:5]
    config = sys.argv[5] if len(sys.argv) > 5 else "cfg2"
    dtype = sys.argv[6] if len(sys.argv) > 6 else "bf16"
    steps = int(sys.argv[7]) if len(sys.argv) > 7 else 3
    fe, wr = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
    # steps the profiled process actually EXECUTED: the cross-entropy kernel runs exactly once per step, also in the
    # eager warm-up steps that train.GraphedStep runs before capturing (the graph-replayed configurations execute
    # 3 + warmup + steps of them; counting `--steps 2 --warmup 1` as 3 doubled every per-step figure of cfg1/3/5)
    ce = [v[1] for k, v in fe.items() if "cross_entropy_kernel" in k]
    if ce:
        steps = ce[0]
    rows = []
    for k in fe:
        n = fe[k][1]
        f = fe[k][0] / n
        w = wr.get(k, [0.0, 1])[0] / max(1, wr.get(k, [0.0, 1])[1])
        rows.append((k, n, f, w))
    rows.sort(key=lambda r: -(2 * r[2] + r[3]) * r[1])
    fam = {}
    for key, pats in FAMILY.items():
        for k, n, f, w in rows:
            if any(p in k for p in pats):
                fam[key] = {"kernel": k[:90], "hbm_bytes_per_launch": int((2 * f + w) * 1024), "fetch_kib": f,
                            "write_kib": w, "launches_sampled": n}
                break
    try:
        head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
    except Exception:
        head = None
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) -- python bench.py "
                         "--steps 2 --warmup 1; bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB (gfx950 FETCH_SIZE halves "
                         "wide coalesced reads, MI355X_MICROARCH.md)", "config": config, "dtype": dtype,
               "csrc_sha16": csrc_sha16(), "git_head_when_folded": head, "families": fam,
               "steps_profiled": steps,
               "kernels": [{"kernel": k[:120], "hbm_bytes_per_launch": int((2 * f + w) * 1024),
                            "launches_per_step": round(n / steps, 3)} for k, n, f, w in rows],
               "hbm_bytes_per_step": int(sum((2 * f + w) * 1024 * n for k, n, f, w in rows) / steps)},
              open(oj, "w"), indent=1)
    with open(ot, "w") as o:
        o.write("HBM traffic per launch from rocprofv3 PMC (separate passes: --pmc FETCH_SIZE / --pmc WRITE_SIZE) over "
                "`python bench.py --steps 2 --warmup 1`\nunits: counter value is KiB; per MI355X_MICROARCH.md FETCH_SIZE "
                "under-reports wide coalesced reads by 2x on gfx950 -> corrected = 2*FETCH\n")
        o.write(f"{'kernel':82s} {'launches':>8s} {'FETCH MiB':>10s} {'2xFETCH MiB':>12s} {'WRITE MiB':>10s}\n")
        for k, n, f, w in rows[:32]:
            o.write(f"{k[:80]:82s} {n:8d} {f / 1024:10.1f} {2 * f / 1024:12.1f} {w / 1024:10.1f}\n")


if __name__ == "__main__":
    main()
