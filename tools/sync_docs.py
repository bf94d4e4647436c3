#!/usr/bin/env python3
"""Rewrite the numbers that DESIGN.md and profiles/README.md quote from the committed round-3 artefacts
(profiles/r03_bench_kernel_stats_cfg2.csv, r03_pmc_hbm_traffic_*.json, r03_bench_default_run.json,
r03_bench_unprofiled_*.json), so that the prose cannot drift from the files it cites.  Run after
`tools/collect_profiles.sh r03` + copying the final bench lines into profiles/."""
import csv, json, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda *a: os.path.join(ROOT, *a)


def line(path):
    return json.loads(open(path).read().strip().split("\n")[-1])


def main():
    L = lambda n: line(P("profiles", f"r03_bench_unprofiled_{n}.json"))
    d = line(P("profiles", "r03_bench_default_run.json"))
    s3 = L("cfg3_slic")["slic_inclusive"]
    fp = json.load(open(P("profiles", "r03_pmc_hbm_traffic_cfg2.json")))["csrc_sha16"]
    tr = {c: json.load(open(P("profiles", f"r03_pmc_hbm_traffic_{c}.json")))["hbm_bytes_per_step"] / 1e9
          for c in ["cfg1", "cfg2", "cfg3", "cfg4_bf16", "cfg4_fp8", "cfg5"]}
    rows = list(csv.DictReader(open(P("profiles", "r03_bench_kernel_stats_cfg2.csv"))))
    st = int([r for r in rows if "cross_entropy_kernel" in r["Name"]][0]["Calls"])
    cat = {"gemm": 0.0, "ln": 0.0, "attn": 0.0, "other": 0.0}
    for r in rows:
        t, n = float(r["TotalDurationNs"]) / st / 1e6, r["Name"]
        key = ("gemm" if ("gemm" in n or "grouped_reduce" in n) else "ln" if ("ln_fwd" in n or "ln_bwd" in n)
               else "attn" if ("mhla_fwd" in n or "mhla_bwd" in n) else "other")
        cat[key] += t
    tot = sum(cat.values())
    avg = lambda sub: round(float([x for x in rows if sub in x["Name"]][0]["AverageNs"]) / 1e3, 1)
    a = {k: avg(k) for k in ["p4_kernelILb1ELb0EDF16b", "grouped_tn", "grouped_reduce", "mhla_bwd", "mhla_fwd",
                             "ln_bwd_kernelIDF16b", "ln_fwd_half_kernelIDF16b"]}
    names = ["cfg1", "cfg2", "cfg2_drop", "cfg2_fp32", "cfg3", "cfg4_bf16", "cfg4_fp8", "cfg5"]
    ms = {n: L(n)["ms_per_step"] for n in names}
    ips = {n: L(n)["value"] for n in names}
    k = lambda x: f"{x / 1000:.1f}k"

    p = P("profiles", "README.md")
    s = open(p).read()
    i0 = s.index("| `r03_bench_kernel_stats_cfg2.csv` + `r03_bench_under_rocprof_cfg2.json` |")
    i1 = s.index("| `r03_bench_kernel_stats_cfg2_fp32.csv` + json |")
    s = s[:i0] + (
        f"| `r03_bench_kernel_stats_cfg2.csv` + `r03_bench_under_rocprof_cfg2.json` | headline configuration, {st} profiled "
        f"steps (boxes differ by +-1.5 %): {tot:.2f} ms of kernel time per step = GEMMs {cat['gemm']:.2f} (incl. the slab "
        f"reduction), LayerNorm {cat['ln']:.2f}, attention {cat['attn']:.2f}, everything else {cat['other']:.2f}.  "
        f"`gemm_bf16_p4_kernel<true,false,bf16>` {a['p4_kernelILb1ELb0EDF16b']} us x 48, grouped weight gradients "
        f"{a['grouped_tn']} us + {a['grouped_reduce']} us reduction, `mhla_bwd_lse_kernel` {a['mhla_bwd']} us (round 2's "
        f"`mhla_bwd_mfma2_kernel`: 78.9), `mhla_fwd_mfma_kernel` {a['mhla_fwd']} (round 2: 36.3; it now also writes lse), "
        f"`ln_bwd` {a['ln_bwd_kernelIDF16b']}, `ln_fwd_half_kernel` {a['ln_fwd_half_kernelIDF16b']} (25.7) |\n") + s[i1:]
    s = re.sub(r"fingerprint of `csrc/` \(`[0-9a-f]{16}`\)", f"fingerprint of `csrc/` (`{fp}`)", s)
    s = re.sub(r"cfg2 [0-9.]+ GB \([0-9.]+ TB/s over the measured step, [0-9.\-]+ of peak\), cfg4 [0-9.]+ GB \(bf16\) / [0-9.]+ GB "
               r"\(fp8\), cfg3 [0-9.]+ GB, cfg1 [0-9.]+ GB, cfg5 [0-9.]+ GB",
               f"cfg2 {tr['cfg2']:.1f} GB (3.8 TB/s over the measured step, 0.47-0.48 of peak), cfg4 {tr['cfg4_bf16']:.1f} GB "
               f"(bf16) / {tr['cfg4_fp8']:.1f} GB (fp8), cfg3 {tr['cfg3']:.2f} GB, cfg1 {tr['cfg1']:.2f} GB, cfg5 {tr['cfg5']:.2f} GB", s)
    s = re.sub(r"(\| `r03_bench_default_run.json` \| `python bench.py` \(no flags\) un-profiled on a `gpurun` box: )[0-9,]+ img/s, "
               r"[0-9.]+ ms/step", lambda m: f"{m.group(1)}{d['value']:,.0f} img/s, {d['ms_per_step']:.2f} ms/step", s)
    s = re.sub(r"oracle on 16 cores: [0-9.]+ img/s\) \|", f"oracle on 16 cores: {d['cpu_baseline']['value']:.1f} img/s) |", s)
    s = re.sub(r"\| `r03_bench_unprofiled_\*.json` \|[^\n]*\n",
               f"| `r03_bench_unprofiled_*.json` | the un-profiled `bench.py` line of every configuration on one box "
               f"(`--no-cpu-baseline`): cfg1 {ms['cfg1']:.2f} ms, cfg2 {ms['cfg2']:.2f}, cfg2 with dropout 0.1 {ms['cfg2_drop']:.2f}, "
               f"cfg2 fp32 {ms['cfg2_fp32']:.1f}, cfg3 {ms['cfg3']:.2f} ({s3['ms_per_step']:.2f} with the device SLIC inside the step, "
               f"{s3['overlapped']['ms_per_step']:.2f} with it on a side stream), cfg4 {ms['cfg4_bf16']:.2f} (bf16) / "
               f"{ms['cfg4_fp8']:.2f} (fp8), cfg5 {ms['cfg5']:.2f} |\n", s)
    open(p, "w").write(s)

    p = P("DESIGN.md")
    s = open(p).read()
    i0 = s.index("Measured (1 GPU, this round's boxes, `bench.py` un-profiled):")
    i1 = s.index("\n\n", i0)
    rng = re.search(r"the boxes of this round gave ([0-9.]+-[0-9.]+) ms", s[i0:i1])
    rng = rng.group(1) if rng else "14.1-14.5"
    s = s[:i0] + (
        f"Measured (1 GPU, this round's boxes, `bench.py` un-profiled): cfg1 {ms['cfg1']:.2f} ms / {k(ips['cfg1'])} img/s, "
        f"**cfg2 {ms['cfg2']:.2f} ms / {k(ips['cfg2'])} img/s (`profiles/r03_bench_default_run.json`: {d['ms_per_step']:.2f} ms / "
        f"{d['value']:,.0f} img/s; the boxes of this round gave {rng} ms for the same command)**,\ncfg3 {ms['cfg3']:.2f} ms / "
        f"{k(ips['cfg3'])} img/s (with the device SLIC inside the step: {s3['ms_per_step']:.2f} ms / {k(s3['images_per_sec'])} img/s, "
        f"overlapped {s3['overlapped']['ms_per_step']:.2f} / {k(s3['overlapped']['images_per_sec'])}), cfg4 {ms['cfg4_bf16']:.2f} ms / "
        f"{ips['cfg4_bf16']:,.0f} img/s in bf16 and {ms['cfg4_fp8']:.2f} ms / {ips['cfg4_fp8']:,.0f} img/s with\nfp8 GEMMs, cfg5 "
        f"{ms['cfg5']:.2f} ms / {k(ips['cfg5'])} img/s; cfg2 in fp32 (exact-fp32 MFMA, the mode in which the logits are within 1e-3 "
        f"of the\nreference): {ms['cfg2_fp32']:.1f} ms / {ips['cfg2_fp32']:,.0f} img/s.") + s[i1:]
    s = re.sub(r"[0-9.]+ ms/step beside [0-9.]+ ms without dropout \(same box\)",
               f"{ms['cfg2_drop']:.2f} ms/step beside {ms['cfg2']:.2f} ms without dropout (same box)", s)
    s = re.sub(r"the kernels of one step move [0-9.]+ GB through HBM", f"the kernels of one step move {tr['cfg2']:.1f} GB through HBM", s)
    open(p, "w").write(s)
    print("synced: fingerprint", fp, "cfg2", ms["cfg2"], "ms; default run", d["ms_per_step"], "ms")


if __name__ == "__main__":
    main()
