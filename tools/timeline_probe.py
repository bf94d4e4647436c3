#!/usr/bin/env python3
"""Per-workgroup timeline of the p4 GEMM (probe build, FAVIT_GEMM_DBG=128): start / main loop done / end of every
workgroup in 10-ns ticks plus its CU (HW_REG_HW_ID, XCC_ID).  Reports, per launch: phase durations, and for every CU
how much of a workgroup's main loop ran while the CU's other workgroup was in ITS main loop (lockstep) vs in its
epilogue (the overlap a two-workgroup-per-CU design hopes for)."""
import ctypes, importlib, os, sys
os.environ["FAVIT_GEMM_DBG"] = str(128 | int(os.environ.get("EXTRA_DBG", "0")))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
pkg = importlib.import_module("focused-attention-vit_amd")
pkg._abi.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe_build", "libfavit_probe.so")
K, A = pkg.kernels, pkg._abi
lib = A.lib()
lib.favit_probe_buffer.argtypes = [ctypes.c_void_p]
lib.favit_probe_buffer.restype = None
dev, bf = "cuda", torch.bfloat16
T, D = 256 * 197, 384
x = torch.randn(T, D, device=dev).to(bf); h = torch.randn(T, 4 * D, device=dev).to(bf)
w1 = torch.randn(4 * D, D, device=dev).to(bf); w2 = torch.randn(D, 4 * D, device=dev).to(bf); wq = torch.randn(3 * D, D, device=dev).to(bf)
b1 = torch.randn(4 * D, device=dev); b2 = torch.randn(D, device=dev); bq = torch.randn(3 * D, device=dev)
res = torch.randn(T, D, device=dev)
o_h = torch.empty(T, 4 * D, device=dev, dtype=bf); o_pre = torch.empty_like(o_h)
o_q = torch.empty(T, 3 * D, device=dev, dtype=bf); o_d = torch.empty(T, D, device=dev)
cases = [
    ("fc1 fwd N=1536 K=384 gelu", lambda: K.gemm(x, w1, o_h, T, 4 * D, D, D, D, 4 * D, bias=b1, act=A.ACT_GELU, aux_out=o_pre, ld_aux_out=4 * D), 197 * 12),
    ("qkv fwd N=1152 K=384", lambda: K.gemm(x, wq, o_q, T, 3 * D, D, D, D, 3 * D, bias=bq), 197 * 9),
    ("fc2 fwd N=384 K=1536 res", lambda: K.gemm(h, w2, o_d, T, D, 4 * D, 4 * D, 4 * D, D, bias=b2, residual=res, ld_res=D), 197 * 3),
]
for name, fn, tiles in cases:
    buf = torch.zeros(tiles * 4, dtype=torch.int64, device=dev)
    # sustained conditions: the stamped launch follows 40 back-to-back launches without a sync (the chip lowers its
    # clock under MFMA load; a launch after an idle gap runs ~25 % faster and is not representative)
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    for _ in range(10): fn()
    e0.record()
    for _ in range(30): fn()
    e1.record()
    lib.favit_probe_buffer(ctypes.c_void_p(buf.data_ptr()))
    fn()
    e2.record()
    torch.cuda.synchronize()
    lib.favit_probe_buffer(None)
    print(f"[{name}] back-to-back average {e0.elapsed_time(e1) / 30 * 1e3:.1f} us, the stamped launch {e1.elapsed_time(e2) * 1e3:.1f} us (events)")
    t = buf.view(tiles, 4).cpu().numpy()
    t0, t1, t2, hw = t[:, 0], t[:, 1], t[:, 2], t[:, 3]
    base = t0.min()
    t0, t1, t2 = (t0 - base) / 100.0, (t1 - base) / 100.0, (t2 - base) / 100.0           # us
    cu = ((hw >> 32) & 0xF) * 256 + ((hw >> 8) & 0xFF)
    print(f"{name}: {tiles} workgroups on {len(np.unique(cu))} CUs, launch span {t2.max():.1f} us; "
          f"main loop {np.mean(t1 - t0):.2f} us (p10 {np.percentile(t1 - t0, 10):.2f}, p90 {np.percentile(t1 - t0, 90):.2f}), "
          f"epilogue {np.mean(t2 - t1):.2f} us (p10 {np.percentile(t2 - t1, 10):.2f}, p90 {np.percentile(t2 - t1, 90):.2f})")
    # overlap accounting per CU
    both_main = main_vs_epi = main_alone = 0.0
    for c in np.unique(cu):
        idx = np.nonzero(cu == c)[0]
        for i in idx:
            a0, a1 = t0[i], t1[i]
            om = oe = 0.0
            for j in idx:
                if j == i: continue
                om += max(0.0, min(a1, t1[j]) - max(a0, t0[j]))
                oe += max(0.0, min(a1, t2[j]) - max(a0, t1[j]))
            both_main += om; main_vs_epi += oe; main_alone += max(0.0, (a1 - a0) - om - oe)
    tot = both_main + main_vs_epi + main_alone
    print(f"   of all main-loop time: {100 * both_main / tot:.0f} % beside the neighbour's main loop, "
          f"{100 * main_vs_epi / tot:.0f} % beside its epilogue, {100 * main_alone / tot:.0f} % alone on the CU")
    # start-time histogram of the first 512 workgroups, and phase alignment chip-wide
    edges = np.arange(0, t2.max() + 2, 2.0)
    in_main = [(np.sum((t0 <= e) & (t1 > e))) for e in edges]
    in_epi = [(np.sum((t1 <= e) & (t2 > e))) for e in edges]
    print("   t(us):   " + " ".join(f"{int(e):4d}" for e in edges[::2]))
    print("   in main: " + " ".join(f"{v:4d}" for v in in_main[::2]))
    print("   in epi:  " + " ".join(f"{v:4d}" for v in in_epi[::2]))
