#!/usr/bin/env python3
"""Informational: the vendor library (torch.mm -> hipBLASLt / rocBLAS) on the step's GEMM shapes, beside
favit_gemm.  Not used by the package."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("focused-attention-vit_amd")
K = pkg.kernels
T, D = 256 * 197, 384
bf = torch.bfloat16
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def rnd(*s): return torch.randn(*s, device="cuda").to(bf)
for name, M, N, Kd, nt in (("qkv NT", T, 3 * D, D, True), ("fc1 NT", T, 4 * D, D, True), ("fc2 NT", T, D, 4 * D, True),
                           ("dXn2 NN", T, D, 4 * D, False), ("dX NN", T, D, 3 * D, False), ("dH NN", T, 4 * D, D, False)):
    a = rnd(M, Kd)
    b = rnd(N, Kd) if nt else rnd(Kd, N)
    c = torch.empty(M, N, device="cuda", dtype=bf)
    lib = t(lambda: torch.mm(a, b.t() if nt else b, out=c))
    mine = t(lambda: K.gemm(a, b, c, M, N, Kd, Kd, Kd if nt else N, N, b_kmajor=nt))
    fl = 2.0 * M * N * Kd
    print(f"{name:8s} M={M} N={N:5d} K={Kd:5d}: vendor {lib:7.1f} us ({fl / lib / 1e6:6.0f} TF)   favit_gemm {mine:7.1f} us ({fl / mine / 1e6:6.0f} TF)", flush=True)
