#!/usr/bin/env python3
"""Does splitting the N = 384 GEMMs of the step along M -- 510 tiles (one full round of 512 workgroup slots) on the
256x128 kernel + the last 6,912 rows on the 64x128 kernel -- beat one launch of 591 tiles (1.15 rounds)?"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("focused-attention-vit_amd")
K = pkg.kernels
A = pkg._abi
dev = "cuda"
T, D = 256 * 197, 384
bf = torch.bfloat16
def rnd(*s, dt=bf): return torch.randn(*s, device=dev).to(dt)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
cases = [("fc2 NT K=1536 res f32", 4 * D, True, torch.float32, True), ("dXn2 NN K=1536", 4 * D, False, bf, False),
         ("dXqkv NN K=1152", 3 * D, False, bf, False), ("proj NT K=384 res f32", D, True, torch.float32, True),
         ("dXproj NN K=384", D, False, bf, False)]
for name, Kd, bk, odt, res in cases:
    a = rnd(T, Kd)
    w = rnd(D, Kd) if bk else rnd(Kd, D)
    out = torch.empty(T, D, device=dev, dtype=odt)
    r = torch.randn(T, D, device=dev) if res else None
    bias = torch.randn(D, device=dev) if res else None
    ldb = Kd if bk else D
    def one():
        K.gemm(a, w, out, T, D, Kd, Kd, ldb, D, b_kmajor=bk, bias=bias, residual=r, ld_res=D)
    for split_rows in (170 * 256, 168 * 256, 160 * 256):
        M0 = split_rows
        def two():
            K.gemm(a, w, out, M0, D, Kd, Kd, ldb, D, b_kmajor=bk, bias=bias, residual=r, ld_res=D)
            K.gemm(a[M0:], w, out[M0:], T - M0, D, Kd, Kd, ldb, D, b_kmajor=bk, bias=bias,
                   residual=None if r is None else r[M0:], ld_res=D)
        print(f"{name:24s} one launch {t(one):7.1f} us   split at {M0 // 256:3d} panels {t(two):7.1f} us", flush=True)
