#!/usr/bin/env python3
"""Does the p4 main loop run faster when an operand is fetched in whole 128-byte lines?  The same GEMM shape with the A
operand k-major ([M][K]: 64-byte row segments per BK = 32 stage, half lines) and mn-major ([K][M]: 256-byte rows, whole
lines), B either way; FAVIT_GEMM_DBG=1 (probe build) times the main loops without their epilogues.
DESIGN.md section 4, "the CU's memory pipe"."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("focused-attention-vit_amd")
if os.environ.get("FAVIT_GEMM_DBG"):
    pkg._abi.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe_build", "libfavit_probe.so")
K = pkg.kernels
dev = "cuda"
def t(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for (M, N, Kd) in ((50432, 1536, 384), (50432, 1152, 384), (50432, 384, 1536), (50432, 384, 384)):
    g = torch.Generator(device=dev).manual_seed(1)
    Ak = torch.randn(M, Kd, device=dev, generator=g).bfloat16()
    Am = Ak.t().contiguous()                       # [K][M]
    Bk = (torch.randn(N, Kd, device=dev, generator=g) * 0.05).bfloat16()
    Bm = Bk.t().contiguous()                       # [K][N]
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    res = []
    for ak, bk in ((True, True), (False, True), (True, False), (False, False)):
        A, lda = (Ak, Kd) if ak else (Am, M)
        B, ldb = (Bk, Kd) if bk else (Bm, N)
        us = t(lambda: K.gemm(A, B, C, M, N, Kd, lda, ldb, N, a_kmajor=ak, b_kmajor=bk))
        res.append(f"A {'k' if ak else 'mn'}-major B {'k' if bk else 'mn'}-major {us:7.1f} us {2.0 * M * N * Kd / us / 1e6:6.0f} TF")
    print(f"M={M} N={N} K={Kd}: " + " | ".join(res))
