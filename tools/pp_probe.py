#!/usr/bin/env python3
"""In-kernel cycle stamps of the ping-pong GEMM (probe build: make -C focused-attention-vit_amd/csrc probe).
Prints, per ping-pong half, the mean cycles a wave spends per 32-deep k-step in the read section + X wait,
in the fence + MFMA issue, and in the Y wait; ideal = 256 + 256 (own cluster + partner's)."""
import ctypes, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("focused-attention-vit_amd")
pkg._abi.LIB_PATH = os.path.join(os.path.dirname(pkg._abi.LIB_PATH), "libfavit_probe.so")
K = pkg.kernels
lib = pkg._abi.lib()
lib.favit_probe_buffer.argtypes = [ctypes.c_void_p]
lib.favit_probe_buffer.restype = None
dev = "cuda"
shapes = [(8192, 8192 + 128, 8192, True), (50432, 384, 1536, True), (50432, 1152, 384, True)]   # N % 256 != 0: not the 256x256 kernel
for M, N, Kd, bk in shapes:
    a = torch.randn(M, Kd, device=dev).bfloat16()
    b = (torch.randn(N, Kd, device=dev) if bk else torch.randn(Kd, N, device=dev)).bfloat16()
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    tiles = ((M + 255) // 256) * ((N + 127) // 128)
    buf = torch.zeros(tiles * 8 * 5, dtype=torch.int64, device=dev)
    lib.favit_probe_buffer(ctypes.c_void_p(buf.data_ptr()))
    for _ in range(3):
        K.gemm(a, b, c, M, N, Kd, Kd, Kd if bk else N, N, b_kmajor=bk)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    K.gemm(a, b, c, M, N, Kd, Kd, Kd if bk else N, N, b_kmajor=bk)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    t = buf.view(tiles, 8, 5).double()
    steps = 2 * (Kd // 64)
    print(f"M={M} N={N} K={Kd} {'NT' if bk else 'NN'}: {us:.1f} us, {2.0 * M * N * Kd / us / 1e6:.0f} TF, {tiles} tiles")
    for h in (0, 1):
        w = t[:, 4 * h:4 * h + 4].reshape(-1, 5).mean(0)
        print(f"  half {h}: per k-step read+X {w[0] / steps:7.1f}  fence+mfma {w[1] / steps:7.1f}  Y wait {w[2] / steps:7.1f}"
              f"  | loop {w[3]:9.0f} cyc ({w[3] / steps:6.1f}/step), whole kernel {w[4]:9.0f} cyc")
    lib.favit_probe_buffer(None)
    # the same launch without stamps (each stamp is an SMEM round trip): event time only
    for _ in range(2):
        K.gemm(a, b, c, M, N, Kd, Kd, Kd if bk else N, N, b_kmajor=bk)
    e0.record()
    for _ in range(5):
        K.gemm(a, b, c, M, N, Kd, Kd, Kd if bk else N, N, b_kmajor=bk)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 5
    print(f"  unstamped: {us:.1f} us, {2.0 * M * N * Kd / us / 1e6:.0f} TF")
