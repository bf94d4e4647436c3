#!/usr/bin/env python3
"""L2->LDS DMA rate of the 256x128 GEMM tile traffic vs fetch granularity (64 B vs 128 B per row
per step), stage count and workgroups per CU."""
import ctypes, os, sys
import torch
pl = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libprobe.so"))
pl.probe_dma.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_long, ctypes.c_long, ctypes.c_int,
                         ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = "cuda"
T = 256 * 197
def run(K, N, chunk, nstage, lds, label):
    A = torch.randn(T, K, device=dev).to(torch.bfloat16)
    W = torch.randn(N, K, device=dev).to(torch.bfloat16)
    tiles_n = N // 128
    ntiles = (T // 256) * tiles_n
    ksteps = K * 2 // chunk
    st = torch.cuda.current_stream().cuda_stream
    def f():
        rc = pl.probe_dma(A.data_ptr(), W.data_ptr(), K * 2, K * 2, T, tiles_n, ntiles, ksteps, chunk, nstage, lds, st)
        assert rc == 0, rc
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    byts = ntiles * ksteps * 384 * chunk
    print(f"K={K:5d} N={N:5d} {label:34s} {us:8.1f} us  {byts / us / 1e6:6.2f} TB/s L2->LDS  (= {2 * T * N * K / us / 1e6:7.1f} TF-equivalent)", flush=True)

for K, N in ((384, 1152), (384, 1536), (1536, 384)):
    run(K, N, 64, 3, 3 * 24576, "64B x3 stages, 2 WG/CU")
    run(K, N, 64, 3, 100 * 1024, "64B x3 stages, 1 WG/CU")
    run(K, N, 64, 4, 4 * 24576, "64B x4 stages, 1 WG/CU")
    run(K, N, 128, 2, 2 * 49152, "128B x2 stages, 1 WG/CU")
    run(K, N, 128, 3, 3 * 49152, "128B x3 stages, 1 WG/CU")
