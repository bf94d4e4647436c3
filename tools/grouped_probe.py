#!/usr/bin/env python3
"""Grouped weight-gradient launch (the four dW = dY^T X of one block) at the cfg2 / cfg4 shapes: time with and
without the fp32-atomic epilogue (probe build, FAVIT_GEMM_DBG=1 skips the epilogue -> wrong results, timing only)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("focused-attention-vit_amd")
if os.environ.get("FAVIT_GEMM_DBG"):
    pkg._abi.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe_build", "libfavit_probe.so")
K = pkg.kernels
dev = "cuda"
for name, T, D in (("cfg2 Small", 256 * 197, 384), ("cfg4 Base", 64 * 577, 768)):
    shapes = [(D, 4 * D), (4 * D, D), (D, D), (3 * D, D)]
    probs = []
    for N, Kd in shapes:
        probs.append((torch.randn(T, N, device=dev).bfloat16(), torch.randn(T, Kd, device=dev).bfloat16(),
                      torch.zeros(N, Kd, device=dev), torch.zeros(N, device=dev), True))
    for _ in range(3):
        assert K.gemm_grouped_tn(probs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        K.gemm_grouped_tn(probs)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    fl = sum(2.0 * T * n * k for n, k in shapes)
    print(f"{name}: T={T} D={D}: {us:.1f} us  {fl / us / 1e6:.0f} TF   dW bytes x splits ~ {sum(n * k for n, k in shapes) * 4 * 8 / 1e6:.0f} MB of atomics")
