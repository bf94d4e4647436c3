#!/usr/bin/env python3
"""What does the GELU / dGELU arithmetic in the GEMM epilogue cost?  fc1 forward (two bf16 outputs) and the fc2
input-gradient GEMM (dGELU on a saved pre-activation) with and without the activation, same stores and loads."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("focused-attention-vit_amd")
K, A = pkg.kernels, pkg._abi
dev, bf = "cuda", torch.bfloat16
T, D = 256 * 197, 384
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
x = torch.randn(T, D, device=dev).to(bf); w1 = torch.randn(4 * D, D, device=dev).to(bf); b1 = torch.randn(4 * D, device=dev)
h = torch.empty(T, 4 * D, device=dev, dtype=bf); pre = torch.empty_like(h)
print(f"fc1 fwd  bias+GELU+aux_out {t(lambda: K.gemm(x, w1, h, T, 4 * D, D, D, D, 4 * D, bias=b1, act=A.ACT_GELU, aux_out=pre, ld_aux_out=4 * D)):7.1f} us"
      f"   bias+aux_out only {t(lambda: K.gemm(x, w1, h, T, 4 * D, D, D, D, 4 * D, bias=b1, aux_out=pre, ld_aux_out=4 * D)):7.1f} us"
      f"   single output {t(lambda: K.gemm(x, w1, h, T, 4 * D, D, D, D, 4 * D, bias=b1)):7.1f} us")
w2 = torch.randn(D, 4 * D, device=dev).to(bf)
print(f"dH  bwd  dGELU(aux_in)     {t(lambda: K.gemm(x, w2, h, T, 4 * D, D, D, 4 * D, 4 * D, b_kmajor=False, act=A.ACT_DGELU, aux_in=pre, ld_aux_in=4 * D)):7.1f} us"
      f"   no activation     {t(lambda: K.gemm(x, w2, h, T, 4 * D, D, D, 4 * D, 4 * D, b_kmajor=False)):7.1f} us")
