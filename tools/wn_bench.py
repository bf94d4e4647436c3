#!/usr/bin/env python3
"""The whole-row GEMM with the next LayerNorm in its epilogue (favit_gemm_residual_ln) against what it replaces
(favit_gemm with bias + residual, then favit_layernorm_fwd) at the two N = 384 shapes of the cfg2 block."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("focused-attention-vit_amd")
if os.environ.get("FAVIT_WN_DBG") or os.environ.get("FAVIT_GEMM_DBG"):
    pkg._abi.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe_build", "libfavit_probe.so")
K = pkg.kernels
dev, bf = "cuda", torch.bfloat16
T, D = 256 * 197, 384
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, Kd in (("proj K=384", D), ("fc2  K=1536", 4 * D)):
    a = torch.randn(T, Kd, device=dev).to(bf); w = (torch.randn(D, Kd, device=dev) * Kd ** -0.5).to(bf)
    bias = torch.randn(D, device=dev); res = torch.randn(T, D, device=dev)
    gamma = torch.rand(D, device=dev) + 0.5; beta = torch.randn(D, device=dev)
    out = torch.empty(T, D, device=dev)
    def sep():
        K.gemm(a, w, out, T, D, Kd, Kd, Kd, D, bias=bias, residual=res, ld_res=D)
        return K.layernorm_fwd(out, D, gamma, beta, T, D, bf)
    def gemm_only():
        K.gemm(a, w, out, T, D, Kd, Kd, Kd, D, bias=bias, residual=res, ld_res=D)
    fused = lambda: K.gemm_residual_ln(a, w, bias, res, gamma, beta)
    x2, xn2, mu2, rs2 = fused()
    xn1, mu1, rs1 = sep()
    torch.cuda.synchronize()
    rel = lambda p, q: float((p.float() - q.float()).norm() / q.float().norm())
    print(f"{name}: x {rel(x2, out):.2e}  xn {rel(xn2, xn1):.2e}  mean {rel(mu2, mu1):.2e}  rstd {rel(rs2, rs1):.2e}")
    print(f"   GEMM {t(gemm_only):7.1f} us   GEMM + LayerNorm {t(sep):7.1f} us   fused {t(fused):7.1f} us")
