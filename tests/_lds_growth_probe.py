"""Run in a FRESH process by tests/test_gpu_cfg4_shapes.py::test_dynamic_lds_limit_grows_with_later_larger_requests:
the same kernel instantiation is first launched with a small dynamic-LDS size, then with a larger one
(hipFuncAttributeMaxDynamicSharedMemorySize is per (kernel, device) and must be raised, csrc/common.h)."""
import importlib
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("focused-attention-vit_amd")
K, V = pkg.kernels, pkg.functional._View
DEV = "cuda"


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm()).item()


def sdpa(dtype, hd, L=24, B=2):
    g = torch.Generator(device=DEV).manual_seed(hd)
    q, k, v = ((torch.randn(B * L, hd, device=DEV, generator=g) * 0.5).to(dtype) for _ in range(3))
    o = torch.empty_like(q)
    vw = lambda t: V(t, 0, hd, L * hd, hd)
    K.sdpa_fwd(vw(q), vw(k), vw(v), vw(o), B, 1, L, L, hd, 1.0 / math.sqrt(hd))
    qf, kf, vf = (t.float().reshape(B, L, hd) for t in (q, k, v))
    ref = torch.softmax(qf @ kf.transpose(1, 2) / math.sqrt(hd), -1) @ vf
    e = rel(o.float().reshape(B, L, hd), ref)
    assert e < (2e-5 if dtype == torch.float32 else 1e-2), (dtype, hd, e)


def mhla_bwd(L, hd=128, B=2, H=1, W=7):
    g = torch.Generator(device=DEV).manual_seed(L)
    D = H * hd
    qkv = (torch.randn(B * L, 3 * D, device=DEV, generator=g) * 0.5).to(torch.bfloat16)
    dout = (torch.randn(B * L, D, device=DEV, generator=g) * 0.5).to(torch.bfloat16)
    out = K.mhla_attn_fwd(qkv, B, L, H, hd, W)
    dqkv = K.mhla_attn_bwd(qkv, dout, B, L, H, hd, W)
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all() and torch.isfinite(dqkv.float()).all()
    # out is linear in V: <dout, out> == <dV, V>
    lhs = (dout.float() * out.float()).sum().item()
    rhs = (dqkv[:, 2 * D:].float() * qkv[:, 2 * D:].float()).sum().item()
    scale = (dout.float() * out.float()).pow(2).sum().sqrt().item()
    assert abs(lhs - rhs) < 3e-2 * scale, (L, lhs, rhs)


# small head dim first, then larger ones on the SAME instantiation (fp32: single-head CrossAttention D = 256 -> 384)
sdpa(torch.float32, 256)
sdpa(torch.float32, 384)
sdpa(torch.bfloat16, 512)
sdpa(torch.bfloat16, 768)
mhla_bwd(197)
mhla_bwd(64)
torch.cuda.synchronize()
print("LDS_GROWTH_OK")
