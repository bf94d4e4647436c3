#!/usr/bin/env python3
"""Micro-benchmark of the MHLA attention kernels at the bench shape (B=256, L=197, H=6, hd=64, W=7)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("focused-attention-vit_amd")
if os.environ.get("FAVIT_MHLA_DBG"):      # probe build: work-skipping switches (wrong results, timing only)
    pkg._abi.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe_build", "libfavit_probe.so")
K = pkg.kernels
B, L, H, hd, W = (int(v) for v in os.environ.get("ATTN_SHAPE", "256,197,6,64,7").split(","))
D = H * hd
qkv = torch.randn(B * L, 3 * D, device="cuda").to(torch.bfloat16)
do = torch.randn(B * L, D, device="cuda").to(torch.bfloat16)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print(f"fwd {t(lambda: K.mhla_attn_fwd(qkv, B, L, H, hd, W)):8.1f} us   bwd {t(lambda: K.mhla_attn_bwd(qkv, do, B, L, H, hd, W)):8.1f} us")
o, lse = K.mhla_attn_fwd(qkv, B, L, H, hd, W, want_lse=True)
if lse is not None:      # the training path's pair: forward leaves lse, backward takes it and o (FAVIT_MHLA_LSE_WAVES=1..4)
    print(f"fwd+lse {t(lambda: K.mhla_attn_fwd(qkv, B, L, H, hd, W, want_lse=True)):8.1f} us   "
          f"bwd(lse) {t(lambda: K.mhla_attn_bwd(qkv, do, B, L, H, hd, W, o=o, lse=lse)):8.1f} us")
