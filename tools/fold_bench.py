#!/usr/bin/env python3
"""The latent_proj fold kernels in isolation: forward fold of 12 layers (one launch), backward fold of 12 layers with
and without its qkv half (frozen qkv projection), at D = 384 / H = 6 and D = 768 / H = 12."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("focused-attention-vit_amd"); K = pkg.kernels
dev = "cuda"
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for D, H in ((384, 6), (768, 12)):
    hd = D // H
    g = torch.Generator(device=dev).manual_seed(D)
    full, lat = [], []
    for i in range(12):
        dweff = torch.randn(3 * D, D, device=dev, generator=g); dbeff = torch.randn(3 * D, device=dev, generator=g)
        wqkv = torch.randn(3 * D, D, device=dev, generator=g); bqkv = torch.randn(3 * D, device=dev, generator=g)
        wl = torch.randn(hd, hd, device=dev, generator=g)
        outs = [torch.zeros(3 * D, D, device=dev), torch.zeros(3 * D, device=dev), torch.zeros(hd, hd, device=dev), torch.zeros(hd, device=dev)]
        full.append((dweff, dbeff, wqkv, bqkv, wl, outs))
        lat.append((dweff, dbeff, wqkv, bqkv, wl, [None, None, outs[2], outs[3]]))
    print(f"D={D} H={H}: fold_bwd_multi full {t(lambda: K.mhla_fold_bwd_multi(full, H)):7.1f} us   latent only {t(lambda: K.mhla_fold_bwd_multi(lat, H)):7.1f} us")
