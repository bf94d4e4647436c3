#!/usr/bin/env python3
"""HBM streaming rates on this box (pure write / copy / read), to price memory-bound kernels."""
import torch
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mb in (155, 620, 2048):
    n = mb * (1 << 20) // 4
    x = torch.empty(n, device="cuda"); y = torch.empty(n, device="cuda")
    s = t(lambda: x.zero_());  print(f"{mb:5d} MiB fill : {mb / 1024 / s / 1e0:7.2f} GiB/ms -> {mb * 1.048576e6 / s / 1e12:5.2f} TB/s write")
    s = t(lambda: y.copy_(x)); print(f"{mb:5d} MiB copy : {2 * mb * 1.048576e6 / s / 1e12:5.2f} TB/s read+write")
    s = t(lambda: x.sum());    print(f"{mb:5d} MiB sum  : {mb * 1.048576e6 / s / 1e12:5.2f} TB/s read")
