#!/usr/bin/env python3
"""Uninitialised-read hunt: torch.empty / empty_like / new_empty are replaced by versions that fill the new tensor with NaN
(floating types) or 0xFF bytes (integer types: as fp32 that is a NaN too), then a configuration's training step runs
eagerly and from HIP graphs.  Any kernel that reads memory nobody wrote turns the loss or a parameter into NaN
deterministically instead of once in a hundred fresh processes.   usage: poison_probe.py [cfg3] [steps]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch

_empty, _empty_like = torch.empty, torch.empty_like


def _poison(t):
    if t.is_cuda and t.numel():
        if t.dtype.is_floating_point:
            t.fill_(float("nan"))
        elif t.dtype in (torch.uint8, torch.int8, torch.int16, torch.int32, torch.int64):
            t.view(torch.uint8).fill_(0xFF) if t.is_contiguous() else None
    return t


torch.empty = lambda *a, **k: _poison(_empty(*a, **k))
torch.empty_like = lambda *a, **k: _poison(_empty_like(*a, **k))

import bench
pkg = importlib.import_module("focused-attention-vit_amd")
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
c = bench.CONFIGS[cfg]
dev = torch.device("cuda", 0)
pkg.set_compute_dtype(os.environ.get("DTYPE", "bf16"))
B = int(os.environ.get("BATCH", c["batch"]))
g = torch.Generator(device=dev).manual_seed(1234)
images = torch.randn(B, 3, c["img"], c["img"], device=dev, generator=g)
labels = torch.randint(0, c["classes"], (B,), device=dev, generator=g)


def fresh():
    torch.manual_seed(1234)
    m = bench.build_model(pkg, cfg, dev, float(os.environ.get("DROPOUT", "0"))).train()
    if cfg in ("cfg3", "cfg5"):
        segs_np = bench.synthetic_label_maps(8, 224, 16, seed=100)
        m.segmentation.set_label_maps(torch.from_numpy(np.stack([segs_np[i % 8] for i in range(B)])).to(dev))
        m.assume_num_tokens = 16
    o = pkg.train.FusedAdamW(pkg.train.param_groups(m, lr=1e-4), lr=1e-4, weight_decay=0.05)
    return m, o


def finite(m):
    return [n for n, p in m.named_parameters() if not bool(torch.isfinite(p).all())]


m, o = fresh()
for s in range(steps):
    loss = pkg.train.train_step(m, images, labels, o)
    print(f"eager step {s}: loss {loss.item():.5f}  non-finite parameters: {finite(m)[:4]}", flush=True)
pkg.functional.clear_lp_mirrors()
m, o = fresh()
gs = pkg.train.GraphedStep(m, o, images, labels)
for s in range(steps):
    loss = gs(images, labels)
    print(f"graph step {s}: loss {loss.item():.5f}  non-finite parameters: {finite(m)[:4]}", flush=True)
