#!/usr/bin/env python3
"""A bench configuration's replayed training step, many times from fresh models: is the loss finite after every run?
usage: stress_graph.py [cfg3] [runs] [steps]   (each run: new model + optimizer + GraphedStep, `steps` replays)"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
pkg = importlib.import_module("focused-attention-vit_amd")
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 20
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 25
c = bench.CONFIGS[cfg]
dev = torch.device("cuda", 0)
pkg.set_compute_dtype("bf16")
B = c["batch"]
g = torch.Generator(device=dev).manual_seed(1234)
images = torch.randn(B, 3, c["img"], c["img"], device=dev, generator=g)
labels = torch.randint(0, c["classes"], (B,), device=dev, generator=g)
bad = 0
for r in range(runs):
    torch.manual_seed(1234)
    model = bench.build_model(pkg, cfg, dev).train()
    if cfg in ("cfg3", "cfg5"):
        segs_np = bench.synthetic_label_maps(8, 224, 16, seed=100)
        model.segmentation.set_label_maps(torch.from_numpy(np.stack([segs_np[i % 8] for i in range(B)])).to(dev))
        model.assume_num_tokens = 16
    opt = pkg.train.FusedAdamW(pkg.train.param_groups(model, lr=1e-4), lr=1e-4, weight_decay=0.05)
    gs = pkg.train.GraphedStep(model, opt, images, labels)
    losses = []
    for s in range(steps):
        losses.append(gs(images, labels))
    torch.cuda.synchronize()
    vals = [float(l) for l in torch.stack([l.detach().float().reshape(()) for l in losses]).tolist()] if False else [float(losses[-1])]
    pfin = all(bool(torch.isfinite(p).all()) for p in model.parameters())
    ok = np.isfinite(vals[-1]) and pfin
    if not ok:
        bad += 1
    print(f"run {r}: last loss {vals[-1]:.5f} params finite {pfin}", flush=True)
    pkg.functional.clear_lp_mirrors()
    del gs, opt, model
print("done:", runs, "runs,", bad, "bad")
