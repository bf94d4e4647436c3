#!/usr/bin/env python3
"""Two forward + backward passes of a bench configuration on the same inputs and weights: which parameter gradients are
bitwise equal, and how far the others differ (fp32 atomics: latent_proj fold, cls / pos gradients) -- a race would
show as a large difference in a gradient that has no atomic on its path.   usage: determinism_probe.py [cfg3]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
pkg = importlib.import_module("focused-attention-vit_amd")
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
c = bench.CONFIGS[cfg]
dev = torch.device("cuda", 0)
pkg.set_compute_dtype("bf16")
torch.manual_seed(1234)
model = bench.build_model(pkg, cfg, dev).train()
B = int(os.environ.get("BATCH", c["batch"]))
g = torch.Generator(device=dev).manual_seed(1234)
images = torch.randn(B, 3, c["img"], c["img"], device=dev, generator=g)
labels = torch.randint(0, c["classes"], (B,), device=dev, generator=g)
if cfg in ("cfg3", "cfg5"):
    segs_np = bench.synthetic_label_maps(8, 224, 16, seed=100)
    model.segmentation.set_label_maps(torch.from_numpy(np.stack([segs_np[i % 8] for i in range(B)])).to(dev))
    model.assume_num_tokens = 16
pkg.set_direct_grads(False)
runs = []
for r in range(3):
    model.zero_grad(set_to_none=True)
    loss = pkg.train.cross_entropy(model(images), labels)
    loss.backward()
    torch.cuda.synchronize()
    runs.append((loss.item(), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}))
print("losses", [r[0] for r in runs])
worst = []
for n in runs[0][1]:
    a, b, c2 = (runs[i][1][n].float() for i in range(3))
    d = max((a - b).abs().max().item(), (a - c2).abs().max().item())
    if d > 0:
        worst.append((d / (a.abs().max().item() + 1e-30), n))
worst.sort(reverse=True)
print(f"{len(runs[0][1]) - len(worst)} of {len(runs[0][1])} gradients bitwise equal over 3 passes; the others (max |diff| / max |g|):")
for d, n in worst[:25]:
    print(f"  {d:10.3e}  {n}")
