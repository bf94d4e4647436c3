#!/usr/bin/env python3
"""Training-step time of SPPP+MHLA Small (BASELINE.json configs[2] per-GPU shape: 128 images, 224/p16,
16 superpixels -> 17 tokens).  Label maps are synthetic Voronoi maps made on the host outside the
timed region (SLIC itself is out of scope)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
pkg = importlib.import_module("focused-attention-vit_amd")
from oracle.favit_oracle import voronoi_labels          # input generator only

B = int(os.environ.get("B", "128"))
mode = os.environ.get("MODE", "bf16")
pkg.set_compute_dtype(mode)
torch.manual_seed(1234)
m = pkg.models.sppp_mhla.SPPPViTMHLA(img_size=224, patch_size=16, num_classes=1000, embed_dim=384, depth=12, num_heads=6,
                                     num_superpixels=16, pooling_type="mean", window_size=7, use_mhla=True).cuda().train()
maps, seed = [], 100
mapper = pkg.models.sppp.PatchToSuperpixelMapper(16)
while len(maps) < 8:
    sm = voronoi_labels(224, 16, seed=seed)
    seed += 1
    if len(mapper.map_patches(torch.from_numpy(sm).cuda(), 224)) == 16:
        maps.append(sm)
segs = torch.from_numpy(np.stack([maps[i % 8] for i in range(B)])).cuda()
m.segmentation.set_label_maps(segs)
x = torch.randn(B, 3, 224, 224, device="cuda")
y = torch.randint(0, 1000, (B,), device="cuda")
opt = pkg.train.FusedAdamW(pkg.train.param_groups(m, lr=1e-4), lr=1e-4, weight_decay=0.05)
graph = os.environ.get("GRAPH", "0") == "1"
if graph:
    m.assume_num_tokens = 16              # skips the per-forward host check of the token count
    step = pkg.train.GraphedStep(m, opt, x, y)
else:
    step = lambda a, b: pkg.train.train_step(m, a, b, opt)
for _ in range(3):
    step(x, y)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    loss = step(x, y)
t_cpu = time.perf_counter() - t0
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"SPPP+MHLA Small B={B} {mode} graph={int(graph)}: {1e3 * dt / n:.2f} ms/step  ({B * n / dt:.0f} img/s), host enqueue {1e3 * t_cpu / n:.2f} ms/step, loss {loss.item():.3f}")
