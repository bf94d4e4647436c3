#!/usr/bin/env python3
"""N forward + backward passes of a bench configuration on the same inputs: every pass must reproduce pass 0 bit for bit
in the gradients that have no fp32 atomic on their path, and stay finite everywhere.  Prints the first tensors that
deviate (a rare race or an uninitialised read shows up here long before it shows in a loss curve).
usage: stress_determinism.py [cfg3] [passes]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
pkg = importlib.import_module("focused-attention-vit_amd")
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
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
ATOMIC = ("latent_proj", "cls_token", "pos_embed", "patch_embed")
ref = None
bad = 0
for it in range(N):
    model.zero_grad(set_to_none=True)
    loss = pkg.train.cross_entropy(model(images), labels)
    loss.backward()
    grads = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
    fin = all(bool(torch.isfinite(v).all()) for v in grads.values()) and bool(torch.isfinite(loss))
    if ref is None:
        ref = {n: v.clone() for n, v in grads.items()}
        ref_loss = loss.item()
        print("pass 0: loss", ref_loss, "finite", fin, flush=True)
        continue
    dev_names = [n for n, v in grads.items() if not any(a in n for a in ATOMIC) and not torch.equal(v, ref[n])]
    if dev_names or not fin or loss.item() != ref_loss:
        bad += 1
        order = list(grads)
        print(f"pass {it}: loss {loss.item()} finite {fin}; {len(dev_names)} deterministic gradients differ; "
              f"last in backward order = first produced: {dev_names[-3:]}", flush=True)
        for n in dev_names[-3:]:
            d = (grads[n].float() - ref[n].float())
            print("    ", n, "max|diff|", d.abs().max().item(), "nan", int(torch.isnan(grads[n]).sum()), "of", d.numel())
        if bad >= 3:
            break
print("done:", N, "passes,", bad, "deviating")
