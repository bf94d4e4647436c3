#!/usr/bin/env python3
"""Stage times of the device SLIC (features / k-means / connectivity) at the SPPP shapes: 128 images of 224x224,
16 segments.  Smooth random images (bicubic-upsampled noise), as in tests/test_slic.py."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("focused-attention-vit_amd")
if os.environ.get("FAVIT_SLIC_DBG"):      # probe build: work-skipping switches of the assignment kernel (timing only)
    pkg._abi.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe_build", "libfavit_probe.so")
K, A = pkg.kernels, pkg._abi
dev = "cuda"
B, H, W, nseg = int(os.environ.get("B", "128")), 224, 224, 16
g = torch.Generator(device=dev).manual_seed(3)
img = torch.nn.functional.interpolate(torch.rand(B, 3, 14, 14, device=dev, generator=g), size=(H, W), mode="bicubic").clamp(0, 1).contiguous()
def t(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print(f"slic total ({B} images {H}x{W}, {nseg} segments): {t(lambda: K.slic(img, nseg, 10.0)):.2f} ms")
if hasattr(K, "slic_stage_times"):
    for name, ms in K.slic_stage_times(img, nseg, 10.0): print(f"   {name:10s} {ms:.2f} ms")
