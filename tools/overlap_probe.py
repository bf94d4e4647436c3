#!/usr/bin/env python3
"""Probe: does an HBM-store stream (fill) overlap with the L2-bound GEMM main loop when both run
concurrently?  Run with FAVIT_GEMM_DBG=1 (main loop only) to isolate the loop."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("focused-attention-vit_amd")
if os.environ.get("FAVIT_GEMM_DBG"):
    pkg._abi.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe_build", "libfavit_probe.so")
K = pkg.kernels
dev = "cuda"
T, D = 256 * 197, 384
bf = torch.bfloat16
x = torch.randn(T, D, device=dev).to(bf)
w = torch.randn(4 * D, D, device=dev).to(bf)
o = torch.empty(T, 4 * D, device=dev, dtype=bf)
sink = torch.empty(T * 4 * D * 2, device=dev, dtype=bf)      # 310 MB
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
reps = 40

def gemm():
    K.gemm(x, w, o, T, 4 * D, D, D, D, 4 * D)

import ctypes
_pl = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libprobe.so"))
_pl.probe_fill.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
_pl.probe_read.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
flag = torch.zeros(4, device=dev, dtype=torch.int32)
NB = int(os.environ.get("PROBE_BLOCKS", "2048"))
def cur(): return torch.cuda.current_stream().cuda_stream
def fill():
    sink.fill_(1.0)
def fill_plain():
    _pl.probe_fill(sink.data_ptr(), sink.numel() * 2, 0, NB, cur())
def fill_nt():
    _pl.probe_fill(sink.data_ptr(), sink.numel() * 2, 1, NB, cur())
def read_nt():
    _pl.probe_read(sink.data_ptr(), sink.numel() * 2, flag.data_ptr(), NB, cur())

def timed(fa, fb):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    s1.wait_event(e0); s2.wait_event(e0)
    with torch.cuda.stream(s1):
        for _ in range(reps):
            if fa: fa()
        ea = torch.cuda.Event(); ea.record()
    with torch.cuda.stream(s2):
        for _ in range(reps):
            if fb: fb()
        eb = torch.cuda.Event(); eb.record()
    torch.cuda.current_stream().wait_event(ea); torch.cuda.current_stream().wait_event(eb)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps

for _ in range(3):
    with torch.cuda.stream(s1): gemm()
    with torch.cuda.stream(s2): fill()
g = timed(gemm, None)
print(f"gemm alone   {g:8.1f} us/iter")
for name, fn in (("torch fill", fill), ("plain fill", fill_plain), ("nt fill", fill_nt), ("nt read", read_nt)):
    for _ in range(2): fn()
    a = timed(None, fn); c = timed(gemm, fn)
    print(f"{name:12s} alone {a:8.1f}  concurrent {c:8.1f}  (sum {g + a:8.1f}, max {max(g, a):8.1f})")
