#!/usr/bin/env python3
"""Grouped weight-gradient launch (dW = dY^T X of the four Linear layers of a block, for NB blocks in one launch) at
the benchmark shapes.  One process per setting (the library reads its knobs once):
    FAVIT_GROUPED_S=n   force n K-splits          NB=4      blocks per launch (default 1)
    CFG=cfg2|cfg4|cfg3|cfg1                        FAVIT_LIB=path   another build of the library (A/B against HEAD)
    FAVIT_GEMM_DBG=1    probe build: main loop only (no epilogue; timing only)
Prints the launch time, the time per block and the K-splits the library chose."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("focused-attention-vit_amd")
if os.environ.get("FAVIT_LIB"):
    pkg._abi.LIB_PATH = os.path.abspath(os.environ["FAVIT_LIB"])
    import ctypes
    _probe = ctypes.CDLL(pkg._abi.LIB_PATH)
    for _n in list(pkg._abi._SIGS):                  # an older build: bind only what it exports
        if not hasattr(_probe, _n):
            del pkg._abi._SIGS[_n]
elif os.environ.get("FAVIT_GEMM_DBG"):
    pkg._abi.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe_build", "libfavit_probe.so")
K = pkg.kernels
dev = "cuda"
CFGS = {"cfg2": (256 * 197, 384), "cfg4": (64 * 577, 768), "cfg3": (128 * 17, 384), "cfg1": (64 * 65, 192)}
NB = int(os.environ.get("NB", "1"))
for name in os.environ.get("CFG", "cfg2,cfg4").split(","):
    T, D = CFGS[name]
    shapes = [(D, 4 * D), (4 * D, D), (D, D), (3 * D, D)]
    probs = []
    for _ in range(NB):
        for N, Kd in shapes:
            probs.append((torch.randn(T, N, device=dev).bfloat16(), torch.randn(T, Kd, device=dev).bfloat16(),
                          torch.zeros(N, Kd, device=dev), torch.zeros(N, device=dev), True))
    old = "favit_gemm_grouped_last_splits" not in pkg._abi._SIGS
    chunks = [probs[i:i + 4] for i in range(0, len(probs), 4)] if (old or os.environ.get("PER_BLOCK")) else [probs]
    for _ in range(3):
        for c in chunks:
            assert K.gemm_grouped_tn(c)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        for c in chunks:
            K.gemm_grouped_tn(c)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    fl = NB * sum(2.0 * T * n * k for n, k in shapes)
    sp = "?" if old else int(pkg._abi.lib().favit_gemm_grouped_last_splits())
    print(f"{name}: T={T} D={D} blocks/launch={NB if len(chunks) == 1 else 1} (x{len(chunks)} launches) splits={sp}: "
          f"{us:.1f} us = {us / NB:.1f} us per block, {fl / us / 1e6:.0f} TF", flush=True)
