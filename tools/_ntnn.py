import importlib, os, sys
sys.path.insert(0, "/root/repo")
import torch
pkg = importlib.import_module("focused-attention-vit_amd")
K = pkg.kernels
T, D = 256 * 197, 384
bf = torch.bfloat16
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def rnd(*s): return torch.randn(*s, device="cuda").to(bf)
for M, N, Kd in ((T, D, 4 * D), (T, D, 3 * D), (T, D, D), (T, 4 * D, D)):
    a = rnd(M, Kd); bn = rnd(N, Kd); bm = rnd(Kd, N); c = torch.empty(M, N, device="cuda", dtype=bf)
    nt = t(lambda: K.gemm(a, bn, c, M, N, Kd, Kd, Kd, N, b_kmajor=True))
    nn = t(lambda: K.gemm(a, bm, c, M, N, Kd, Kd, N, N, b_kmajor=False))
    print(f"N={N:5d} K={Kd:5d}: NT {nt:7.1f} us   NN {nn:7.1f} us   ratio {nn / nt:.3f}", flush=True)
