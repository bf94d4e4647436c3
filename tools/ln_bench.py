import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("focused-attention-vit_amd"); K = pkg.kernels
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M, D in ((50432, 384), (36928, 768), (2176, 384)):
    x = torch.randn(M, D, device="cuda"); g = torch.randn(D, device="cuda"); b = torch.randn(D, device="cuda")
    print(M, D, f"{t(lambda: K.layernorm_fwd(x, D, g, b, M, D, torch.bfloat16)):.1f} us")
