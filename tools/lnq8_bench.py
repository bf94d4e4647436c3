#!/usr/bin/env python3
"""LayerNorm passes with and without the fused fp8 quantisation (favit_layernorm_*_q8) at the cfg4 shape: the two-pass
form (LayerNorm, then favit_fp8_quantize of its bf16 output) against the one-pass form, forward and backward."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("focused-attention-vit_amd")
K = pkg.kernels
rows, D = int(os.environ.get("ROWS", 64 * 577)), int(os.environ.get("D", 768))
dev = "cuda"
x = torch.randn(rows, D, device=dev); g = torch.ones(D, device=dev); b = torch.zeros(D, device=dev)
dy = torch.randn(rows, D, device=dev).to(torch.bfloat16); dres = torch.randn(rows, D, device=dev)
e4, e5 = torch.float8_e4m3fn, torch.float8_e5m2

def t(name, fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:48s} {e0.elapsed_time(e1) * 1e3 / n:8.1f} us", flush=True)

h = [K.Fp8History(torch.device(dev)) for _ in range(4)]
y, mu, rs = K.layernorm_fwd(x, D, g, b, rows, D, torch.bfloat16)
for hh in h: K.fp8_quantize(y, e4, hist=hh)          # first call of a site: measures first
t("ln_fwd", lambda: K.layernorm_fwd(x, D, g, b, rows, D, torch.bfloat16))
t("fp8_quantize (e4m3, delayed)", lambda: K.fp8_quantize(y, e4, hist=h[0]))
t("ln_fwd + fp8_quantize", lambda: K.fp8_quantize(K.layernorm_fwd(x, D, g, b, rows, D, torch.bfloat16)[0], e4, hist=h[0]))
t("ln_fwd_q8 (one pass)", lambda: K.layernorm_fwd(x, D, g, b, rows, D, torch.bfloat16, q8=(e4, h[1])))
t("ln_bwd", lambda: K.layernorm_bwd(dy, x, D, g, mu, rs, rows, D, dres=dres, want_lp=True))
t("ln_bwd + fp8_quantize", lambda: K.fp8_quantize(K.layernorm_bwd(dy, x, D, g, mu, rs, rows, D, dres=dres, want_lp=True)[1], e5, hist=h[2]))
t("ln_bwd_q8 (one pass)", lambda: K.layernorm_bwd(dy, x, D, g, mu, rs, rows, D, dres=dres, want_lp=True, q8=(e5, h[3])))
