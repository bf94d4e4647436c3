#!/usr/bin/env python3
"""Micro-benchmark of favit_gemm on the GEMM shapes of the ViT-MHLA-Small training step
(B=256, L=197): prints TFLOP/s per shape/layout; used under rocprofv3 --pmc for counters."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("focused-attention-vit_amd")
if os.environ.get("FAVIT_GEMM_DBG"):      # work-skipping switches exist only in the probe build (make -C .../csrc probe)
    pkg._abi.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe_build", "libfavit_probe.so")
K = pkg.kernels
dev = "cuda"
T, D = int(os.environ.get("TOKENS", 256 * 197)), 384
reps = int(os.environ.get("REPS", "20"))
only = os.environ.get("ONLY")

def run(name, fn, flops):
    if only and only not in name:
        return
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"{name:28s} {us:9.1f} us  {flops / us / 1e6:8.1f} TFLOP/s", flush=True)

bf = torch.float32 if os.environ.get("F32") else torch.bfloat16        # F32=1: the exact-fp32 kernels on the same shapes
def rnd(*s, dt=bf): return torch.randn(*s, device=dev).to(dt)
x = rnd(T, D); h = rnd(T, 4 * D)
wqkv = rnd(3 * D, D); w1 = rnd(4 * D, D); w2 = rnd(D, 4 * D); wp = rnd(D, D)
bq = torch.randn(3 * D, device=dev); b1 = torch.randn(4 * D, device=dev); b2 = torch.randn(D, device=dev)
res = torch.randn(T, D, device=dev)
o_qkv = torch.empty(T, 3 * D, device=dev, dtype=bf); o_h = torch.empty(T, 4 * D, device=dev, dtype=bf); o_pre = torch.empty_like(o_h)
o_d = torch.empty(T, D, device=dev); o_db = torch.empty(T, D, device=dev, dtype=bf)
dw1 = torch.empty(4 * D, D, device=dev); dw2 = torch.empty(D, 4 * D, device=dev); dwq = torch.empty(3 * D, D, device=dev)
db = torch.zeros(4 * D, device=dev)
A = pkg._abi
run("fwd qkv  NT K=384 N=1152", lambda: K.gemm(x, wqkv, o_qkv, T, 3 * D, D, D, D, 3 * D, bias=bq), 2 * T * 3 * D * D)
run("fwd fc1  NT K=384 N=1536 gelu", lambda: K.gemm(x, w1, o_h, T, 4 * D, D, D, D, 4 * D, bias=b1, act=A.ACT_GELU, aux_out=o_pre, ld_aux_out=4 * D), 2 * T * 4 * D * D)
run("fwd fc1  plain bf16 out", lambda: K.gemm(x, w1, o_h, T, 4 * D, D, D, D, 4 * D, bias=b1), 2 * T * 4 * D * D)
run("fwd fc1  gelu, no saved pre", lambda: K.gemm(x, w1, o_h, T, 4 * D, D, D, D, 4 * D, bias=b1, act=A.ACT_GELU), 2 * T * 4 * D * D)
run("fwd fc1  gelu+saved derivative", lambda: K.gemm(x, w1, o_h, T, 4 * D, D, D, D, 4 * D, bias=b1, act=A.ACT_GELU_SAVEGRAD, aux_out=o_pre, ld_aux_out=4 * D), 2 * T * 4 * D * D)
run("fwd fc1  gelu+saved derivative, dropout 0.1", lambda: K.gemm(x, w1, o_h, T, 4 * D, D, D, D, 4 * D, bias=b1, act=A.ACT_GELU_SAVEGRAD, aux_out=o_pre, ld_aux_out=4 * D, dropout_p=0.1, dropout_seed=1234), 2 * T * 4 * D * D)
run("fwd fc2  NT K=1536 N=384 res", lambda: K.gemm(h, w2, o_d, T, D, 4 * D, 4 * D, 4 * D, D, bias=b2, residual=res, ld_res=D), 2 * T * 4 * D * D)
run("fwd proj NT K=384 N=384 res", lambda: K.gemm(x, wp, o_d, T, D, D, D, D, D, bias=b2, residual=res, ld_res=D), 2 * T * D * D)
run("bwd dX   NN K=1152 N=384", lambda: K.gemm(o_qkv, wqkv, o_db, T, D, 3 * D, 3 * D, D, D, b_kmajor=False), 2 * T * 3 * D * D)
run("bwd dH   NN K=384 N=1536 dgelu", lambda: K.gemm(x, w2, o_h, T, 4 * D, D, D, 4 * D, 4 * D, b_kmajor=False, act=A.ACT_DGELU, aux_in=o_pre, ld_aux_in=4 * D), 2 * T * 4 * D * D)
run("bwd dH   plain bf16 out", lambda: K.gemm(x, w2, o_h, T, 4 * D, D, D, 4 * D, 4 * D, b_kmajor=False), 2 * T * 4 * D * D)
run("bwd dH   x saved derivative", lambda: K.gemm(x, w2, o_h, T, 4 * D, D, D, 4 * D, 4 * D, b_kmajor=False, act=A.ACT_MULAUX, aux_in=o_pre, ld_aux_in=4 * D), 2 * T * 4 * D * D)
run("bwd dH   x saved derivative, dropout 0.1", lambda: K.gemm(x, w2, o_h, T, 4 * D, D, D, 4 * D, 4 * D, b_kmajor=False, act=A.ACT_MULAUX, aux_in=o_pre, ld_aux_in=4 * D, dropout_p=0.1, dropout_seed=1234), 2 * T * 4 * D * D)
run("bwd dXn2 NN K=1536 N=384", lambda: K.gemm(h, w1, o_db, T, D, 4 * D, 4 * D, D, D, b_kmajor=False), 2 * T * 4 * D * D)
run("bwd dW1  TN [1536,384]", lambda: K.gemm(h, x, dw1, 4 * D, D, T, 4 * D, D, D, a_kmajor=False, b_kmajor=False, a_rowsum=db), 2 * T * 4 * D * D)
run("bwd dW2  TN [384,1536]", lambda: K.gemm(x, h, dw2, D, 4 * D, T, D, 4 * D, 4 * D, a_kmajor=False, b_kmajor=False, a_rowsum=db), 2 * T * 4 * D * D)
run("bwd dWqkv TN [1152,384]", lambda: K.gemm(o_qkv, x, dwq, 3 * D, D, T, 3 * D, D, D, a_kmajor=False, b_kmajor=False, a_rowsum=db), 2 * T * 3 * D * D)
big = 8192
a8 = rnd(big, big); b8 = rnd(big, big); c8 = torch.empty(big, big, device=dev, dtype=bf)
run("square 8192^3 NT", lambda: K.gemm(a8, b8, c8, big, big, big, big, big, big), 2 * big ** 3)

if os.environ.get("BASE"):
    # ViT-MHLA-Base 384/p16 (BASELINE.json configs[3] per-GPU shape: 64 images x 577 tokens, D = 768)
    Tb, Db = 64 * 577, 768
    xb = rnd(Tb, Db); hb = rnd(Tb, 4 * Db)
    wq = rnd(3 * Db, Db); wf1 = rnd(4 * Db, Db); wf2 = rnd(Db, 4 * Db)
    oq = torch.empty(Tb, 3 * Db, device=dev, dtype=bf); oh = torch.empty(Tb, 4 * Db, device=dev, dtype=bf)
    od = torch.empty(Tb, Db, device=dev); resb = torch.randn(Tb, Db, device=dev); odb = torch.empty(Tb, Db, device=dev, dtype=bf)
    run("base qkv  NT K=768 N=2304", lambda: K.gemm(xb, wq, oq, Tb, 3 * Db, Db, Db, Db, 3 * Db), 2 * Tb * 3 * Db * Db)
    run("base fc1  NT K=768 N=3072", lambda: K.gemm(xb, wf1, oh, Tb, 4 * Db, Db, Db, Db, 4 * Db, act=A.ACT_GELU), 2 * Tb * 4 * Db * Db)
    run("base fc2  NT K=3072 N=768 res", lambda: K.gemm(hb, wf2, od, Tb, Db, 4 * Db, 4 * Db, 4 * Db, Db, residual=resb, ld_res=Db), 2 * Tb * 4 * Db * Db)
    run("base dXn2 NN K=3072 N=768", lambda: K.gemm(hb, wf1, odb, Tb, Db, 4 * Db, 4 * Db, Db, Db, b_kmajor=False), 2 * Tb * 4 * Db * Db)
    run("base dX   NN K=2304 N=768", lambda: K.gemm(oq, wq, odb, Tb, Db, 3 * Db, 3 * Db, Db, Db, b_kmajor=False), 2 * Tb * 3 * Db * Db)
    run("base dH   NN K=768 N=3072", lambda: K.gemm(xb, wf2, oh, Tb, 4 * Db, Db, Db, 4 * Db, 4 * Db, b_kmajor=False), 2 * Tb * 4 * Db * Db)

if os.environ.get("LIBCMP"):
    # The vendor library on the same shapes (torch -> hipBLASLt / rocBLAS; plain bf16 output, bias only where the library
    # fuses it): a yardstick for the main loops, not part of the product path.
    import torch.nn.functional as F
    bqh, b1h, b2h = bq.to(bf), b1.to(bf), b2.to(bf)
    run("lib  qkv  linear+bias", lambda: F.linear(x, wqkv, bqh), 2 * T * 3 * D * D)
    run("lib  fc1  linear+bias", lambda: F.linear(x, w1, b1h), 2 * T * 4 * D * D)
    run("lib  fc2  linear+bias", lambda: F.linear(h, w2, b2h), 2 * T * 4 * D * D)
    run("lib  proj linear+bias", lambda: F.linear(x, wp, b2h), 2 * T * D * D)
    run("lib  dX   qkv @ Wqkv", lambda: torch.mm(o_qkv, wqkv), 2 * T * 3 * D * D)
    run("lib  dH   x @ W2", lambda: torch.mm(x, w2), 2 * T * 4 * D * D)
    run("lib  dXn2 h @ W1", lambda: torch.mm(h, w1), 2 * T * 4 * D * D)
    run("lib  dW1  h^T @ x", lambda: torch.mm(h.t(), x), 2 * T * 4 * D * D)
    run("lib  dW2  x^T @ h", lambda: torch.mm(x.t(), h), 2 * T * 4 * D * D)
    run("lib  dWqkv qkv^T @ x", lambda: torch.mm(o_qkv.t(), x), 2 * T * 3 * D * D)
    run("lib  square 8192^3 NT", lambda: F.linear(a8, b8), 2 * big ** 3)
    if os.environ.get("BASE"):
        run("lib  base qkv", lambda: F.linear(xb, wq), 2 * Tb * 3 * Db * Db)
        run("lib  base fc1", lambda: F.linear(xb, wf1), 2 * Tb * 4 * Db * Db)
        run("lib  base fc2", lambda: F.linear(hb, wf2), 2 * Tb * 4 * Db * Db)
        run("lib  base dXn2 h @ W1", lambda: torch.mm(hb, wf1), 2 * Tb * 4 * Db * Db)
        run("lib  base dX  qkv @ Wq", lambda: torch.mm(oq, wq), 2 * Tb * 3 * Db * Db)
        run("lib  base dH  x @ W2", lambda: torch.mm(xb, wf2), 2 * Tb * 4 * Db * Db)
