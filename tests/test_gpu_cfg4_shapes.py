"""GPU parity of the kernels `bench.py --config cfg4` times, at THEIR shapes (BASELINE.json configs[3]:
ViT-MHLA-Base 384/p16, 64 images x 577 tokens = 36,928 token rows, D = 768), and of the fp8 delayed-scaling path
that configuration's fp8 line runs.

* the grouped weight-gradient launch at D = 768, T = 36,928 (and a small qualifying T), slab and atomic
  reductions, against an fp64 dY^T X;
* the 256x256-tile kernel ("p7": the D = 768 forward GEMMs), the 256x128-tile kernel ("p4": the input-gradient
  GEMMs) and the fp8 256x128 kernel at M = 36,928 with N in {768, 2304, 3072}, K in {768, 2304, 3072} and the
  epilogues the model uses there -- each asserting through favit_gemm_last_kernel which kernel family really ran;
* the whole cfg4 step at B = 64 (the golden image tiled 64x), bf16 and fp8, against the B = 1 HIP gradients
  element-wise and the reference's golden gradient norms (configs.npz: cfg4/gnorm/*);
* fp8 delayed scaling (kernels.Fp8History: this call's scale = the amax the site's PREVIOUS call measured):
  scale rotation, saturation beyond the previous amax and the amax hand-over against a torch restatement, and a
  four-step training trajectory in fp8 mode against the oracle's fp32 gradients at the same weights.  The
  reference is fp32 only: the fp8 tolerances are OURS (tests/test_configs_golden.py: FP8_TOL)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_l2
from test_gpu_fullsize import _block_problems, _rand

pytestmark = pytest.mark.gpu
DEV = "cuda"
T_CFG4 = 64 * 577
CF = load_golden("configs.npz")
FP8_TOL = dict(logits=0.15, loss=4e-2, gnorm=0.25)


@pytest.fixture(scope="module")
def K(favit):
    return favit.kernels


def _last(favit):
    return favit._abi.lib().favit_gemm_last_kernel().decode()


# ------------------------------------------------------------------ grouped weight gradients at D = 768
@pytest.mark.parametrize("slabs", [True, False])
@pytest.mark.parametrize("T,accumulate", [(1280, False), (T_CFG4, False), (T_CFG4, True)])
def test_gemm_grouped_tn_block_problems_d768(K, T, accumulate, slabs):
    gen = torch.Generator(device=DEV).manual_seed(T + 7 * int(accumulate))
    probs, refs = _block_problems(T, gen, accumulate, D=768)
    assert K.gemm_grouped_tn(probs, use_workspace=slabs), "the grouped launch must take the ViT-Base block at T % 32 == 0"
    torch.cuda.synchronize()
    for (dy, x, dw, db, _), (ref_w, ref_b) in zip(probs, refs):
        assert torch.isfinite(dw).all()
        assert rel_l2(dw, ref_w) < 2e-5, (tuple(dw.shape), rel_l2(dw, ref_w))
        assert rel_l2(db, ref_b) < 2e-5, (tuple(dw.shape), rel_l2(db, ref_b))


# ------------------------------------------------------------------ forward / input-gradient GEMMs at M = 36,928
# (epilogue, N, K): the four Linear layers of a ViT-Base block, forward (B k-major) and input gradient (B mn-major)
FWD = [("bias_bf16", 2304, 768), ("gelu_savegrad", 3072, 768), ("res_f32", 768, 768), ("res_f32", 768, 3072)]
BWD = [("plain_bf16", 768, 2304), ("plain_bf16", 768, 3072), ("mulaux", 3072, 768), ("plain_bf16", 768, 768)]


def _epilogue_case(K, abi, a, b, bk, epi, M, N, Kd, gen, scale_a=None, scale_b=None, base=None, tol_lp=1e-2, tol_f32=2e-5):
    ldb = Kd if bk else N
    kw = dict(b_kmajor=bk, scale_a=scale_a, scale_b=scale_b)
    bias = _rand((N,), torch.float32, gen)
    if epi == "bias_bf16":
        out = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
        K.gemm(a, b, out, M, N, Kd, Kd, ldb, N, bias=bias, **kw)
        assert rel_l2(out.float(), base + bias) < tol_lp
    elif epi == "plain_bf16":
        out = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
        K.gemm(a, b, out, M, N, Kd, Kd, ldb, N, **kw)
        assert rel_l2(out.float(), base) < tol_lp
    elif epi == "gelu_savegrad":
        out = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
        aux = torch.empty_like(out)
        K.gemm(a, b, out, M, N, Kd, Kd, ldb, N, bias=bias, act=abi.ACT_GELU_SAVEGRAD, aux_out=aux, ld_aux_out=N, **kw)
        u = (base + bias).requires_grad_(True)
        h = torch.nn.functional.gelu(u)
        h.sum().backward()
        assert rel_l2(out.float(), h.detach()) < tol_lp
        assert rel_l2(aux.float(), u.grad) < tol_lp
    elif epi == "mulaux":
        aux = _rand((M, N), torch.bfloat16, gen)
        out = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
        K.gemm(a, b, out, M, N, Kd, Kd, ldb, N, act=abi.ACT_MULAUX, aux_in=aux, ld_aux_in=N, **kw)
        assert rel_l2(out.float(), base * aux.float()) < tol_lp
    else:
        res = _rand((M, N), torch.float32, gen)
        out = torch.empty((M, N), dtype=torch.float32, device=DEV)
        K.gemm(a, b, out, M, N, Kd, Kd, ldb, N, bias=bias, residual=res, ld_res=N, **kw)
        assert rel_l2(out, base + bias + res) < tol_f32


@pytest.mark.parametrize("epi,N,Kd", FWD)
def test_bf16_forward_gemms_at_cfg4_m_run_the_256x256_kernel(K, favit, epi, N, Kd):
    M = T_CFG4
    gen = torch.Generator(device=DEV).manual_seed(N + Kd)
    a = _rand((M, Kd), torch.bfloat16, gen)
    b = _rand((N, Kd), torch.bfloat16, gen, 0.05)
    _epilogue_case(K, favit._abi, a, b, True, epi, M, N, Kd, gen, base=a.float() @ b.float().t())
    assert _last(favit) == "p7", _last(favit)


@pytest.mark.parametrize("epi,N,Kd", BWD)
def test_bf16_input_gradient_gemms_at_cfg4_m_run_the_256x128_kernel(K, favit, epi, N, Kd):
    M = T_CFG4
    gen = torch.Generator(device=DEV).manual_seed(3 * N + Kd)
    a = _rand((M, Kd), torch.bfloat16, gen)
    b = _rand((Kd, N), torch.bfloat16, gen, 0.05)
    _epilogue_case(K, favit._abi, a, b, False, epi, M, N, Kd, gen, base=a.float() @ b.float())
    assert _last(favit) == "p4", _last(favit)


@pytest.mark.parametrize("afmt,epi,N,Kd", [(torch.float8_e4m3fn, *c) for c in FWD] + [(torch.float8_e5m2, *c) for c in BWD])
def test_fp8_gemms_at_cfg4_m(K, favit, afmt, epi, N, Kd):
    """e4m3 x e4m3 forward GEMMs and e5m2 x e4m3 input-gradient GEMMs (B = the transposed weight copy the
    quantise pass writes) against an fp32 matmul of the dequantised operands."""
    M = T_CFG4
    gen = torch.Generator(device=DEV).manual_seed(5 * N + Kd)
    a = _rand((M, Kd), torch.bfloat16, gen)
    b = _rand((N, Kd), torch.bfloat16, gen, 0.05)
    aq, _, sa = K.fp8_quantize(a, afmt)
    bq, _, sb = K.fp8_quantize(b, torch.float8_e4m3fn)
    base = (aq.float() * sa) @ (bq.float() * sb).t()
    _epilogue_case(K, favit._abi, aq, bq, True, epi, M, N, Kd, gen, scale_a=sa, scale_b=sb, base=base, tol_lp=4e-3)
    assert _last(favit) == "p4", _last(favit)


# ------------------------------------------------------------------ the whole cfg4 step at B = 64
def _base384(favit):
    torch.manual_seed(1234)
    m = favit.models.vit_mhla.VisionTransformerMHLA(img_size=384, patch_size=16, num_classes=1000, embed_dim=768,
                                                    depth=12, num_heads=12, window_size=7, use_mhla=True)
    x = torch.randn(1, 3, 384, 384)
    y = torch.randint(0, 1000, (1,))
    assert abs(x.double().sum().item() - float(CF["cfg4/x_sum"])) < 1e-6 and torch.equal(y, torch.from_numpy(CF["cfg4/y"]))
    return m, x, y


@pytest.mark.parametrize("mode,tol_elem,tol_gn,tol_logits", [("bf16", 1.5e-2, 5e-2, 2e-2), ("fp8", 0.25, FP8_TOL["gnorm"], FP8_TOL["logits"])])
def test_full_size_cfg4_forward_backward_matches_golden(favit, K, mode, tol_elem, tol_gn, tol_logits):
    """B = 64 forward + backward through bench.py's flow (flat gradient buffers, grouped weight gradients) == the
    B = 1 HIP result == the reference's golden run.  fp8: the B = 1 pass is every site's first call (it measures its
    own amax), the B = 64 pass the second (delayed: activations reuse the B = 1 amax, which the tiled batch shares;
    the gradients of a mean loss over 64 images are 64x smaller than the B = 1 amax their scale comes from, so they
    are quantised 6 binades lower in e5m2's range -- the element-wise tolerance of the fp8 case is therefore the fp8
    gradient-noise level (measured: up to 0.17 on the 768-element cls_token gradient), not a rounding-order one)."""
    favit.set_compute_dtype(mode)
    try:
        m, x, y = _base384(favit)
        m.to(DEV).train()
        x, y = x.to(DEV), y.to(DEV)
        logits1 = m(x)
        assert rel_l2(logits1.detach().float().cpu(), CF["cfg4/logits"]) < tol_logits
        favit.train.cross_entropy(logits1, y).backward()
        g1 = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
        for p in m.parameters():
            p.grad = None
        opt = favit.train.FusedAdamW(favit.train.param_groups(m, lr=1e-4), distributed=False)
        xb, yb = x.repeat(64, 1, 1, 1).contiguous(), y.repeat(64).contiguous()
        opt.zero_grad()
        K.GEMM_TRACE = []
        try:
            logits = m(xb)
            loss = favit.train.cross_entropy(logits, yb)
            loss.backward()
            torch.cuda.synchronize()
            keys = {(t[3], t[5]) for t in K.GEMM_TRACE}
        finally:
            K.GEMM_TRACE = None
        assert ("bf16_MM_of32_grouped", "grouped_tn") in keys, sorted(keys)
        if mode == "bf16":
            assert any(kern == "p7" for _, kern in keys), f"no 256x256-tile launch in the cfg4 step: {sorted(keys)}"
        else:
            assert any(k.startswith("fp8") for k, _ in keys), sorted(keys)
        assert rel_l2(logits[63:].detach().float().cpu(), CF["cfg4/logits"]) < tol_logits
        rl = float(CF["cfg4/loss"])
        assert abs(loss.item() - rl) < (FP8_TOL["loss"] if mode == "fp8" else tol_logits) * abs(rl)
        worst_e, worst_n = 0.0, 0.0
        for k, p in m.named_parameters():
            assert p.grad is not None and torch.isfinite(p.grad).all(), k
            e = rel_l2(p.grad, g1[k])
            worst_e = max(worst_e, e)
            # fp8: the small reduction-type gradients (64-element latent_proj bias, 768-element cls_token: sums over every
            # token with heavy cancellation) carry the most quantisation noise -- measured 0.17 .. 0.25 -- twice the bound
            te = tol_elem * (2.0 if (mode == "fp8" and p.numel() < 4096) else 1.0)
            assert e < te, f"{k}: B=64 vs B=1 gradient rel-L2 {e}"
            r = float(CF[f"cfg4/gnorm/{k}"])
            n = abs(p.grad.norm().item() - r) / max(r, 1e-12)
            worst_n = max(worst_n, n)
            assert n < tol_gn, f"{k}: gradient norm off the reference's by {n}"
        print(f"[cfg4 {mode}] worst element-wise rel-L2 vs B=1: {worst_e:.2e}; worst gradient-norm deviation vs golden: {worst_n:.2e}")
    finally:
        favit.set_compute_dtype("fp32")
        favit.functional.clear_lp_mirrors()


# ------------------------------------------------------------------ fp8 delayed scaling
@pytest.mark.parametrize("src_dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("fmt", [torch.float8_e4m3fn, torch.float8_e5m2])
def test_fp8_delayed_scaling_rotation_and_saturation(K, fmt, src_dtype):
    """One quantisation site fed tensors whose amax grows and shrinks: call i must use the scale of tensor i-1's amax
    (call 0: its own), saturate what exceeds it at +-max, and hand tensor i's amax to call i+1 -- bit-exact against
    a torch restatement of favit_fp8_quantize (csrc/fp8.hip)."""
    fmax = 448.0 if fmt == torch.float8_e4m3fn else 57344.0
    hist = K.Fp8History(torch.device(DEV))
    gen = torch.Generator(device=DEV).manual_seed(17)
    prev_amax = None
    saturated = []
    for i, s in enumerate([1.0, 2.0, 16.0, 0.5, 0.03, 4.0, 4.0]):
        rows, cols = (300, 200) if i % 2 == 0 else (577, 192)       # the site sees different shapes too
        t = (_rand((rows, cols), torch.float32, gen) * s).to(src_dtype)
        q, qt, sinv = K.fp8_quantize(t, fmt, want=True, want_t=True, hist=hist)
        own = np.float32(t.float().abs().max().item())
        am = own if prev_amax is None else prev_amax
        assert sinv.item() == float(am / np.float32(fmax)), (i, sinv.item(), am)
        scale = torch.tensor(float(np.float32(fmax) / am), dtype=torch.float32, device=DEV)
        ref = (t.float() * scale).clamp(-fmax, fmax).to(fmt)
        assert torch.equal(q.view(torch.uint8), ref.view(torch.uint8)), i
        assert torch.equal(qt[:, :rows].view(torch.uint8), ref.t().contiguous().view(torch.uint8)), i
        saturated.append(int((ref.float().abs() == fmax).sum().item()))
        # dequantised values reproduce the tensor to the format's precision wherever nothing saturated
        ok = (t.float().abs() * scale) < fmax
        deq = q.float() * sinv
        if ok.any() and s >= 0.5:
            assert rel_l2(deq[ok], t.float()[ok]) < (0.04 if fmt == torch.float8_e4m3fn else 0.08), i
        prev_amax = own
        assert hist.calls == i + 1
    # growth 2 -> 16 saturates many values (call 2 runs on call 1's amax); a shrinking tensor saturates none
    assert saturated[2] > 1000 and saturated[3] <= 1 and saturated[4] <= 1, saturated
    assert saturated[6] >= 1                  # equal-scale successor: only its own maximum may touch the limit


def test_fp8_training_trajectory_against_oracle(favit):
    """Four optimizer steps in fp8 mode (delayed scaling from the second step on; a fresh batch at a different
    input scale every step) with the fused AdamW: at every step logits, loss and per-parameter gradient norms
    are within the stated fp8 tolerance of the oracle's fp32 results AT THE SAME WEIGHTS, and the fp8 loss
    trajectory stays within 2 % of the bf16 trajectory from the same start (experiments/mhla_pretrained.py:363-367
    is the loop this mirrors)."""
    from oracle import favit_oracle as O

    def build():
        torch.manual_seed(31)
        m = favit.models.vit_mhla.VisionTransformerMHLA(img_size=32, patch_size=4, num_classes=10, embed_dim=128, depth=3,
                                                        num_heads=2, window_size=7, use_mhla=True).to(DEV).train()
        opt = favit.train.FusedAdamW(favit.train.param_groups(m, lr=1e-3), lr=1e-3, weight_decay=0.05, distributed=False)
        return m, opt

    gen = torch.Generator(device=DEV).manual_seed(9)
    xs = [torch.randn(32, 3, 32, 32, device=DEV, generator=gen) * s for s in (1.0, 2.5, 0.4, 1.5)]
    ys = [torch.randint(0, 10, (32,), device=DEV, generator=gen) for _ in range(4)]
    losses = {}
    try:
        for mode in ("bf16", "fp8"):
            favit.set_compute_dtype(mode)
            m, opt = build()
            names = [k for k, _ in m.named_parameters()]
            losses[mode] = []
            for step, (x, y) in enumerate(zip(xs, ys)):
                opt.zero_grad()
                logits = m(x)
                loss = favit.train.cross_entropy(logits, y)
                loss.backward()
                losses[mode].append(loss.item())
                if mode == "fp8":
                    sd = {k: v.detach().float().cpu().clone().requires_grad_(v.is_floating_point())
                          for k, v in m.state_dict().items()}
                    ref_logits = O.vit_mhla_forward(x.cpu(), sd, 4, 2, 7, True)
                    ref_loss = O.cross_entropy(ref_logits, y.cpu())
                    ref_loss.backward()
                    err = rel_l2(logits.detach().float().cpu(), ref_logits.detach())
                    assert err < FP8_TOL["logits"], (step, err)
                    assert abs(loss.item() - ref_loss.item()) < FP8_TOL["loss"] * abs(ref_loss.item()), (step, loss.item(), ref_loss.item())
                    worst = 0.0
                    for k, p in zip(names, m.parameters()):
                        r = sd[k].grad.norm().item()
                        worst = max(worst, abs(p.grad.norm().item() - r) / max(r, 1e-10))
                        # direction, not only length: the fp8 gradient points where the fp32 gradient points
                        if p.grad.numel() >= 128:
                            cos = torch.nn.functional.cosine_similarity(p.grad.flatten().cpu(), sd[k].grad.flatten(), dim=0).item()
                            assert cos > 0.9, (step, k, cos)
                    assert worst < FP8_TOL["gnorm"], (step, worst)
                opt.step()
            favit.functional.clear_lp_mirrors()
        for a, b in zip(losses["bf16"], losses["fp8"]):
            assert abs(a - b) < 2e-2 * abs(a), losses
    finally:
        favit.set_compute_dtype("fp32")
        favit.functional.clear_lp_mirrors()


def test_fp8_steps_are_identical_with_and_without_the_fused_layernorm_quantisation(favit, monkeypatch):
    """fp8 mode, four steps on fixed weights (learning rate 0: the delayed-scaling histories still rotate, every step
    sees a batch of another magnitude): with the LayerNorm passes quantising their bf16 outputs (favit_layernorm_*_q8,
    the default) and with the stand-alone quantising passes (FAVIT_FP8_NO_LNQ8=1) the losses and the LayerNorm
    gradients of every step are BIT-IDENTICAL and every other gradient agrees to what fp32 atomics allow -- the fused
    form changes where the bytes are produced, not the bytes; and the fused form really ran (four stand-alone passes
    per block fewer from the second step on)."""
    gen = torch.Generator(device=DEV).manual_seed(19)
    xs = [torch.randn(16, 3, 32, 32, device=DEV, generator=gen) * s for s in (1.0, 2.5, 0.4, 1.5)]
    ys = [torch.randint(0, 10, (16,), device=DEV, generator=gen) for _ in range(4)]
    K = favit.kernels
    out, calls = {}, {}
    real = K.fp8_quantize
    try:
        favit.set_compute_dtype("fp8")
        for mode in ("fused", "two_pass"):
            if mode == "two_pass":
                monkeypatch.setenv("FAVIT_FP8_NO_LNQ8", "1")
            else:
                monkeypatch.delenv("FAVIT_FP8_NO_LNQ8", raising=False)
            n = [0]
            def counted(*a, _n=n, **kw):
                _n[0] += 1
                return real(*a, **kw)
            monkeypatch.setattr(K, "fp8_quantize", counted)
            torch.manual_seed(31)
            m = favit.models.vit_mhla.VisionTransformerMHLA(img_size=32, patch_size=4, num_classes=10, embed_dim=128, depth=3,
                                                            num_heads=2, window_size=7, use_mhla=True, dropout=0.1).to(DEV).train()
            opt = favit.train.FusedAdamW(favit.train.param_groups(m, lr=0.0), lr=0.0, weight_decay=0.0, distributed=False)
            steps, per_step = [], []
            for x, y in zip(xs, ys):
                before = n[0]
                loss = favit.train.train_step(m, x, y, opt).item()
                per_step.append(n[0] - before)
                steps.append((loss, {k: p.grad.detach().clone() for k, p in m.named_parameters()}))
            out[mode], calls[mode] = steps, per_step
    finally:
        monkeypatch.setattr(K, "fp8_quantize", real)
        favit.set_compute_dtype("fp32")
        favit.functional.clear_lp_mirrors()
    for i, ((la, ga), (lb, gb)) in enumerate(zip(out["fused"], out["two_pass"])):
        assert la == lb, (i, la, lb)
        for k, a in ga.items():
            if ".norm" in k:                       # partial sums folded in a fixed order: no atomics anywhere upstream
                assert torch.equal(a, gb[k]), (i, k)
            else:
                assert rel_l2(a, gb[k]) < 1e-5, (i, k)
    # first step: every site measures first (two passes); afterwards 4 stand-alone passes per block fewer, except the
    # stream gradient entering the last block from the head (not produced by a LayerNorm backward of the encoder)
    assert calls["fused"][0] == calls["two_pass"][0]
    assert calls["two_pass"][-1] - calls["fused"][-1] == 4 * 3 - 1, (calls["fused"], calls["two_pass"])


def test_dynamic_lds_limit_grows_with_later_larger_requests():
    """favit_ensure_dyn_lds used to keep the FIRST dynamic-LDS size requested per (kernel, device): a later, larger
    request of the same instantiation (sdpa fp32 hd 256 -> 384, bf16 512 -> 768; MHLA backward hd 128 at L = 197
    then 64) failed with FAVIT_ERR_LAUNCH depending on call order.  Needs a fresh process (the limit is per-process
    state): tests/_lds_growth_probe.py."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "_lds_growth_probe.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "LDS_GROWTH_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
