"""GPU unit tests of the libfavit kernels through the C ABI (ctypes), each compared with a
plain fp32 PyTorch restatement of the same op on the same seeded inputs.
fp32 path: tolerance 2e-5 rel-L2 (exact-fp32 MFMA); bf16 path: inputs are rounded to bf16
first, then 1e-2 (bf16 output rounding)."""
import math

import numpy as np
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _tol(dtype):
    return 2e-5 if dtype == torch.float32 else 1e-2


@pytest.fixture(scope="module")
def K(favit):
    return favit.kernels


def _rand(shape, dtype, gen):
    return torch.randn(shape, generator=gen, device=DEV, dtype=torch.float32).to(dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("ak,bk", [(True, True), (True, False), (False, True), (False, False)])
@pytest.mark.parametrize("M,N,Kd", [(256, 384, 128), (130, 70, 50), (128, 128, 64), (1, 64, 64), (197, 1000, 384),
                                    (64, 10, 192), (300, 200, 8)])
def test_gemm_layouts(K, dtype, ak, bk, M, N, Kd):
    g = torch.Generator(device=DEV).manual_seed(M * 7 + N * 3 + Kd)
    A = _rand((M, Kd) if ak else (Kd, M), dtype, g)
    B = _rand((N, Kd) if bk else (Kd, N), dtype, g)
    C = torch.empty((M, N), dtype=torch.float32, device=DEV)
    K.gemm(A, B, C, M, N, Kd, A.stride(0), B.stride(0), N, a_kmajor=ak, b_kmajor=bk)
    Af = A.float() if ak else A.float().t()
    Bf = B.float() if bk else B.float().t()
    ref = Af.double() @ Bf.double().t()
    assert rel_l2(C, ref) < (2e-5 if dtype == torch.float32 else 2e-5), "fp32-accumulated product of exact inputs"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_epilogues(K, favit, dtype):
    from importlib import import_module
    abi = favit._abi
    g = torch.Generator(device=DEV).manual_seed(5)
    M, N, Kd = 256, 384, 128
    A, W = _rand((M, Kd), dtype, g), _rand((N, Kd), dtype, g)
    bias = torch.randn(N, generator=g, device=DEV)
    res = torch.randn((M, N), generator=g, device=DEV)
    base = A.float() @ W.float().t() + bias
    # bias + GELU with saved pre-activation
    out = torch.empty((M, N), dtype=dtype, device=DEV)
    pre = torch.empty_like(out)
    K.gemm(A, W, out, M, N, Kd, Kd, Kd, N, bias=bias, act=abi.ACT_GELU, aux_out=pre, ld_aux_out=N)
    assert rel_l2(pre.float(), base) < _tol(dtype)
    assert rel_l2(out.float(), torch.nn.functional.gelu(base)) < _tol(dtype)
    # bias + residual, fp32 out
    out32 = torch.empty((M, N), dtype=torch.float32, device=DEV)
    K.gemm(A, W, out32, M, N, Kd, Kd, Kd, N, bias=bias, residual=res, ld_res=N)
    assert rel_l2(out32, base + res) < 2e-5
    # dGELU epilogue:  (A W^T) * gelu'(pre)
    prex = _rand((M, N), dtype, g)
    K.gemm(A, W, out32, M, N, Kd, Kd, Kd, N, act=abi.ACT_DGELU, aux_in=prex, ld_aux_in=N)
    x = prex.float().requires_grad_(True)
    torch.nn.functional.gelu(x).sum().backward()
    assert rel_l2(out32, (A.float() @ W.float().t()) * x.grad) < 2e-5
    # accumulate (fp32 atomics) and alpha
    acc = res.clone()
    K.gemm(A, W, acc, M, N, Kd, Kd, Kd, N, accumulate=True, alpha=0.5)
    assert rel_l2(acc, res + 0.5 * (A.float() @ W.float().t())) < 2e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_weight_grad_splitk_and_bias_grad(K, dtype):
    """dW = dY^T X with the token dim as K (split-K + atomics) and the fused column sum."""
    g = torch.Generator(device=DEV).manual_seed(9)
    T, N, Kd = 4133, 200, 136
    dY, X = _rand((T, N), dtype, g), _rand((T, Kd), dtype, g)
    dW = torch.empty((N, Kd), dtype=torch.float32, device=DEV)
    db = torch.zeros(N, dtype=torch.float32, device=DEV)
    K.gemm(dY, X, dW, N, Kd, T, N, Kd, Kd, a_kmajor=False, b_kmajor=False, a_rowsum=db)
    assert rel_l2(dW, dY.double().t() @ X.double()) < 2e-5
    assert rel_l2(db, dY.double().sum(0)) < 2e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_batched_strided(K, dtype):
    """Per-(batch, head) Q K^T out of an interleaved [B*L, 3D] qkv buffer."""
    g = torch.Generator(device=DEV).manual_seed(11)
    B, H, L, hd = 3, 4, 37, 16
    D = H * hd
    qkv = _rand((B * L, 3 * D), dtype, g)
    S = torch.empty((B * H, L, L), dtype=torch.float32, device=DEV)
    K.gemm(qkv, qkv, S, L, L, hd, 3 * D, 3 * D, L, alpha=0.25, batch=B * H, batch_inner=H, sA=(L * 3 * D, hd),
           sB=(L * 3 * D, hd), sC=(H * L * L, L * L), a_off=0, b_off=D)
    t = qkv.float().reshape(B, L, 3, H, hd).permute(2, 0, 3, 1, 4)
    ref = 0.25 * (t[0] @ t[1].transpose(-2, -1)).reshape(B * H, L, L)
    assert rel_l2(S, ref) < 2e-5


def test_dropout_draws_are_uniform_and_independent_across_seeds(K):
    """The counter-based dropout draws (common.h favit_rand_u32: affine index map keyed by the seed + 32-bit
    finaliser): keep rate, no correlation between neighbouring elements, between strided elements, between the masks of
    different seeds (consecutive, random, and seeds that differ only in the high word), and no shifted copies."""
    n = 1 << 22
    ones = torch.ones(n, device=DEV)
    def mask(seed, p=0.25):
        return (K.dropout(ones, p, seed) != 0).float()
    seeds = [1, 2, 3, 0x123456789ABCDEF, 0x123456789ABCDEF + (1 << 32), 0x7FFFFFFF00000000, 0x42]
    ms = [mask(s) for s in seeds]
    tol = 5.0 / (n ** 0.5)                                   # five sigma of a Bernoulli mean / correlation estimate
    for m in ms:
        assert abs(float(m.mean()) - 0.75) < tol
        c = m - 0.75
        for lag in (1, 2, 3, 64, 384, 1536, 4096):
            assert abs(float((c[:-lag] * c[lag:]).mean()) / 0.1875) < tol, lag
    for i in range(len(ms)):
        for j in range(i + 1, len(ms)):
            ci, cj = ms[i] - 0.75, ms[j] - 0.75
            assert abs(float((ci * cj).mean()) / 0.1875) < tol, (seeds[i], seeds[j])
            for sh in (1, 7, 4097):                          # not a shifted copy either way
                assert abs(float((ci[sh:] * cj[:-sh]).mean()) / 0.1875) < tol
                assert abs(float((ci[:-sh] * cj[sh:]).mean()) / 0.1875) < tol
    for p in (0.1, 0.5, 0.9):
        assert abs(float(mask(99, p).mean()) - (1 - p)) < tol


def test_dropout_epoch_word_changes_the_masks(K, favit):
    """favit_set_dropout_epoch: the device word is mixed into every dropout seed at execution time (GEMM epilogue,
    stand-alone dropout, MHLA and dense attention).  Epoch 0 == no word; different epochs give different masks with the
    same keep rate; the GEMM epilogue and the dropout kernel stay consistent with each other under one epoch."""
    F = favit.functional
    g = torch.Generator(device=DEV).manual_seed(8)
    x = torch.ones(64 * 1024, device=DEV)
    base = K.dropout(x, 0.3, 1234)
    ep = torch.zeros(1, dtype=torch.int64, device=DEV)
    try:
        F.set_dropout_epoch(ep)
        assert torch.equal(K.dropout(x, 0.3, 1234), base), "epoch 0 must reproduce the by-value seed"
        ep.fill_(1)
        m1 = K.dropout(x, 0.3, 1234)
        ep.fill_(2)
        m2 = K.dropout(x, 0.3, 1234)
        assert not torch.equal(m1, base) and not torch.equal(m1, m2)
        for m in (m1, m2):
            assert abs((m == 0).float().mean().item() - 0.3) < 0.01
        agree = ((m1 == 0) == (m2 == 0)).float().mean().item()
        assert abs(agree - (0.3 * 0.3 + 0.7 * 0.7)) < 0.01, "masks of different epochs are independent"
        # GEMM epilogue under the same epoch == dropout kernel under the same epoch (index = m * N + n)
        M, N, Kd = 300, 256, 64
        a = _rand((M, Kd), torch.float32, g)
        b = _rand((N, Kd), torch.float32, g)
        out = torch.empty((M, N), device=DEV)
        K.gemm(a, b, out, M, N, Kd, Kd, Kd, N, dropout_p=0.25, dropout_seed=77)
        plain = torch.empty((M, N), device=DEV)
        K.gemm(a, b, plain, M, N, Kd, Kd, Kd, N)
        assert rel_l2(out, K.dropout(plain, 0.25, 77)) < 1e-6
        # attention: forward / backward mask consistency holds under a non-zero epoch (<dout, out(V)> == <dV, V>)
        B, H, L, hd, W = 2, 2, 40, 64, 7
        D = H * hd
        qkv = _rand((B * L, 3 * D), torch.bfloat16, g)
        dout = _rand((B * L, D), torch.bfloat16, g)
        o = K.mhla_attn_fwd(qkv, B, L, H, hd, W, None, 0.25, 9)
        dqkv = K.mhla_attn_bwd(qkv, dout, B, L, H, hd, W, None, 0.25, 9)
        lhs = (dout.float() * o.float()).sum().item()
        rhs = (dqkv[:, 2 * D:].float() * qkv[:, 2 * D:].float()).sum().item()
        assert abs(lhs - rhs) < 3e-2 * (dout.float() * o.float()).pow(2).sum().sqrt().item()
        ep.fill_(0)
        o0 = K.mhla_attn_fwd(qkv, B, L, H, hd, W, None, 0.25, 9)
        assert not torch.equal(o0, o)
    finally:
        F.set_dropout_epoch(None)
    assert torch.equal(K.dropout(x, 0.3, 1234), base)


def test_gemm_dropout_epilogue_matches_dropout_kernel(K):
    g = torch.Generator(device=DEV).manual_seed(13)
    M, N, Kd = 256, 128, 64
    A, W = _rand((M, Kd), torch.float32, g), _rand((N, Kd), torch.float32, g)
    y0 = torch.empty((M, N), device=DEV)
    y1 = torch.empty((M, N), device=DEV)
    K.gemm(A, W, y0, M, N, Kd, Kd, Kd, N)
    K.gemm(A, W, y1, M, N, Kd, Kd, Kd, N, dropout_p=0.3, dropout_seed=1234)
    ref = K.dropout(y0, 0.3, 1234)
    assert torch.equal(y1 == 0, ref == 0)
    assert rel_l2(y1, ref) < 1e-6
    keep = (y1 != 0).float().mean().item()
    assert abs(keep - 0.7) < 0.02


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,D", [(37, 64), (500, 384), (9, 192), (33, 768), (5, 1536)])
def test_layernorm_fwd_bwd(K, dtype, rows, D):
    g = torch.Generator(device=DEV).manual_seed(rows + D)
    x = torch.randn((rows, D), generator=g, device=DEV) * 2 + 0.5
    gam = torch.randn(D, generator=g, device=DEV)
    bet = torch.randn(D, generator=g, device=DEV)
    y, mu, rs = K.layernorm_fwd(x, D, gam, bet, rows, D, dtype)
    xr = x.clone().requires_grad_(True)
    gr, br = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xr, (D,), gr, br, 1e-5)
    assert rel_l2(y.float(), yr) < _tol(dtype)
    dy = _rand((rows, D), dtype, g)
    dres = torch.randn((rows, D), generator=g, device=DEV)
    dx, dx_lp, dg, db = K.layernorm_bwd(dy, x, D, gam, mu, rs, rows, D, dres=dres, want_lp=True)
    yr.backward(dy.float())
    assert rel_l2(dx, xr.grad + dres) < 2e-5
    assert rel_l2(dx_lp.float(), xr.grad + dres) < _tol(dtype)
    assert rel_l2(dg, gr.grad) < 2e-5 and rel_l2(db, br.grad) < 2e-5


def test_layernorm_strided_rows_for_cls_head(K):
    B, L, D = 6, 17, 64
    x = torch.randn(B, L, D, device=DEV)
    gam, bet = torch.randn(D, device=DEV), torch.randn(D, device=DEV)
    y, mu, rs = K.layernorm_fwd(x, L * D, gam, bet, B, D, torch.float32)
    assert rel_l2(y, torch.nn.functional.layer_norm(x[:, 0], (D,), gam, bet)) < 2e-5


def test_cast_roundtrip(K):
    x = torch.randn(100003, device=DEV)
    b = K.cast(x, torch.bfloat16)
    assert torch.equal(b, x.to(torch.bfloat16))
    assert torch.equal(K.cast(b, torch.float32), b.float())


@pytest.mark.parametrize("B,C,HW,P", [(2, 3, 32, 4), (3, 3, 224, 16), (2, 1, 30, 5), (2, 3, 36, 6), (1, 3, 384, 16)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_patchify_matches_einops_order(K, B, C, HW, P, dtype):
    """Both patchify kernels (LDS strip with 16-byte accesses; element-wise fallback for odd sizes) against the
    reference's einops order 'b c (h p1) (w p2) -> b (h w) (p1 p2 c)' -- a pure permutation: bit-exact."""
    from oracle import favit_oracle as O
    g = torch.Generator(device=DEV).manual_seed(HW + P)
    x = torch.randn(B, C, HW, HW, device=DEV, generator=g)
    Kp = P * P * C
    p = K.patchify_fwd(x, P, dtype)
    ref = O.patch_rearrange(x.cpu(), P).reshape(-1, Kp)
    assert torch.equal(p.cpu(), ref.to(dtype))
    if dtype == torch.float32:
        d = torch.randn_like(p)
        back = K.patchify_bwd(d, B, C, HW, P)
        xr = x.cpu().clone().requires_grad_(True)
        (O.patch_rearrange(xr, P).reshape(-1, Kp) * d.cpu()).sum().backward()
        assert torch.equal(back.cpu(), xr.grad)


@pytest.mark.parametrize("B,N,D", [(3, 16, 64), (64, 16, 64), (256, 196, 384), (40, 7, 20)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_embed_prologue(K, B, N, D, dtype):
    g = torch.Generator(device=DEV).manual_seed(B + N)
    tok = torch.randn(B, N, D, device=DEV, generator=g)
    cls, pos = torch.randn(D, device=DEV, generator=g), torch.randn(N + 1, D, device=DEV, generator=g)
    x = K.embed_prologue_fwd(tok, cls, pos, B, N, D)
    ref = torch.cat([cls.expand(B, 1, D), tok], 1) + pos
    assert torch.equal(x, ref)
    dx = torch.randn(B, N + 1, D, device=DEV, generator=g)
    dtok, dcls, dpos = K.embed_prologue_bwd(dx, B, N, D, dtype)
    assert torch.equal(dtok.reshape(B, N, D), dx[:, 1:].to(dtype))
    assert rel_l2(dcls, dx[:, 0].double().sum(0)) < 1e-6 and rel_l2(dpos, dx.double().sum(0)) < 1e-6


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("L,W,hd,masked", [(5, 3, 16, False), (5, 7, 16, False), (7, 7, 16, True), (12, 7, 64, False),
                                           (17, 5, 32, True), (65, 7, 64, False), (197, 7, 64, False),
                                           (197, 15, 16, False), (33, 9, 128, False), (64, 7, 64, False),
                                           (100, 3, 64, True)])
@pytest.mark.parametrize("bwd_kernel", ["default", "tables", "valu", "lse"])
def test_mhla_core_fwd_bwd_vs_window_gather(K, dtype, L, W, hd, masked, bwd_kernel, monkeypatch):
    """The attention core against a direct restatement of the reference's gather-based windows
    (duplicated pad indices take part in the softmax, models/mhla.py:117-154).  bf16, hd >= 32 has three
    backward kernels without saved statistics: the default two-owner-pass MFMA kernel, the table formulation and the
    8-lanes-per-row one; "lse" is the training path's pair (forward leaves lse, backward takes it and the output)."""
    from oracle import favit_oracle as O
    if bwd_kernel == "lse":
        if not K.mhla_attn_lse_supported(L, hd, W, dtype):
            pytest.skip("saved-statistics kernels: bf16, hd = 64, W <= 7 (11 with L > 16), L >= W + 1")
    elif bwd_kernel != "default":
        if dtype != torch.bfloat16 or hd < 32:
            pytest.skip("kernel selection only exists for bf16, hd >= 32")
        monkeypatch.setenv("FAVIT_MHLA_BWD_TABLES" if bwd_kernel == "tables" else "FAVIT_MHLA_VALU", "1")   # read per call
    B, H = 2, 3
    D = H * hd
    g = torch.Generator(device=DEV).manual_seed(L * 31 + W)
    qkv = _rand((B * L, 3 * D), dtype, g)
    dout = _rand((B * L, D), dtype, g)
    mask = None
    if masked:
        mask = (torch.rand(B, L, L, generator=g, device=DEV) > 0.4)
        mask |= torch.eye(L, dtype=torch.bool, device=DEV)
        mask = mask.to(torch.uint8).contiguous()
    lse = None
    if bwd_kernel == "lse":
        out, lse = K.mhla_attn_fwd(qkv, B, L, H, hd, W, mask, want_lse=True)
        dqkv = K.mhla_attn_bwd(qkv, dout, B, L, H, hd, W, mask, o=out, lse=lse)
    else:
        out = K.mhla_attn_fwd(qkv, B, L, H, hd, W, mask)
        dqkv = K.mhla_attn_bwd(qkv, dout, B, L, H, hd, W, mask)
    idx = torch.from_numpy(O.window_indices(L, W)).to(DEV)
    t = qkv.float().reshape(B, L, 3, H, hd).permute(2, 0, 3, 1, 4).detach().clone().requires_grad_(True)
    q, k, v = t[0], t[1], t[2]
    kw, vw = k[:, :, idx], v[:, :, idx]                       # [B,H,L,W,hd]
    s = (q.unsqueeze(3) @ kw.transpose(-2, -1)).squeeze(3) / (hd ** 0.5)
    if mask is not None:
        wm = torch.gather(mask[:, None].expand(B, H, L, L), 3, idx[None, None].expand(B, H, L, W))
        s = s.masked_fill(wm == 0, float("-inf"))
    o = (torch.softmax(s, -1).unsqueeze(3) @ vw).squeeze(3).transpose(1, 2).reshape(B * L, D)
    assert rel_l2(out.float(), o) < _tol(dtype)
    if lse is not None:
        assert (lse - torch.logsumexp(s.detach(), -1)).abs().max().item() < 2e-3
    o.backward(dout.float())
    gref = t.grad.permute(1, 3, 0, 2, 4).reshape(B * L, 3 * D)
    assert rel_l2(dqkv.float(), gref) < (5e-5 if dtype == torch.float32 else 1.5e-2)


@pytest.mark.parametrize("dtype,hd,L,tol,lse", [(torch.float32, 16, 40, 1e-3, False), (torch.bfloat16, 64, 70, 3e-2, False),
                                                 (torch.bfloat16, 64, 5, 3e-2, False), (torch.bfloat16, 16, 40, 3e-2, False),
                                                 (torch.bfloat16, 64, 70, 3e-2, True), (torch.bfloat16, 64, 197, 3e-2, True),
                                                 (torch.bfloat16, 64, 17, 3e-2, True)])
def test_mhla_core_dropout_consistency(K, dtype, hd, L, tol, lse):
    """Train-mode attention dropout: fwd (MFMA kernel for bf16 hd>=32) and bwd use the same
    per-window-slot mask (out is linear in V, so <dout, out(V)> == <dV, V>).  lse: the saved-statistics pair, whose
    backward must also agree with the statistics-free kernel on all of dqkv (same masks, same seeds)."""
    B, H, W = 2, 2, 7
    D = H * hd
    g = torch.Generator(device=DEV).manual_seed(1000 + L * 7 + hd)
    qkv = _rand((B * L, 3 * D), dtype, g)
    dout = _rand((B * L, D), dtype, g)
    if lse:
        out, st = K.mhla_attn_fwd(qkv, B, L, H, hd, W, None, 0.25, 77, want_lse=True)
        assert st is not None
        dqkv = K.mhla_attn_bwd(qkv, dout, B, L, H, hd, W, None, 0.25, 77, o=out, lse=st)
        assert rel_l2(dqkv.float(), K.mhla_attn_bwd(qkv, dout, B, L, H, hd, W, None, 0.25, 77).float()) < 1e-2
    else:
        out = K.mhla_attn_fwd(qkv, B, L, H, hd, W, None, 0.25, 77)
        dqkv = K.mhla_attn_bwd(qkv, dout, B, L, H, hd, W, None, 0.25, 77)
    lhs = (dout.float() * out.float()).sum().item()
    rhs = (dqkv[:, 2 * D:].float() * qkv[:, 2 * D:].float()).sum().item()
    # both sides are sums of random-sign terms: rounding noise AND a mask mismatch both scale with
    # the root-sum-square of the terms (a mismatch gives O(1) of it, rounding O(2^-8))
    scale = (dout.float() * out.float()).pow(2).sum().sqrt().item()
    assert abs(lhs - rhs) < tol * max(1e-6, scale)
    keep = (out.float().abs().sum(-1) > 0).float().mean().item()
    assert keep > 0.9            # rows are not dropped wholesale
    out2 = K.mhla_attn_fwd(qkv, B, L, H, hd, W, None, 0.25, 78)
    assert not torch.allclose(out.float(), out2.float())


def test_mhla_fold_matches_separate_latent_proj(K):
    H, hd = 4, 16
    D = H * hd
    g = torch.Generator(device=DEV).manual_seed(3)
    wqkv, bqkv = torch.randn(3 * D, D, generator=g, device=DEV), torch.randn(3 * D, generator=g, device=DEV)
    wl, bl = torch.randn(hd, hd, generator=g, device=DEV), torch.randn(hd, generator=g, device=DEV)
    weff, beff = K.mhla_fold_fwd(wqkv, bqkv, wl, bl, H, torch.float32)
    w, b, l, lb = [t.clone().requires_grad_(True) for t in (wqkv, bqkv, wl, bl)]
    wk = w.reshape(3, H, hd, D)
    weff_ref = torch.cat([wk[0].reshape(D, D), (l @ wk[1]).reshape(D, D), (l @ wk[2]).reshape(D, D)])
    bk = b.reshape(3, H, hd)
    beff_ref = torch.cat([bk[0].reshape(D), (bk[1] @ l.t() + lb).reshape(D), (bk[2] @ l.t() + lb).reshape(D)])
    assert rel_l2(weff, weff_ref) < 2e-5 and rel_l2(beff, beff_ref) < 2e-5
    dweff, dbeff = torch.randn(3 * D, D, generator=g, device=DEV), torch.randn(3 * D, generator=g, device=DEV)
    ((weff_ref * dweff).sum() + (beff_ref * dbeff).sum()).backward()
    dwqkv, dbqkv, dwl, dbl = K.mhla_fold_bwd(dweff, dbeff, wqkv, bqkv, wl, H)
    for got, ref in ((dwqkv, w.grad), (dbqkv, b.grad), (dwl, l.grad), (dbl, lb.grad)):
        assert rel_l2(got, ref) < 2e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_softmax_fwd_bwd(K, dtype):
    Z, Lq, Lk, H = 6, 9, 70, 3
    g = torch.Generator(device=DEV).manual_seed(21)
    S = torch.randn(Z, Lq, Lk, generator=g, device=DEV) * 3
    mask = (torch.rand(Z // H, Lq, Lk, generator=g, device=DEV) > 0.3)
    mask[..., 0] = True
    P, Pd = K.softmax_fwd(S, dtype, H, Z, Lq, Lk, mask.to(torch.uint8).contiguous(), Lq * Lk, Lk)
    Sr = S.clone().requires_grad_(True)
    ref = torch.softmax(Sr.masked_fill(~mask.repeat_interleave(H, 0), float("-inf")), -1)
    assert rel_l2(P.float(), ref) < _tol(dtype)
    dP = torch.randn(Z, Lq, Lk, generator=g, device=DEV)
    dS = K.softmax_bwd(P, dP, Z, Lq, Lk)
    ref.backward(dP)
    assert rel_l2(dS.float(), Sr.grad) < (2e-5 if dtype == torch.float32 else 2e-2)


def _sdpa_views(favit, B, H, L, hd, t, col0, ld):
    V = favit.functional._View
    return V(t, col0, ld, L * ld, hd)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,Lq,Lk,hd,mask_kind", [(2, 3, 65, 65, 64, None), (2, 4, 17, 17, 16, None), (3, 2, 5, 40, 32, "full"),
                                                  (2, 1, 33, 20, 64, "keys"), (1, 2, 197, 197, 64, None), (2, 1, 9, 9, 768, "full"),
                                                  (1, 3, 70, 100, 48, "keys"), (2, 2, 1, 1, 16, None), (1, 1, 130, 64, 192, None),
                                                  (2, 5, 64, 32, 128, "full")])
@pytest.mark.parametrize("waves", [0, 4, 5])
def test_sdpa_fused_fwd_bwd(K, favit, dtype, B, H, Lq, Lk, hd, mask_kind, waves, monkeypatch):
    """The fused attention kernels (forward, dQ, dK/dV) against softmax(q k^T * scale, masked) v in fp32 torch:
    separate and interleaved operand layouts, both mask forms, head dims that need padding (16, 48) and
    column chunking (192, 768), Lq != Lk, single-row problems; with the library's own choice of four or five waves
    (64 / 80 owner rows) per workgroup and with either forced."""
    if waves:
        monkeypatch.setenv("FAVIT_SDPA_WAVES", str(waves))
    g = torch.Generator(device=DEV).manual_seed(B * 1000 + Lq * 10 + hd)
    D = H * hd
    V = favit.functional._View
    self_attn = Lq == Lk
    if self_attn:        # q, k, v interleaved in one [B*L, 3D] buffer, as the fused qkv projection writes them
        qkv = _rand((B * Lq, 3 * D), dtype, g)
        q, k, v = (V(qkv, s * D, 3 * D, Lq * 3 * D, hd) for s in range(3))
        qf, kf, vf = (qkv.float().reshape(B, Lq, 3, H, hd)[:, :, s].permute(0, 2, 1, 3) for s in range(3))
    else:
        qt, kt, vt = _rand((B * Lq, D), dtype, g), _rand((B * Lk, D), dtype, g), _rand((B * Lk, D), dtype, g)
        q, k, v = V(qt, 0, D, Lq * D, hd), V(kt, 0, D, Lk * D, hd), V(vt, 0, D, Lk * D, hd)
        qf = qt.float().reshape(B, Lq, H, hd).permute(0, 2, 1, 3)
        kf, vf = (t.float().reshape(B, Lk, H, hd).permute(0, 2, 1, 3) for t in (kt, vt))
    mask, m_sb, m_sq, mb = None, 0, 0, None
    if mask_kind == "full":
        mb = torch.rand(B, Lq, Lk, generator=g, device=DEV) > 0.4
        mb[..., 0] = True
        mask, m_sb, m_sq = mb.to(torch.uint8).contiguous(), Lq * Lk, Lk
        mb = mb[:, None]
    elif mask_kind == "keys":
        mk = torch.rand(B, Lk, generator=g, device=DEV) > 0.3
        mk[:, 0] = True
        mask, m_sb, m_sq = mk.to(torch.uint8).contiguous(), Lk, 0
        mb = mk[:, None, None, :]
    scale = 1.0 / math.sqrt(hd)
    ot = torch.empty((B * Lq, D), dtype=dtype, device=DEV)
    o = V(ot, 0, D, Lq * D, hd)
    lse = K.sdpa_fwd(q, k, v, o, B, H, Lq, Lk, hd, scale, mask, m_sb, m_sq)
    qr, kr, vr = (t.detach().clone().requires_grad_(True) for t in (qf, kf, vf))
    sr = (qr @ kr.transpose(-2, -1)) * scale
    if mb is not None:
        sr = sr.masked_fill(~mb, float("-inf"))
    oref = torch.softmax(sr, -1) @ vr                                    # [B, H, Lq, hd]
    tol = 2e-5 if dtype == torch.float32 else 1e-2
    assert rel_l2(ot.float().reshape(B, Lq, H, hd).permute(0, 2, 1, 3), oref) < tol
    assert rel_l2(lse.reshape(B, H, Lq), torch.logsumexp(sr, -1)) < 2e-5 + (0 if dtype == torch.float32 else 3e-3)
    dot = _rand((B * Lq, D), dtype, g)
    dqt = torch.empty((B * Lq, D), dtype=dtype, device=DEV)
    dkt, dvt = (torch.empty((B * Lk, D), dtype=dtype, device=DEV) for _ in range(2))
    K.sdpa_bwd(q, k, v, o, V(dot, 0, D, Lq * D, hd), V(dqt, 0, D, Lq * D, hd), V(dkt, 0, D, Lk * D, hd),
               V(dvt, 0, D, Lk * D, hd), lse, B, H, Lq, Lk, hd, scale, mask, m_sb, m_sq)
    oref.backward(dot.float().reshape(B, Lq, H, hd).permute(0, 2, 1, 3))
    gtol = 5e-5 if dtype == torch.float32 else 2e-2
    for got, ref, L in ((dqt, qr.grad, Lq), (dkt, kr.grad, Lk), (dvt, vr.grad, Lk)):
        assert rel_l2(got.float().reshape(B, L, H, hd).permute(0, 2, 1, 3), ref) < gtol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_sdpa_fused_dropout_consistency(K, favit, dtype):
    """Train-mode attention dropout of the dense variants: forward and both backward kernels draw the same
    mask (the output is linear in V: <dO, O(V)> == <dV, V>), the keep rate is 1 - p, another seed differs."""
    B, H, L, hd, p = 2, 2, 70, 32, 0.25
    D = H * hd
    V = favit.functional._View
    g = torch.Generator(device=DEV).manual_seed(77)
    qkv = _rand((B * L, 3 * D), dtype, g)
    q, k, v = (V(qkv, s * D, 3 * D, L * 3 * D, hd) for s in range(3))
    ot = torch.empty((B * L, D), dtype=dtype, device=DEV)
    o = V(ot, 0, D, L * D, hd)
    lse = K.sdpa_fwd(q, k, v, o, B, H, L, L, hd, hd ** -0.5, p=p, seed=5)
    dot = _rand((B * L, D), dtype, g)
    dqkv = torch.empty_like(qkv)
    dq, dk, dv = (V(dqkv, s * D, 3 * D, L * 3 * D, hd) for s in range(3))
    K.sdpa_bwd(q, k, v, o, V(dot, 0, D, L * D, hd), dq, dk, dv, lse, B, H, L, L, hd, hd ** -0.5, p=p, seed=5)
    lhs = (dot.float() * ot.float()).sum().item()
    rhs = (dqkv[:, 2 * D:].float() * qkv[:, 2 * D:].float()).sum().item()
    scale = (dot.float() * ot.float()).pow(2).sum().sqrt().item()
    assert abs(lhs - rhs) < (1e-3 if dtype == torch.float32 else 3e-2) * scale
    ot2 = torch.empty_like(ot)
    K.sdpa_fwd(q, k, v, V(ot2, 0, D, L * D, hd), B, H, L, L, hd, hd ** -0.5, p=p, seed=6)
    assert not torch.allclose(ot.float(), ot2.float())
    # with V = ones the output row is sum_j Pd_ij = (kept mass) / (1 - p): its mean over rows is ~1
    ones = qkv.clone()
    ones[:, 2 * D:] = 1.0
    K.sdpa_fwd(*(V(ones, s * D, 3 * D, L * 3 * D, hd) for s in range(3)), V(ot2, 0, D, L * D, hd), B, H, L, L, hd,
               hd ** -0.5, p=p, seed=9)
    assert abs(ot2.float().mean().item() - 1.0) < 0.03


def test_cross_entropy_and_adamw(K):
    B, Cn = 37, 1000
    logits = torch.randn(B, Cn, device=DEV) * 3
    labels = torch.randint(0, Cn, (B,), device=DEV)
    rows, dlog = K.cross_entropy(logits, labels, grad_scale=1.0 / B)
    lr = logits.clone().requires_grad_(True)
    loss = torch.nn.functional.cross_entropy(lr, labels)
    loss.backward()
    assert abs(rows.mean().item() - loss.item()) < 1e-5 and rel_l2(dlog, lr.grad) < 2e-5
    p = torch.randn(10007, device=DEV)
    gr = torch.randn_like(p)
    pr = p.clone().requires_grad_(True)
    opt = torch.optim.AdamW([pr], lr=1e-3, weight_decay=0.05)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    lp = torch.empty_like(p, dtype=torch.bfloat16)
    for step in (1, 2, 3):
        pr.grad = gr.clone()
        opt.step()
        K.adamw(p, gr, m, v, 1e-3, 0.9, 0.999, 1e-8, 0.05, step, p_lp=lp)
    assert rel_l2(p, pr.detach()) < 1e-6
    assert torch.equal(lp, p.to(torch.bfloat16))


@pytest.mark.parametrize("M,D,N,mode", [(128 * 17, 384, 1152, "plain"), (128 * 17, 384, 1536, "gelu"), (64 * 65, 192, 576, "plain"),
                                       (64 * 65, 192, 768, "gelu"), (1000 - 3, 256, 200, "plain"), (333, 64, 1000, "gelu"),
                                       (700, 512, 384, "plain"), (2176, 128, 128, "gelu")])
def test_layernorm_fused_into_small_gemm(K, favit, M, D, N, mode):
    """favit_ln_gemm (LayerNorm in the A-operand staging of the 64-row GEMM) == favit_layernorm_fwd followed by
    favit_gemm: the saved xn / mean / rstd and the product, against the two-launch path and an fp32 restatement."""
    g = torch.Generator(device=DEV).manual_seed(M + D + N)
    x = torch.randn(M, D, device=DEV, generator=g) * 1.7 + 0.3
    gamma = torch.randn(D, device=DEV, generator=g) * 0.2 + 1.0
    beta = torch.randn(D, device=DEV, generator=g) * 0.1
    w = (torch.randn(N, D, device=DEV, generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=DEV, generator=g) * 0.1
    act = favit._abi.ACT_GELU_SAVEGRAD if mode == "gelu" else favit._abi.ACT_NONE
    out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
    pre = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV) if mode == "gelu" else None
    r = K.ln_gemm(x, D, gamma, beta, w, out, M, N, D, bias=bias, act=act, aux_out=pre)
    assert r is not None, "the library must take this shape"
    xn, mean, rstd = r
    xn2, mean2, rstd2 = K.layernorm_fwd(x, D, gamma, beta, M, D, torch.bfloat16)
    out2 = torch.empty_like(out)
    pre2 = torch.empty_like(out) if mode == "gelu" else None
    K.gemm(xn2, w, out2, M, N, D, D, D, N, bias=bias, act=act, aux_out=pre2, ld_aux_out=N)
    torch.cuda.synchronize()
    assert rel_l2(mean, mean2) < 1e-6 and rel_l2(rstd, rstd2) < 1e-6
    # the same arithmetic up to the summation order of the two reductions: bf16 values differ by at most one rounding
    assert float((xn.float() - xn2.float()).abs().max()) <= 2.0 ** -7 * float(xn2.float().abs().max())
    assert rel_l2(xn, xn2) < 1e-3
    assert rel_l2(out, out2) < 4e-3
    ref_xn = torch.nn.functional.layer_norm(x, (D,), gamma, beta)
    ref = ref_xn.to(torch.bfloat16).float() @ w.float().t() + bias
    if mode == "gelu":
        assert rel_l2(pre, pre2) < 4e-3
        u = ref.double()
        dref = 0.5 * (1 + torch.erf(u / math.sqrt(2))) + u * torch.exp(-0.5 * u * u) / math.sqrt(2 * math.pi)
        assert rel_l2(pre, dref) < 1e-2
        ref = torch.nn.functional.gelu(ref)
    assert rel_l2(out, ref) < 1e-2
    assert torch.isfinite(out.float()).all()


def test_layernorm_fused_gemm_declines_large_or_odd_shapes(K):
    x = torch.randn(50432, 384, device=DEV)
    gm, bt = torch.ones(384, device=DEV), torch.zeros(384, device=DEV)
    w = torch.randn(1152, 384, device=DEV).to(torch.bfloat16)
    out = torch.empty(50432, 1152, dtype=torch.bfloat16, device=DEV)
    assert K.ln_gemm(x, 384, gm, bt, w, out, 50432, 1152, 384) is None            # cfg2: the large-tile kernels' regime
    x = torch.randn(256, 320, device=DEV)
    w = torch.randn(128, 320, device=DEV).to(torch.bfloat16)
    out = torch.empty(256, 128, dtype=torch.bfloat16, device=DEV)
    assert K.ln_gemm(x, 320, torch.ones(320, device=DEV), torch.zeros(320, device=DEV), w, out, 256, 128, 320) is None


@pytest.mark.parametrize("bk", [True, False])
@pytest.mark.parametrize("M,N,Kd", [(128 * 17, 384, 384), (128 * 17, 384, 1152), (128 * 17, 384, 1536), (64 * 65, 192, 768),
                                    (64 * 65, 192, 576), (128 * 17 - 13, 384, 320), (1000, 200, 256), (4160, 136, 1984)])
@pytest.mark.parametrize("out_dtype,epi", [(torch.bfloat16, "plain"), (torch.bfloat16, "dgelu"), (torch.float32, "res")])
def test_gemm_small_m_split_k_inside_the_workgroup(K, favit, bk, M, N, Kd, out_dtype, epi):
    """s64k2: launches of the 64-row kernel with no more tiles than CUs run their K loop in two wave groups (even / odd
    k-steps) and add the halves through LDS.  Against fp64 on bf16-rounded operands, even and odd k-step counts, ragged
    M and N, both B layouts, the epilogues the short-token steps use; bitwise reproducible."""
    g = torch.Generator(device=DEV).manual_seed(M + N + Kd)
    A = _rand((M, Kd), torch.bfloat16, g)
    B = _rand((N, Kd) if bk else (Kd, N), torch.bfloat16, g)
    bias = torch.randn(N, device=DEV, generator=g)
    res = torch.randn(M, N, device=DEV, generator=g) if epi == "res" else None
    aux = _rand((M, N), torch.bfloat16, g) if epi == "dgelu" else None
    outs = []
    for _ in range(2):
        C = torch.full((M, N), float("nan"), dtype=out_dtype, device=DEV)
        K.gemm(A, B, C, M, N, Kd, Kd, B.stride(0), N, b_kmajor=bk, bias=bias, residual=res, ld_res=N,
               act=favit._abi.ACT_DGELU if epi == "dgelu" else favit._abi.ACT_NONE, aux_in=aux, ld_aux_in=N)
        assert favit._abi.lib().favit_gemm_last_kernel().decode() == "s64k2"
        outs.append(C)
    ref = A.double() @ (B.double().t() if bk else B.double()) + bias.double()
    if epi == "dgelu":
        u = aux.double()
        ref = ref * (0.5 * (1 + torch.erf(u / math.sqrt(2))) + u * torch.exp(-0.5 * u * u) / math.sqrt(2 * math.pi))
    if res is not None:
        ref = ref + res.double()
    assert torch.isfinite(outs[0].float()).all()
    assert rel_l2(outs[0], ref) < (2e-5 if out_dtype == torch.float32 else 1e-2)
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("wgs", ["8", "64", "256"])
@pytest.mark.parametrize("M,N,Kd", [(2048, 384, 384), (4096, 1152, 384), (8192, 384, 1536), (50432, 384, 384), (50432, 384, 1536)])
@pytest.mark.parametrize("epi", ["bf16_bias", "bf16_plain", "f32_res"])
def test_gemm_persistent_deferred_epilogue_kernel(K, favit, monkeypatch, wgs, M, N, Kd, epi):
    """The experimental persistent kernel (FAVIT_GEMM_PD=1: one 8-wave workgroup per CU walks its tiles, the epilogue of
    tile t runs in slices between the k-steps of tile t + 1, every memory operation counted for the ring's waits):
    BITWISE equal to the 256x128-tile kernel (same accumulation chain per element, same epilogue arithmetic), for 1 to
    many tiles per workgroup (FAVIT_GEMM_PD_WGS), with and without bias / residual."""
    g = torch.Generator(device=DEV).manual_seed(M + N + Kd)
    A = _rand((M, Kd), torch.bfloat16, g)
    B = _rand((N, Kd), torch.bfloat16, g)
    bias = torch.randn(N, device=DEV, generator=g) if epi != "bf16_plain" else None
    res = torch.randn(M, N, device=DEV, generator=g) if epi == "f32_res" else None
    odt = torch.float32 if epi == "f32_res" else torch.bfloat16
    ref = torch.full((M, N), float("nan"), dtype=odt, device=DEV)
    K.gemm(A, B, ref, M, N, Kd, Kd, Kd, N, bias=bias, residual=res, ld_res=N)
    base = favit._abi.lib().favit_gemm_last_kernel().decode()
    monkeypatch.setenv("FAVIT_GEMM_PD", "1")
    monkeypatch.setenv("FAVIT_GEMM_PD_WGS", wgs)
    out = torch.full((M, N), float("nan"), dtype=odt, device=DEV)
    K.gemm(A, B, out, M, N, Kd, Kd, Kd, N, bias=bias, residual=res, ld_res=N)
    assert favit._abi.lib().favit_gemm_last_kernel().decode() == "pd"
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all()
    r64 = A.double() @ B.double().t()
    if bias is not None:
        r64 = r64 + bias.double()
    if res is not None:
        r64 = r64 + res.double()
    assert rel_l2(out, r64) < (2e-5 if odt == torch.float32 else 1e-2)
    if base in ("p4", "s64", "s64k2") and base != "s64k2":
        assert torch.equal(out, ref), f"differs from the {base} kernel"


def test_zero_fills_survive_graph_replays(K):
    """The launches that zero a destination and then add into it with atomics (embed-prologue backward: cls_token /
    pos_embed gradients; latent_proj fold backward), captured ONCE in a HIP graph and replayed: every replay equals the
    eager result.  With hipMemsetAsync as the zero-fill the second and later replays came back with dword 2 of every 16
    bytes at an arbitrary huge constant (ROCm 7.2 graph memset nodes; tools/graph_memset_probe.py) -- the cause of the
    frozen pos_embed and of the rare NaN losses of graph-replayed steps."""
    g = torch.Generator(device=DEV).manual_seed(5)
    B, N, D, H = 64, 64, 192, 3
    dx = torch.randn(B, N + 1, D, device=DEV, generator=g)
    ref_tok, ref_cls, ref_pos = K.embed_prologue_bwd(dx, B, N, D, torch.bfloat16)
    assert rel_l2(ref_pos, dx.sum(0)) < 1e-5 and rel_l2(ref_cls, dx[:, 0].sum(0)) < 1e-5
    hd = D // H
    dweff = torch.randn(3 * D, D, device=DEV, generator=g)
    dbeff = torch.randn(3 * D, device=DEV, generator=g)
    wqkv = torch.randn(3 * D, D, device=DEV, generator=g) * 0.05
    bqkv = torch.randn(3 * D, device=DEV, generator=g) * 0.05
    wl = torch.randn(hd, hd, device=DEV, generator=g) * 0.1
    ref_fold = K.mhla_fold_bwd(dweff, dbeff, wqkv, bqkv, wl, H)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        K.embed_prologue_bwd(dx, B, N, D, torch.bfloat16)
        K.mhla_fold_bwd(dweff, dbeff, wqkv, bqkv, wl, H)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        tok, cls, pos = K.embed_prologue_bwd(dx, B, N, D, torch.bfloat16)
        fold = K.mhla_fold_bwd(dweff, dbeff, wqkv, bqkv, wl, H)
    for rep in range(4):
        for t in (cls, pos, fold[2], fold[3]):
            t.fill_(7.0)                                  # whatever the previous replay (or anybody) left there
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(tok, ref_tok), rep
        # (atomics: order-dependent rounding only)
        assert rel_l2(pos, ref_pos) < 1e-6 and rel_l2(cls, ref_cls) < 1e-6, rep
        assert float((pos - ref_pos).abs().max()) < 1e-3 and float((cls - ref_cls).abs().max()) < 1e-3, rep
        for a, b in zip(fold, ref_fold):
            assert rel_l2(a, b) < 1e-5 and float((a - b).abs().max()) < 1e-2 * float(b.abs().max()), rep


def test_health_word_notes_first_non_finite_loss_gradient_and_parameter(favit, K):
    """include/favit.h favit_set_health_word: clean calls leave the four words at (0, 0, 0, launches); a NaN gradient,
    an inf logit row and an overflowing parameter are each noted with the AdamW launch index they first appeared at."""
    h = favit.train.Health(torch.device(DEV))
    try:
        p = torch.randn(5000, device=DEV)
        g = torch.randn_like(p)
        m, v = torch.zeros_like(p), torch.zeros_like(p)
        logits = torch.randn(9, 10, device=DEV)
        labels = torch.randint(0, 10, (9,), device=DEV)
        for step in (1, 2):
            K.cross_entropy(logits, labels, grad_scale=1.0)
            K.adamw(p, g, m, v, 1e-3, 0.9, 0.999, 1e-8, 0.05, step)
        assert h.poll() is None and h.words.tolist() == [0, 0, 0, 2]
        # out-of-range label: a NaN loss row by contract, NOT a health event
        K.cross_entropy(logits, torch.full_like(labels, -100), grad_scale=1.0)
        assert h.poll() is None
        g2 = g.clone()
        g2[4321] = float("nan")
        K.adamw(p, g2, m, v, 1e-3, 0.9, 0.999, 1e-8, 0.05, 3)           # launch 3: gradient and parameter go bad
        r = h.poll()
        # (block 0 counts the launch when IT is done: blocks that finish later read the counter one higher)
        assert r["non_finite"] == ["gradient", "parameter"] and r["adamw_launch_of_first_bad_gradient_or_parameter"] in (3, 4)
        assert r["adamw_launches"] == 3 and r["adamw_launch_of_first_bad_loss"] is None
        bad_logits = logits.clone()
        bad_logits[5, 3] = float("inf")
        K.cross_entropy(bad_logits, labels, grad_scale=1.0)
        r = h.poll()
        assert r["non_finite"] == ["loss", "gradient", "parameter"] and r["adamw_launch_of_first_bad_loss"] == 4
        K.adamw(p, g, m, v, 1e-3, 0.9, 0.999, 1e-8, 0.05, 4)            # later launches do not move the first-event marks
        r = h.poll()
        assert r["adamw_launch_of_first_bad_gradient_or_parameter"] in (3, 4) and r["adamw_launches"] == 4
    finally:
        h.close()
    # switched off: the kernels run without the word
    K.adamw(p, g, m, v, 1e-3, 0.9, 0.999, 1e-8, 0.05, 5)
    assert h.words.tolist()[3] == 4


def test_sppp_kernels_vs_golden(K):
    from conftest import case, load_golden
    SP = load_golden("sppp.npz")
    emb = torch.from_numpy(SP["emb"]).to(DEV)
    for nm in ("grid", "vor16", "vor15"):
        c = case(SP, nm)
        seg = torch.from_numpy(c["segmap"].astype(np.int64)).to(DEV)
        rank, ntok, perm, offs, dom = K.sppp_map_patches(seg[None], 16)
        np.testing.assert_array_equal(rank[0].cpu().numpy(), c["patch_rank"])     # bit-exact integer work
        R = int(ntok[0].item())
        assert R == len(c["map_keys"])
        first = [int(dom[0][(rank[0] == r).nonzero()[0, 0]].item()) for r in range(R)]
        assert first == c["map_keys"].tolist()
        for kind, kname in enumerate(("mean", "max", "attention")):
            out, argmax = K.sppp_pool_fwd(emb[None].contiguous(), perm, offs, kind, R)
            assert rel_l2(out[0], c[f"pool_{kname}"]) < 2e-5, (nm, kname)
            gout = torch.from_numpy(c[f"pool_{kname}_gout"]).to(DEV)[None].contiguous()
            gin = K.sppp_pool_bwd(gout, emb[None].contiguous(), perm, offs, argmax, kind, R)
            assert rel_l2(gin[0], c[f"pool_{kname}_gin"]) < 2e-5, (nm, kname)
        segs = torch.stack([seg, torch.roll(seg, 5, dims=1)])
        cent = K.sppp_centroids(segs, 16)
        assert rel_l2(cent, c["centroids"]) < 2e-5
        pe = K.sppp_posenc_fwd(torch.from_numpy(c["posenc_in"]).to(DEV), cent)
        assert rel_l2(pe, c["posenc_out"]) < 2e-5
    pe = K.sppp_posenc_fwd(torch.from_numpy(SP["posenc_nocentroid_in"]).to(DEV), None)
    assert rel_l2(pe, SP["posenc_nocentroid_out"]) < 2e-5


# ---------------------------------------------------------------------------------------------
# fp8 GEMM path (BASELINE.json configs[3]); no reference counterpart -- checked against torch's own
# OCP float8 conversions and an fp32 matmul of the dequantised operands.
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("src_dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("fmt", [torch.float8_e4m3fn, torch.float8_e5m2])
@pytest.mark.parametrize("rows,cols", [(64, 64), (197, 384), (577, 96), (50, 130), (1, 16)])
def test_fp8_quantize_bit_exact_vs_torch(K, src_dtype, fmt, rows, cols):
    g = torch.Generator(device=DEV).manual_seed(rows * 1000 + cols)
    x = (_rand((rows, cols), torch.float32, g) * 3.0).to(src_dtype)
    colsum = torch.zeros(cols, device=DEV)
    q, qt, sinv = K.fp8_quantize(x, fmt, want=True, want_t=True, colsum=colsum)
    fmax = 448.0 if fmt == torch.float8_e4m3fn else 57344.0
    amax = np.float32(x.float().abs().max().item())
    # IEEE fp32 divisions (torch divides a tensor by a scalar through a reciprocal multiply)
    assert sinv.item() == float(amax / np.float32(fmax))
    scale = torch.tensor(float(np.float32(fmax) / amax), dtype=torch.float32, device=DEV)
    ref = (x.float() * scale).clamp(-fmax, fmax).to(fmt)
    assert torch.equal(q.view(torch.uint8), ref.view(torch.uint8))
    ld_t = (rows + 63) // 64 * 64
    assert qt.shape == (cols, ld_t)
    assert torch.equal(qt[:, :rows].view(torch.uint8), ref.t().contiguous().view(torch.uint8))
    assert int(qt[:, rows:].view(torch.uint8).to(torch.int32).abs().sum()) == 0          # zero pad
    assert rel_l2(colsum, x.float().sum(0)) < 1e-5


@pytest.mark.parametrize("afmt", [torch.float8_e4m3fn, torch.float8_e5m2])
@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,Kd", [(256, 128, 64), (300, 200, 128), (1154, 1152, 384), (33, 1000, 768), (5, 7, 64)])
def test_fp8_gemm_vs_dequantised_fp32_matmul(K, afmt, out_dtype, M, N, Kd):
    g = torch.Generator(device=DEV).manual_seed(M + N + Kd)
    a = _rand((M, Kd), torch.bfloat16, g)
    b = _rand((N, Kd), torch.bfloat16, g)
    bias = _rand((N,), torch.float32, g)
    res = _rand((M, N), torch.float32, g) if out_dtype == torch.float32 else None
    aq, _, sa = K.fp8_quantize(a, afmt)
    bq, _, sb = K.fp8_quantize(b, torch.float8_e4m3fn)
    out = torch.empty((M, N), dtype=out_dtype, device=DEV)
    K.gemm(aq, bq, out, M, N, Kd, Kd, Kd, N, bias=bias, residual=res, ld_res=N, scale_a=sa, scale_b=sb)
    ref = (aq.float() * sa) @ (bq.float() * sb).t() + bias
    if res is not None:
        ref = ref + res
    assert rel_l2(out.float(), ref) < (2e-5 if out_dtype == torch.float32 else 4e-3)
    # and the quantisation error itself is what fp8 promises (e4m3: 3 mantissa bits, e5m2: 2)
    exact = a.float() @ b.float().t() + bias + (res if res is not None else 0)
    assert rel_l2(out.float(), exact) < (0.06 if afmt == torch.float8_e4m3fn else 0.12)


def test_fp8_gemm_split_k_weight_gradient_shape(K):
    """dW = dY^T X through the NT fp8 kernel: both operands transposed (k = tokens, zero-padded to 64),
    split-K with fp32 atomics, accumulate into an existing buffer."""
    g = torch.Generator(device=DEV).manual_seed(9)
    T, N, Kd = 2 * 577, 384, 128
    dy = _rand((T, N), torch.bfloat16, g)
    x = _rand((T, Kd), torch.bfloat16, g)
    db = torch.zeros(N, device=DEV)
    _, dyt, sdy = K.fp8_quantize(dy, torch.float8_e5m2, want=False, want_t=True, colsum=db)
    _, xt, sx = K.fp8_quantize(x, torch.float8_e4m3fn, want=False, want_t=True)
    Tp = dyt.shape[1]
    assert Tp % 64 == 0 and xt.shape[1] == Tp
    dw = torch.ones((N, Kd), device=DEV)
    K.gemm(dyt, xt, dw, N, Kd, Tp, Tp, Tp, Kd, accumulate=True, scale_a=sdy, scale_b=sx)
    ref = 1.0 + (dyt.float() * sdy) @ (xt.float() * sx).t()
    assert rel_l2(dw, ref) < 2e-5
    assert rel_l2(db, dy.float().sum(0)) < 1e-5


def _abi():
    import importlib
    return importlib.import_module("focused-attention-vit_amd")._abi


@pytest.mark.parametrize("M,N,Kd", [(128, 10, 192), (256, 10, 384), (300, 64, 1000), (1, 3, 7), (64, 1000 // 16, 768)])
def test_small_linear_head_kernels(K, M, N, Kd):
    """favit_small_linear_fwd / _bwd (the classification head as exact-fp32 dot products, one launch each way): against
    fp64, from zero and accumulating into existing gradient buffers, with and without dx / bias."""
    g = torch.Generator(device=DEV).manual_seed(3)
    x = _rand((M, Kd), torch.float32, g)
    w = _rand((N, Kd), torch.float32, g)
    b = _rand((N,), torch.float32, g)
    dy = _rand((M, N), torch.float32, g)
    y = K.small_linear_fwd(x, w, b)
    assert rel_l2(y.double(), x.double() @ w.double().t() + b.double()) < 1e-6
    assert rel_l2(K.small_linear_fwd(x, w, None).double(), x.double() @ w.double().t()) < 1e-6
    dx, dw, db = K.small_linear_bwd(dy, x, w)
    assert rel_l2(dx.double(), dy.double() @ w.double()) < 1e-6
    assert rel_l2(dw.double(), dy.double().t() @ x.double()) < 1e-6
    assert rel_l2(db.double(), dy.double().sum(0)) < 1e-6
    gw, gb = dw.clone(), db.clone()
    dx2, dw2, db2 = K.small_linear_bwd(dy, x, w, want_dx=False, dw_out=gw, db_out=gb)
    assert dx2 is None and dw2 is None and db2 is None
    assert rel_l2(gw.double(), 2 * dy.double().t() @ x.double()) < 1e-6 and rel_l2(gb.double(), 2 * dy.double().sum(0)) < 1e-6
    dx3, dw3, db3 = K.small_linear_bwd(dy, x, w, want_db=False)
    assert db3 is None and torch.equal(dw3, dw) and torch.equal(dx3, dx)
    # a strided row view (x[:, 0] of a [M, L, K] stream) is accepted as it is
    big = _rand((M, 3, Kd), torch.float32, g)
    assert rel_l2(K.small_linear_fwd(big[:, 0], w, b).double(), big[:, 0].double() @ w.double().t() + b.double()) < 1e-6


@pytest.mark.parametrize("rows,D", [(1000, 768), (37, 384), (4099, 192)])
def test_layernorm_passes_quantise_for_the_fp8_gemms(K, rows, D):
    """favit_layernorm_fwd_q8 / _bwd_q8 (fp8 mode): the bf16 tensor a LayerNorm pass writes also leaves it quantised with
    the consumer site's delayed scale.  Over four calls with drifting magnitudes (values beyond the previous amax
    saturate) the bf16 output, the fp8 bytes, the scale and the amax history are BIT-IDENTICAL to the two-pass form
    (favit_layernorm_* followed by favit_fp8_quantize on the bf16 tensor)."""
    g = torch.Generator(device=DEV).manual_seed(5)
    gamma = 1 + 0.1 * _rand((D,), torch.float32, g)
    beta = 0.1 * _rand((D,), torch.float32, g)
    e4, e5 = torch.float8_e4m3fn, torch.float8_e5m2
    ha, hb, ga, gb = (K.Fp8History(torch.device(DEV)) for _ in range(4))
    for step in range(4):
        amp = (1.0, 3.0, 0.5, 2.0)[step]
        x = amp * _rand((rows, D), torch.float32, g) + 0.3
        gam_s = gamma * amp
        # forward: two passes / one pass
        y, mu, rs = K.layernorm_fwd(x, D, gam_s, beta, rows, D, torch.bfloat16)
        q_ref, _, s_ref = K.fp8_quantize(y, e4, hist=ha)
        y2, mu2, rs2 = K.layernorm_fwd(x, D, gam_s, beta, rows, D, torch.bfloat16, q8=(e4, hb))
        if step == 0:                                    # no history yet: the site's first call measures first (two passes)
            assert getattr(y2, "_favit_q8", None) is None
            q2, _, s2 = K.fp8_quantize(y2, e4, hist=hb)
        else:
            q2, _, s2 = y2._favit_q8[1]
        assert torch.equal(y, y2) and torch.equal(mu, mu2) and torch.equal(rs, rs2)
        assert torch.equal(q_ref.view(torch.uint8), q2.view(torch.uint8)) and torch.equal(s_ref, s2)
        assert torch.equal(ha.slots.view(3, -1).max(1).values, hb.slots.view(3, -1).max(1).values) and ha.calls == hb.calls
        # backward (low-precision copy of the stream gradient, with the dropout mask of the branch it feeds)
        dy = (amp * _rand((rows, D), torch.float32, g)).to(torch.bfloat16)
        dres = _rand((rows, D), torch.float32, g)
        drop = (0.1, 1234 + step) if step % 2 else (0.0, 0)
        dx, lp, dg, db = K.layernorm_bwd(dy, x, D, gam_s, mu, rs, rows, D, dres=dres, want_lp=True, lp_drop=drop)
        p_ref, _, t_ref = K.fp8_quantize(lp, e5, hist=ga)
        dx2, lp2, dg2, db2 = K.layernorm_bwd(dy, x, D, gam_s, mu, rs, rows, D, dres=dres, want_lp=True, lp_drop=drop, q8=(e5, gb))
        if step == 0:
            p2, _, t2 = K.fp8_quantize(lp2, e5, hist=gb)
        else:
            p2, _, t2 = lp2._favit_q8[1]
        assert torch.equal(dx, dx2) and torch.equal(lp, lp2) and torch.equal(dg, dg2) and torch.equal(db, db2)
        assert torch.equal(p_ref.view(torch.uint8), p2.view(torch.uint8)) and torch.equal(t_ref, t2)
        assert torch.equal(ga.slots.view(3, -1).max(1).values, gb.slots.view(3, -1).max(1).values)


@pytest.mark.parametrize("ak,bk", [(True, True), (True, False), (False, True), (False, False)])
@pytest.mark.parametrize("M,N,Kd", [(50432, 384, 384), (8192 + 40, 1152, 400), (32768 + 4, 200, 64), (36928, 768, 768)])
@pytest.mark.parametrize("epi", ["plain", "gelu", "dgelu", "res"])
def test_gemm_exact_fp32_dma_kernel(K, favit, ak, bk, M, N, Kd, epi):
    """The exact-fp32 256x128 DMA kernel (p4f: global_load_lds ring + v_mfma_f32_32x32x2_f32; the GEMMs of the fp32
    parity mode at benchmark sizes): every operand layout, ragged M and N tiles, fused epilogues, against an fp64
    product."""
    g = torch.Generator(device=DEV).manual_seed(91)
    a = _rand((M, Kd) if ak else (Kd, M), torch.float32, g)
    b = _rand((N, Kd) if bk else (Kd, N), torch.float32, g)
    bias = _rand((N,), torch.float32, g)
    out = torch.empty((M, N), dtype=torch.float32, device=DEV)
    ref = ((a if ak else a.t()).double() @ (b.t() if bk else b).double())
    lda, ldb = (Kd if ak else M), (Kd if bk else N)
    kw = dict(a_kmajor=ak, b_kmajor=bk)
    A = _abi()
    if epi == "plain":
        K.gemm(a, b, out, M, N, Kd, lda, ldb, N, **kw)
    elif epi == "gelu":
        pre = torch.empty_like(out)
        K.gemm(a, b, out, M, N, Kd, lda, ldb, N, bias=bias, act=A.ACT_GELU, aux_out=pre, ld_aux_out=N, **kw)
        ref = ref + bias.double()
        assert rel_l2(pre.double(), ref) < 2e-6
        ref = torch.nn.functional.gelu(ref)
    elif epi == "dgelu":
        pre = _rand((M, N), torch.float32, g)
        K.gemm(a, b, out, M, N, Kd, lda, ldb, N, act=A.ACT_DGELU, aux_in=pre, ld_aux_in=N, **kw)
        x = pre.double().requires_grad_(True)
        torch.nn.functional.gelu(x).sum().backward()
        ref = ref * x.grad
    else:
        res = _rand((M, N), torch.float32, g)
        K.gemm(a, b, out, M, N, Kd, lda, ldb, N, bias=bias, residual=res, ld_res=N, alpha=0.5, **kw)
        ref = 0.5 * ref + bias.double() + res.double()
    assert favit._abi.lib().favit_gemm_last_kernel().decode() in ("p4f", "p4f128")     # (256- or 128-row tiles: by balance)
    assert rel_l2(out.double(), ref) < 2e-6


@pytest.mark.parametrize("M,N,T", [(1536, 384, 50432), (384, 1536, 50432), (1152, 384, 50432), (768, 3072, 36928), (200, 136, 65536)])
def test_gemm_exact_fp32_dma_kernel_weight_gradients(K, favit, M, N, T):
    """The same kernel on the weight-gradient shape (both operands token-major, split over the tokens with fp32 atomics,
    fused bias gradient), from zero and accumulating into an existing gradient."""
    g = torch.Generator(device=DEV).manual_seed(92)
    dy = _rand((T, M), torch.float32, g)
    x = _rand((T, N), torch.float32, g)
    dw = torch.empty((M, N), dtype=torch.float32, device=DEV)
    db = torch.zeros((M,), dtype=torch.float32, device=DEV)
    K.gemm(dy, x, dw, M, N, T, M, N, N, a_kmajor=False, b_kmajor=False, a_rowsum=db)
    assert favit._abi.lib().favit_gemm_last_kernel().decode() in ("p4f", "p4f128")
    ref = dy.double().t() @ x.double()
    assert rel_l2(dw.double(), ref) < 2e-6
    assert rel_l2(db.double(), dy.double().sum(0)) < 2e-6
    K.gemm(dy, x, dw, M, N, T, M, N, N, a_kmajor=False, b_kmajor=False, a_rowsum=db, accumulate=True)
    assert rel_l2(dw.double(), 2 * ref) < 2e-6
    assert rel_l2(db.double(), 2 * dy.double().sum(0)) < 2e-6


@pytest.mark.parametrize("bk", [True, False])
@pytest.mark.parametrize("out_dtype,epi", [(torch.bfloat16, "gelu"), (torch.bfloat16, "dgelu"), (torch.float32, "res")])
def test_gemm_256x256_tile_kernel(K, bk, out_dtype, epi):
    """The 16-wave 256x256-tile kernel (NT, K >= 512, N multiple of 256, >= 256 tiles) with its fused epilogues and a
    ragged M; the mn-major-B twin of each case runs the 256x128 kernel on the same shape."""
    g = torch.Generator(device=DEV).manual_seed(77)
    M, N, Kd = 8192 + 40, 2048, 544
    a = _rand((M, Kd), torch.bfloat16, g)
    b = _rand((N, Kd) if bk else (Kd, N), torch.bfloat16, g)
    bias = _rand((N,), torch.float32, g)
    out = torch.empty((M, N), dtype=out_dtype, device=DEV)
    ref = a.float() @ (b.float().t() if bk else b.float())
    kw = dict(b_kmajor=bk)
    ldb = Kd if bk else N
    if epi == "gelu":
        pre = torch.empty_like(out)
        K.gemm(a, b, out, M, N, Kd, Kd, ldb, N, bias=bias, act=_abi().ACT_GELU, aux_out=pre, ld_aux_out=N, **kw)
        ref = ref + bias
        assert rel_l2(pre.float(), ref) < 1e-2
        ref = torch.nn.functional.gelu(ref)
    elif epi == "dgelu":
        pre = _rand((M, N), torch.bfloat16, g)
        K.gemm(a, b, out, M, N, Kd, Kd, ldb, N, act=_abi().ACT_DGELU, aux_in=pre, ld_aux_in=N, **kw)
        x = pre.float().requires_grad_(True)
        torch.nn.functional.gelu(x).sum().backward()
        ref = ref * x.grad
    else:
        res = _rand((M, N), torch.float32, g)
        K.gemm(a, b, out, M, N, Kd, Kd, ldb, N, bias=bias, residual=res, ld_res=N, **kw)
        ref = ref + bias + res
    assert rel_l2(out.float(), ref) < (1e-2 if out_dtype == torch.bfloat16 else 2e-5)


@pytest.mark.parametrize("bk", [True, False])
@pytest.mark.parametrize("M,N,Kd,forced", [(8192 + 40, 1024 + 128, 4096, False), (2048 + 40, 384, 512, True),
                                           (1024, 200, 128, True), (300, 128, 1536, True)])
@pytest.mark.parametrize("out_dtype,epi", [(torch.bfloat16, "gelu"), (torch.bfloat16, "dgelu"), (torch.float32, "res")])
def test_gemm_ping_pong_kernel(K, bk, M, N, Kd, forced, out_dtype, epi, monkeypatch):
    """The 12-wave ping-pong kernel (two MFMA halves one barrier apart + four loader waves): taken by itself for
    K >= 4096 problems the 256x256 kernel cannot serve, forced (FAVIT_GEMM_PP, read per call) on small / ragged /
    short-K shapes (2 .. 24 stages, ragged M and N, both B layouts, fused epilogues)."""
    if forced:
        monkeypatch.setenv("FAVIT_GEMM_PP", "1")
    if not bk and N % 8:
        pytest.skip("mn-major DMA images need N % 8 == 0")
    g = torch.Generator(device=DEV).manual_seed(M + N + Kd)
    a = _rand((M, Kd), torch.bfloat16, g)
    b = _rand((N, Kd) if bk else (Kd, N), torch.bfloat16, g)
    bias = _rand((N,), torch.float32, g)
    out = torch.empty((M, N), dtype=out_dtype, device=DEV)
    ref = a.float() @ (b.float().t() if bk else b.float())
    kw = dict(b_kmajor=bk)
    ldb = Kd if bk else N
    if epi == "gelu":
        pre = torch.empty_like(out)
        K.gemm(a, b, out, M, N, Kd, Kd, ldb, N, bias=bias, act=_abi().ACT_GELU, aux_out=pre, ld_aux_out=N, **kw)
        ref = ref + bias
        assert rel_l2(pre.float(), ref) < 1e-2
        ref = torch.nn.functional.gelu(ref)
    elif epi == "dgelu":
        pre = _rand((M, N), torch.bfloat16, g)
        K.gemm(a, b, out, M, N, Kd, Kd, ldb, N, act=_abi().ACT_DGELU, aux_in=pre, ld_aux_in=N, **kw)
        x = pre.float().requires_grad_(True)
        torch.nn.functional.gelu(x).sum().backward()
        ref = ref * x.grad
    else:
        res = _rand((M, N), torch.float32, g)
        K.gemm(a, b, out, M, N, Kd, Kd, ldb, N, bias=bias, residual=res, ld_res=N, **kw)
        ref = ref + bias + res
    assert rel_l2(out.float(), ref) < (1e-2 if out_dtype == torch.bfloat16 else 2e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,Kd", [(300, 200, 96), (1000, 256, 128), (21760 + 37, 384, 64), (50432, 1536, 384)])
def test_gemm_gelu_savegrad_and_mulaux_epilogues(K, dtype, M, N, Kd):
    """ACT_GELU_SAVEGRAD (out = GELU(v), aux_out = GELU'(v)) and ACT_MULAUX (out = v * aux_in): the pair the bf16 MLP
    chain uses so that the GELU arithmetic runs once; every epilogue implementation (128x128 vector / scalar, 256x128
    wave-private vector / scalar, quarter-tile tail) against torch's erf GELU and its autograd derivative."""
    A = _abi()
    g = torch.Generator(device=DEV).manual_seed(M + N)
    a = _rand((M, Kd), dtype, g)
    w = _rand((N, Kd), dtype, g)
    bias = _rand((N,), torch.float32, g)
    out = torch.empty((M, N), dtype=dtype, device=DEV)
    gd = torch.empty((M, N), dtype=dtype, device=DEV)
    K.gemm(a, w, out, M, N, Kd, Kd, Kd, N, bias=bias, act=A.ACT_GELU_SAVEGRAD, aux_out=gd, ld_aux_out=N)
    u = (a.float() @ w.float().t() + bias).requires_grad_(True)
    h = torch.nn.functional.gelu(u)
    h.sum().backward()
    tol = 2e-5 if dtype == torch.float32 else 1e-2
    assert rel_l2(out.float(), h.detach()) < tol
    assert rel_l2(gd.float(), u.grad) < tol
    # backward partner: dU = (dY . W2) * saved GELU'
    dy = _rand((M, Kd), dtype, g)
    w2 = _rand((Kd, N), dtype, g)                      # mn-major B: the input-gradient layout
    du = torch.empty((M, N), dtype=dtype, device=DEV)
    K.gemm(dy, w2, du, M, N, Kd, Kd, N, N, b_kmajor=False, act=A.ACT_MULAUX, aux_in=gd, ld_aux_in=N)
    ref = (dy.float() @ w2.float()) * gd.float()
    assert rel_l2(du.float(), ref) < tol
    with pytest.raises(Exception):                     # SAVEGRAD without a destination for the derivative
        K.gemm(a, w, out, M, N, Kd, Kd, Kd, N, act=A.ACT_GELU_SAVEGRAD)


@pytest.mark.parametrize("bk", [True, False])
@pytest.mark.parametrize("M,N,Kd", [(50432, 384, 384), (50432 + 100, 384, 1536), (45000 - 8, 1152, 128), (131072 + 64 * 3 + 5, 200, 64)])
@pytest.mark.parametrize("out_dtype,epi", [(torch.bfloat16, "gelu"), (torch.bfloat16, "dgelu"), (torch.float32, "res")])
def test_gemm_quarter_tile_tail(K, bk, M, N, Kd, out_dtype, epi):
    """256x128-tile launches whose last round would fill at most 60 % of the 512 workgroup slots run that round as
    64x128 quarter tiles (blocks past the full rounds): the benchmark's N = 384 GEMMs (591 tiles = 512 + 79), a
    ragged M whose last panel is a partial quarter, a multi-round launch (1584 = 3 * 512 + 48 tiles), a ragged N;
    both B layouts and the fused epilogues."""
    if not bk and N % 8:
        pytest.skip("mn-major DMA images need N % 8 == 0")
    t4 = ((M + 255) // 256) * ((N + 127) // 128)
    assert t4 > 512 and 0 < t4 % 512 <= 307            # the shapes above must reach the quarter-tile path
    g = torch.Generator(device=DEV).manual_seed(M % 1000 + N + Kd)
    a = _rand((M, Kd), torch.bfloat16, g)
    b = _rand((N, Kd) if bk else (Kd, N), torch.bfloat16, g)
    bias = _rand((N,), torch.float32, g)
    out = torch.empty((M, N), dtype=out_dtype, device=DEV)
    ref = a.float() @ (b.float().t() if bk else b.float())
    kw = dict(b_kmajor=bk)
    ldb = Kd if bk else N
    if epi == "gelu":
        pre = torch.empty_like(out)
        K.gemm(a, b, out, M, N, Kd, Kd, ldb, N, bias=bias, act=_abi().ACT_GELU, aux_out=pre, ld_aux_out=N, **kw)
        ref = ref + bias
        assert rel_l2(pre.float(), ref) < 1e-2
        ref = torch.nn.functional.gelu(ref)
    elif epi == "dgelu":
        pre = _rand((M, N), torch.bfloat16, g)
        K.gemm(a, b, out, M, N, Kd, Kd, ldb, N, act=_abi().ACT_DGELU, aux_in=pre, ld_aux_in=N, **kw)
        x = pre.float().requires_grad_(True)
        torch.nn.functional.gelu(x).sum().backward()
        ref = ref * x.grad
    else:
        res = _rand((M, N), torch.float32, g)
        K.gemm(a, b, out, M, N, Kd, Kd, ldb, N, bias=bias, residual=res, ld_res=N, **kw)
        ref = ref + bias + res
    tol = 1e-2 if out_dtype == torch.bfloat16 else 2e-5
    assert rel_l2(out.float(), ref) < tol
    # the tail rows on their own (a wrong quarter mapping would hide in the whole-matrix norm)
    tail_rows = slice((M // 256 - 30) * 256, M)
    assert rel_l2(out[tail_rows].float(), ref[tail_rows]) < tol
    assert torch.isfinite(out.float()).all()


# ---------------------------------------------------------------------------------------------
# Seeded random sweeps: every dispatch branch of favit_gemm (register-staged / DMA 128x128 / 256x128 /
# 256x256 / split-K / fp32) is reached by some shape below; ragged sizes and odd leading dimensions.
# ---------------------------------------------------------------------------------------------
def _sweep_cases(n, seed):
    rs = np.random.RandomState(seed)
    out = []
    for i in range(n):
        kind = i % 4
        if kind == 0:       # tiny / ragged
            M, N, Kd = int(rs.randint(1, 200)), int(rs.randint(1, 200)), int(rs.randint(1, 200))
        elif kind == 1:     # DMA-eligible mid sizes (K multiple of 64)
            M, N, Kd = int(rs.randint(100, 1500)), int(rs.randint(8, 160) * 8), int(rs.randint(1, 8) * 64)
        elif kind == 2:     # big-M (256x128 / 256x256 kernels)
            M, N, Kd = int(rs.randint(20000, 40000)), int(rs.choice([256, 384, 512, 1000, 1152])), int(rs.choice([64, 96, 512, 544]))
        else:               # weight-gradient shape (long K, split-K)
            M, N, Kd = int(rs.choice([64, 192, 384, 1000])), int(rs.choice([48, 192, 384])), int(rs.randint(3000, 20000))
        out.append((M, N, Kd, bool(rs.randint(2)), bool(rs.randint(2)), int(rs.randint(3)), i))
    return out


@pytest.mark.parametrize("M,N,Kd,ak,bk,variant,idx", _sweep_cases(48, 2024))
def test_gemm_random_sweep(K, M, N, Kd, ak, bk, variant, idx):
    dtype = torch.bfloat16 if idx % 3 else torch.float32
    g = torch.Generator(device=DEV).manual_seed(1000 + idx)
    pad_a, pad_b = (0, 8)[idx % 2], (8, 0)[idx % 2]           # leading dimensions larger than the row
    Afull = _rand(((M, Kd + pad_a) if ak else (Kd, M + pad_a)), dtype, g)
    Bfull = _rand(((N, Kd + pad_b) if bk else (Kd, N + pad_b)), dtype, g)
    A = Afull[:, :Kd] if ak else Afull[:, :M]
    B = Bfull[:, :Kd] if bk else Bfull[:, :N]
    Af = (A.float() if ak else A.float().t()).double()
    Bf = (B.float() if bk else B.float().t()).double()
    ref = Af @ Bf.t()
    bias = _rand((N,), torch.float32, g)
    if variant == 0:        # plain fp32 output
        C = torch.empty((M, N), dtype=torch.float32, device=DEV)
        K.gemm(Afull, Bfull, C, M, N, Kd, Afull.stride(0), Bfull.stride(0), N, a_kmajor=ak, b_kmajor=bk)
        assert rel_l2(C, ref) < 3e-5
    elif variant == 1:      # bias, output in the input dtype
        C = torch.empty((M, N), dtype=dtype, device=DEV)
        K.gemm(Afull, Bfull, C, M, N, Kd, Afull.stride(0), Bfull.stride(0), N, a_kmajor=ak, b_kmajor=bk, bias=bias)
        assert rel_l2(C.float(), ref + bias.double()) < (3e-5 if dtype == torch.float32 else 6e-3)
    else:                   # accumulate into an existing fp32 buffer with alpha
        C0 = _rand((M, N), torch.float32, g)
        C = C0.clone()
        K.gemm(Afull, Bfull, C, M, N, Kd, Afull.stride(0), Bfull.stride(0), N, a_kmajor=ak, b_kmajor=bk,
               accumulate=True, alpha=0.25)
        assert rel_l2(C, C0.double() + 0.25 * ref) < 3e-5


def _attn_cases(n, seed):
    rs = np.random.RandomState(seed)
    out = []
    for i in range(n):
        W = int(rs.choice([3, 5, 7, 9, 11]))
        hd = int(rs.choice([16, 32, 64, 128]))
        L = int(rs.choice([1, 2, 3, W - 1, W, W + 1, 16, 17, 31, 50, 64, 65, 130, 197, 300])) or 1
        out.append((max(1, L), W, hd, int(rs.randint(1, 4)), int(rs.randint(1, 4)), bool(rs.randint(2)), i))
    return out


@pytest.mark.parametrize("L,W,hd,B,H,masked,idx", _attn_cases(40, 7))
def test_mhla_core_random_sweep(K, L, W, hd, B, H, masked, idx):
    """Random (L, W, hd, B, H, mask) against the gather-based window restatement, fwd + bwd, both dtypes."""
    from oracle import favit_oracle as O
    dtype = torch.bfloat16 if idx % 2 else torch.float32
    D = H * hd
    g = torch.Generator(device=DEV).manual_seed(500 + idx)
    qkv = _rand((B * L, 3 * D), dtype, g)
    dout = _rand((B * L, D), dtype, g)
    mask = None
    if masked:
        mask = (torch.rand(B, L, L, generator=g, device=DEV) > 0.4)
        mask |= torch.eye(L, dtype=torch.bool, device=DEV)
        mask = mask.to(torch.uint8).contiguous()
    out = K.mhla_attn_fwd(qkv, B, L, H, hd, W, mask)
    dqkv = K.mhla_attn_bwd(qkv, dout, B, L, H, hd, W, mask)
    idxs = torch.from_numpy(O.window_indices(L, W)).to(DEV)
    t = qkv.float().reshape(B, L, 3, H, hd).permute(2, 0, 3, 1, 4).detach().clone().requires_grad_(True)
    q, k, v = t[0], t[1], t[2]
    kw, vw = k[:, :, idxs], v[:, :, idxs]
    s = (q.unsqueeze(3) @ kw.transpose(-2, -1)).squeeze(3) / (hd ** 0.5)
    if mask is not None:
        wm = torch.gather(mask[:, None].expand(B, H, L, L), 3, idxs[None, None].expand(B, H, L, W))
        s = s.masked_fill(wm == 0, float("-inf"))
    o = (torch.softmax(s, -1).unsqueeze(3) @ vw).squeeze(3).transpose(1, 2).reshape(B * L, D)
    assert rel_l2(out.float(), o) < _tol(dtype)
    o.backward(dout.float())
    gref = t.grad.permute(1, 3, 0, 2, 4).reshape(B * L, 3 * D)
    assert rel_l2(dqkv.float(), gref) < (5e-5 if dtype == torch.float32 else 1.5e-2)
    if K.mhla_attn_lse_supported(L, hd, W, dtype):            # the training path's pair on the same case
        out2, lse = K.mhla_attn_fwd(qkv, B, L, H, hd, W, mask, want_lse=True)
        assert torch.equal(out2, out)
        dq2 = K.mhla_attn_bwd(qkv, dout, B, L, H, hd, W, mask, o=out2, lse=lse)
        assert rel_l2(dq2.float(), gref) < 1.5e-2


def _lse_cases():
    rs = np.random.RandomState(11)
    out = []
    for i in range(36):
        W = int(rs.choice([3, 5, 7, 9, 11]))
        lo = max(W + 1, 17 if W > 7 else 0)
        L = int(rs.choice([lo, lo + 1, 16, 17, 31, 32, 33, 47, 48, 49, 63, 64, 65, 66, 127, 128, 129, 197, 200, 577]))
        out.append((max(L, lo), W, int(rs.randint(1, 3)), int(rs.randint(1, 4)), int(rs.randint(3)), i))
    return out


@pytest.mark.parametrize("L,W,B,H,mode,idx", _lse_cases())
@pytest.mark.parametrize("waves", [0, 1, 4])
def test_mhla_saved_statistics_backward_sweep(K, L, W, B, H, mode, idx, waves, monkeypatch):
    """The saved-statistics pair (hd = 64, bf16) over sequence lengths around every tile / block edge, plain, masked
    (mode 1) and with dropout (mode 2), at the default block size (3 tiles) and at 1 and 4 tiles per block: dqkv against the
    statistics-free MFMA kernel on the same inputs (masks and dropout draws are identical by construction)."""
    hd, dtype = 64, torch.bfloat16
    if waves:
        monkeypatch.setenv("FAVIT_MHLA_LSE_WAVES", str(waves))
    assert K.mhla_attn_lse_supported(L, hd, W, dtype)
    D = H * hd
    g = torch.Generator(device=DEV).manual_seed(900 + idx)
    qkv = _rand((B * L, 3 * D), dtype, g)
    dout = _rand((B * L, D), dtype, g)
    mask, p = None, 0.0
    if mode == 1:
        mask = (torch.rand(B, L, L, generator=g, device=DEV) > 0.4)
        mask |= torch.eye(L, dtype=torch.bool, device=DEV)
        mask = mask.to(torch.uint8).contiguous()
    if mode == 2:
        p = 0.2
    out, lse = K.mhla_attn_fwd(qkv, B, L, H, hd, W, mask, p, 5 + idx, want_lse=True)
    assert torch.equal(out, K.mhla_attn_fwd(qkv, B, L, H, hd, W, mask, p, 5 + idx))
    got = K.mhla_attn_bwd(qkv, dout, B, L, H, hd, W, mask, p, 5 + idx, o=out, lse=lse)
    ref = K.mhla_attn_bwd(qkv, dout, B, L, H, hd, W, mask, p, 5 + idx)
    assert torch.isfinite(got.float()).all()
    for part, name in enumerate(("dq", "dk", "dv")):
        a, b = got[:, part * D:(part + 1) * D].float(), ref[:, part * D:(part + 1) * D].float()
        assert rel_l2(a, b) < 8e-3, name
        # row by row: a wrong halo / wrap row would hide in the global norm
        err = (a - b).reshape(B, L, D).norm(dim=-1) / (b.reshape(B, L, D).norm(dim=-1) + 1e-3 * b.norm() / (B * L) ** 0.5)
        assert err.max().item() < 6e-2, (name, int(err.argmax()))


@pytest.mark.parametrize("seed", list(range(12)))
def test_sppp_map_pool_centroids_random_label_maps(K, seed):
    """Random label maps (blocky noise: ties inside patches, labels that never dominate a patch, gaps in the
    label range) against the oracle: the integer work (dominant label with smallest-label tie-break, first
    appearance rank) must be bit-exact; pooling / centroids to fp32 rounding."""
    from oracle import favit_oracle as O
    rs = np.random.RandomState(seed)
    P = int(rs.choice([4, 8, 16]))
    g = int(rs.choice([2, 5, 14]))
    img = P * g
    S = int(rs.choice([4, 9, 16]))
    cell = int(rs.choice([1, 2, P // 2, P, 2 * P]))
    base = rs.randint(0, S + 3, size=((img + cell - 1) // cell, (img + cell - 1) // cell))     # labels may exceed S
    seg = np.kron(base, np.ones((cell, cell), dtype=np.int64))[:img, :img].astype(np.int64)
    mapping = O.map_patches(seg, img, P)
    rank_ref = np.full(g * g, -1, dtype=np.int64)
    for r, (_, idx) in enumerate(mapping.items()):
        rank_ref[idx] = r
    segt = torch.from_numpy(seg).to(DEV)
    rank, ntok, perm, offs, dom = K.sppp_map_patches(segt[None], P)
    np.testing.assert_array_equal(rank[0].cpu().numpy(), rank_ref)
    R = int(ntok[0].item())
    assert R == len(mapping)
    D = 32
    emb = torch.randn(g * g, D, generator=torch.Generator().manual_seed(seed))
    for kind, kname in enumerate(("mean", "max", "attention")):
        ref = O.pool(emb.clone(), mapping, kname)
        out, _ = K.sppp_pool_fwd(emb.to(DEV)[None].contiguous(), perm, offs, kind, R)
        assert rel_l2(out[0], ref) < 2e-5, kname
    cent = K.sppp_centroids(segt[None], S)
    assert rel_l2(cent, O.superpixel_centroids(seg[None], S)) < 2e-5


@pytest.mark.parametrize("idx", list(range(16)))
def test_layernorm_random_sweep(K, idx):
    rs = np.random.RandomState(100 + idx)
    rows, D = int(rs.randint(1, 700)), int(rs.randint(1, 400) * 4)
    dtype = torch.bfloat16 if idx % 2 else torch.float32
    g = torch.Generator(device=DEV).manual_seed(idx)
    x = torch.randn((rows, D), generator=g, device=DEV) * float(rs.uniform(0.1, 30)) + float(rs.uniform(-5, 5))
    gam, bet = torch.randn(D, generator=g, device=DEV), torch.randn(D, generator=g, device=DEV)
    y, mu, rstd = K.layernorm_fwd(x, D, gam, bet, rows, D, dtype)
    xr = x.clone().requires_grad_(True)
    gr, br = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xr, (D,), gr, br, 1e-5)
    assert rel_l2(y.float(), yr) < _tol(dtype)
    dy = _rand((rows, D), dtype, g)
    dx, _, dg, db = K.layernorm_bwd(dy, x, D, gam, mu, rstd, rows, D)
    yr.backward(dy.float())
    assert rel_l2(dx, xr.grad) < 5e-5
    assert rel_l2(dg, gr.grad) < 5e-5 and rel_l2(db, br.grad) < 5e-5
