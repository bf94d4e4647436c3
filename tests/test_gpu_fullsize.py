"""GPU parity at the BENCHMARK shapes (BASELINE.json configs[1]: 256 images x 197 tokens = 50,432 token
rows, D = 384): the kernels bench.py times must be the kernels the parity tests check.

* favit_gemm_grouped_tn (the grouped weight-gradient launch, only taken when the token count is a multiple
  of 32 and the operands are bf16) against an fp64 dY^T X / column sum, at T = 50,432 and at a small
  qualifying T;
* the 256x128-tile kernel's fused epilogues (GELU + saved pre-activation, dGELU, residual, dropout) at
  p4-sized M with N = 384 / 1536, both B layouts;
* the whole cfg2 training step at B = 256, forward AND backward, in bf16 and fp32: the two golden cfg2 images
  tiled 128x (mean cross-entropy => the gradients equal the B = 2 gradients) against the B = 2 HIP gradients
  element-wise and against the reference's golden gradient norms (models.npz: cfg2/gnorm/*,
  experiments/mhla_pretrained.py:363-367 is the loop this mirrors).
"""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"
T_BENCH = 256 * 197
D = 384


@pytest.fixture(scope="module")
def K(favit):
    return favit.kernels


def _rand(shape, dtype, gen, scale=1.0):
    return (torch.randn(shape, generator=gen, device=DEV, dtype=torch.float32) * scale).to(dtype)


def _block_problems(T, gen, accumulate, D=D):
    """The four dW = dY^T X problems of one transformer block (fc2, fc1, proj, folded qkv)."""
    shapes = [(D, 4 * D), (4 * D, D), (D, D), (3 * D, D)]          # (N, K) of dW[N, K]
    probs, refs = [], []
    for N, Kd in shapes:
        dy = _rand((T, N), torch.bfloat16, gen, 0.5)
        x = _rand((T, Kd), torch.bfloat16, gen)
        if accumulate:
            dw0 = _rand((N, Kd), torch.float32, gen)
            db0 = _rand((N,), torch.float32, gen)
        else:
            dw0 = torch.full((N, Kd), float("nan"), device=DEV)      # must be overwritten, not added to
            db0 = torch.zeros(N, device=DEV)                          # a_rowsum is always ADDED (fused db)
        dw, db = dw0.clone(), db0.clone()
        probs.append((dy, x, dw, db, accumulate))
        ref_w = dy.double().t() @ x.double()
        ref_b = dy.double().sum(0)
        if accumulate:
            ref_w, ref_b = ref_w + dw0.double(), ref_b + db0.double()
        else:
            ref_b = ref_b + db0.double()
        refs.append((ref_w, ref_b))
    return probs, refs


@pytest.mark.parametrize("slabs", [True, False])
@pytest.mark.parametrize("accumulate", [False, True])
@pytest.mark.parametrize("T", [512, 2048 + 32, T_BENCH])
def test_gemm_grouped_tn_block_problems(K, T, accumulate, slabs):
    """slabs: the split-K partials go through the workspace and are summed in a fixed order (the product path);
    not slabs: fp32 atomics into the destination (the fallback without a workspace)."""
    gen = torch.Generator(device=DEV).manual_seed(T + int(accumulate))
    probs, refs = _block_problems(T, gen, accumulate)
    assert K.gemm_grouped_tn(probs, use_workspace=slabs), "the grouped launch must accept the block's four problems at T %% 32 == 0"
    torch.cuda.synchronize()
    for (dy, x, dw, db, _), (ref_w, ref_b) in zip(probs, refs):
        assert torch.isfinite(dw).all()
        assert rel_l2(dw, ref_w) < 2e-5, (tuple(dw.shape), rel_l2(dw, ref_w))
        assert rel_l2(db, ref_b) < 2e-5, (tuple(dw.shape), rel_l2(db, ref_b))


def test_gemm_grouped_tn_slab_reduction_is_deterministic(K):
    """Weight gradients through the workspace are bitwise reproducible (fixed summation order); through fp32
    atomics they are not guaranteed to be."""
    outs = []
    for _ in range(3):
        gen = torch.Generator(device=DEV).manual_seed(11)
        probs, _ = _block_problems(T_BENCH, gen, False)
        assert K.gemm_grouped_tn(probs)
        outs.append([(p[2].clone(), p[3].clone()) for p in probs])
    for o in outs[1:]:
        for (w0, b0), (w1, b1) in zip(outs[0], o):
            assert torch.equal(w0, w1) and torch.equal(b0, b1)


def test_gemm_grouped_tn_short_token_counts(K):
    """ViT-Tiny at 64 images x 65 tokens (BASELINE.json configs[0]): 20 tiles would like 24 K-splits, 4,160 tokens
    only feed 16 -- the launch takes the best feasible split count instead of declining."""
    gen = torch.Generator(device=DEV).manual_seed(5)
    probs, refs = _block_problems(64 * 65, gen, True, D=192)
    assert K.gemm_grouped_tn(probs)
    for (dy, x, dw, db, _), (ref_w, ref_b) in zip(probs, refs):
        assert rel_l2(dw, ref_w) < 2e-5 and rel_l2(db, ref_b) < 2e-5


def test_gemm_grouped_tn_single_problem_short_token_counts(K):
    """One problem per launch (fine-tuning with frozen layers leaves only the folded-qkv weight gradient of a block) at
    the two token counts of the cfg5 buckets: 8 and 32 K-splits."""
    for T in (128 * 17, 128 * 16):
        gen = torch.Generator(device=DEV).manual_seed(T)
        probs, refs = _block_problems(T, gen, False)
        assert K.gemm_grouped_tn(probs[3:])
        (dy, x, dw, db, _), (ref_w, ref_b) = probs[3], refs[3]
        assert rel_l2(dw, ref_w) < 2e-5 and rel_l2(db, ref_b) < 2e-5


def test_gemm_grouped_tn_workspace_outgrown_under_a_captured_graph(K):
    """A captured graph holds the ADDRESS of the slab workspace.  When a later launch needs a larger one, the old
    buffer must stay alive: torch.cuda.graph() empties the allocator cache at the next capture, and a freed workspace
    would be unmapped under the first graph's replays (the two-bucket cfg5 bench faulted exactly like this)."""
    gen = torch.Generator(device=DEV).manual_seed(21)
    small, refs = _block_problems(512, gen, False, D=192)
    K._GROUPED_WS.clear()                                  # the graph's launch allocates the workspace it captures
    assert K.gemm_grouped_tn(small)                        # warm-up outside the capture (allocates)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        assert K.gemm_grouped_tn(small)
    first = K._GROUPED_WS[torch.cuda.current_device()]
    big, _ = _block_problems(2048 + 32, gen, False)        # D = 384, more tokens: a larger workspace
    assert K.gemm_grouped_tn(big)
    assert K._GROUPED_WS[torch.cuda.current_device()] is not first and any(w is first for w in K._GROUPED_WS_RETIRED)
    torch.cuda.synchronize()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):                             # (empties the allocator cache on entry)
        assert K.gemm_grouped_tn(big)
    for p in small:
        p[2].fill_(float("nan"))
        p[3].zero_()
    g.replay()
    torch.cuda.synchronize()
    for (dy, x, dw, db, _), (ref_w, ref_b) in zip(small, refs):
        assert rel_l2(dw, ref_w) < 2e-5 and rel_l2(db, ref_b) < 2e-5


def test_gemm_grouped_tn_declines_what_it_cannot_group(K):
    gen = torch.Generator(device=DEV).manual_seed(3)
    T = 197 * 2                                  # not a multiple of 32: callers fall back to single launches
    probs, _ = _block_problems(T, gen, False)
    assert K.gemm_grouped_tn(probs) is False
    probs, refs = _block_problems(224, gen, False)  # seven k-steps: few splits (round 3 declined what 8 splits could not take)
    assert K.gemm_grouped_tn(probs)
    for (dy, x, dw, db, _), (ref_w, ref_b) in zip(probs, refs):
        assert rel_l2(dw, ref_w) < 2e-5 and rel_l2(db, ref_b) < 2e-5


@pytest.mark.parametrize("T,Dm,nblocks,accumulate,want_splits", [
    (128 * 17, 384, 12, True, 1),        # cfg3: the whole encoder's 48 problems in one launch, NO K-split, no workspace
    (128 * 17, 384, 12, False, 1),
    (64 * 65, 192, 12, True, None),      # cfg1
    (128 * 17, 384, 4, True, None),      # a data-parallel / graph-segment group of four blocks at 17 tokens
    (T_BENCH, 384, 4, True, 8),          # cfg2: four blocks = 16 problems, 8 splits of 6,304 tokens (32 units: four per XCD), slabs
    (T_BENCH, 384, 3, False, None),      # a ragged group (12 blocks = 4 + 4 + 4, but 10 = 4 + 4 + 2 ...)
])
def test_gemm_grouped_tn_several_blocks_per_launch(K, favit, T, Dm, nblocks, accumulate, want_splits):
    """Round 4: functional.flush_wgrads hands the weight gradients of SEVERAL blocks to one grouped launch.  Against
    fp64; the split count the cost model picks is what the design says for the two benchmark cases; results are
    bitwise reproducible (slab reduction in split order, or a single accumulation chain with one split)."""
    gen = torch.Generator(device=DEV).manual_seed(T + nblocks)
    probs, refs = [], []
    for _ in range(nblocks):
        p_, r_ = _block_problems(T, gen, accumulate, D=Dm)
        probs += p_
        refs += r_
    start = [(p[2].clone(), p[3].clone()) for p in probs]
    assert K.gemm_grouped_tn(probs)
    splits = int(favit._abi.lib().favit_gemm_grouped_last_splits())
    if want_splits is not None:
        assert splits == want_splits, splits
    torch.cuda.synchronize()
    for (dy, x, dw, db, _), (ref_w, ref_b) in zip(probs, refs):
        assert torch.isfinite(dw).all()
        assert rel_l2(dw, ref_w) < 2e-5, (tuple(dw.shape), rel_l2(dw, ref_w))
        assert rel_l2(db, ref_b) < 2e-5, (tuple(dw.shape), rel_l2(db, ref_b))
    first = [(p[2].clone(), p[3].clone()) for p in probs]
    for p, (w0, b0) in zip(probs, start):
        p[2].copy_(w0)
        p[3].copy_(b0)
    assert K.gemm_grouped_tn(probs)
    torch.cuda.synchronize()
    for p, (w1, b1) in zip(probs, first):
        assert torch.equal(p[2], w1), "weight gradients must be bitwise reproducible"
        if splits > 1:
            assert torch.equal(p[3], b1)          # (one split: the bias gradient is one fp32 atomic per row -- exact too)
        else:
            assert torch.equal(p[3], b1)


def _abi(favit):
    return favit._abi


@pytest.mark.parametrize("bk", [True, False])
@pytest.mark.parametrize("epi,N,Kd", [("gelu_aux", 1536, 384), ("dgelu", 1536, 384), ("dgelu_drop", 1536, 384),
                                      ("res_f32", 384, 1536), ("res_f32", 384, 384), ("res_drop_f32", 384, 384),
                                      ("plain_bf16", 384, 1152), ("plain_bf16", 1152, 384)])
def test_gemm_p4_epilogues_at_bench_sized_m(K, favit, bk, epi, N, Kd):
    """M large enough that favit_gemm takes the 256x128-tile kernel for N = 384 (>= 256 tiles), ragged in M."""
    abi = _abi(favit)
    M = 86 * 256 + 40
    gen = torch.Generator(device=DEV).manual_seed(N * 7 + Kd + int(bk))
    a = _rand((M, Kd), torch.bfloat16, gen)
    b = _rand((N, Kd) if bk else (Kd, N), torch.bfloat16, gen, 0.05)
    ldb = Kd if bk else N
    bias = _rand((N,), torch.float32, gen)
    base = a.float() @ (b.float().t() if bk else b.float())
    kw = dict(b_kmajor=bk)
    if epi == "gelu_aux":
        out = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
        pre = torch.empty_like(out)
        K.gemm(a, b, out, M, N, Kd, Kd, ldb, N, bias=bias, act=abi.ACT_GELU, aux_out=pre, ld_aux_out=N, **kw)
        assert rel_l2(pre.float(), base + bias) < 1e-2
        assert rel_l2(out.float(), torch.nn.functional.gelu(base + bias)) < 1e-2
    elif epi in ("dgelu", "dgelu_drop"):
        pre = _rand((M, N), torch.bfloat16, gen)
        out = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
        p, seed = (0.2, 4242) if epi == "dgelu_drop" else (0.0, 0)
        K.gemm(a, b, out, M, N, Kd, Kd, ldb, N, act=abi.ACT_DGELU, aux_in=pre, ld_aux_in=N, dropout_p=p,
               dropout_seed=seed, **kw)
        x = pre.float().requires_grad_(True)
        torch.nn.functional.gelu(x).sum().backward()
        ref = base * x.grad
        if p > 0:
            ref = K.dropout(ref, p, seed)             # the same counter-based mask (index = m * N + n)
            assert abs((out == 0).float().mean().item() - p) < 0.01
        assert rel_l2(out.float(), ref) < 1e-2
    elif epi in ("res_f32", "res_drop_f32"):
        res = _rand((M, N), torch.float32, gen)
        out = torch.empty((M, N), dtype=torch.float32, device=DEV)
        p, seed = (0.1, 99) if epi == "res_drop_f32" else (0.0, 0)
        K.gemm(a, b, out, M, N, Kd, Kd, ldb, N, bias=bias, residual=res, ld_res=N, dropout_p=p, dropout_seed=seed, **kw)
        y = base + bias
        if p > 0:
            y = K.dropout(y, p, seed)
        assert rel_l2(out, y + res) < 2e-5
    else:
        out = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
        K.gemm(a, b, out, M, N, Kd, Kd, ldb, N, bias=bias, **kw)
        assert rel_l2(out.float(), base + bias) < 1e-2


MD = load_golden("models.npz")


def _cfg2_model(favit):
    torch.manual_seed(1234)
    m = favit.models.vit_mhla.VisionTransformerMHLA(img_size=224, patch_size=16, num_classes=1000, embed_dim=384,
                                                    depth=12, num_heads=6, window_size=7, use_mhla=True)
    x = torch.randn(2, 3, 224, 224)
    y = torch.randint(0, 1000, (2,))
    assert abs(x.double().sum().item() - float(MD["cfg2/x_sum"])) < 1e-6 and torch.equal(y, torch.from_numpy(MD["cfg2/y"]))
    return m, x, y


@pytest.mark.parametrize("mode,tol_elem,tol_gn,tol_logits", [("fp32", 2e-4, 2e-3, 1e-3), ("bf16", 1.5e-2, 5e-2, 2e-2)])
def test_full_size_cfg2_forward_backward_matches_golden(favit, K, mode, tol_elem, tol_gn, tol_logits):
    """B = 256 forward + backward through the fused-optimizer flow of bench.py (flat .grad buffers, direct
    gradient accumulation, grouped weight-gradient launch) == the B = 2 result == the reference's golden run.
    bf16 element-wise tolerance: 1.5e-2 since round 4 -- the B = 2 pass (394 token rows) runs its N = 384 projections
    in the split-K-inside-the-workgroup kernel (s64k2: even and odd k-steps summed separately), the B = 256 pass in the
    256x128-tile kernel (one chain): the fp32 sums differ in the last bit, some bf16 roundings of the products flip, and
    twelve layers later the gradients differ by 0.5-0.8 % (round 3: identical summation order in both kernels, 2e-3).
    The fp32 mode (exact-fp32 kernels, one chain everywhere) keeps 2e-4."""
    favit.set_compute_dtype(mode)
    try:
        m, x, y = _cfg2_model(favit)
        m.to(DEV).train()                                   # dropout = 0 in this configuration: train == eval math
        x, y = x.to(DEV), y.to(DEV)
        # B = 2, plain autograd gradients
        logits2 = m(x)
        assert rel_l2(logits2.detach().cpu(), MD["cfg2/logits"]) < tol_logits
        favit.train.cross_entropy(logits2, y).backward()
        g2 = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
        for p in m.parameters():
            p.grad = None
        # B = 256: the pair tiled 128x, flat gradient buffers as in bench.py
        opt = favit.train.FusedAdamW(favit.train.param_groups(m, lr=1e-4), distributed=False)
        xb, yb = x.repeat(128, 1, 1, 1).contiguous(), y.repeat(128).contiguous()
        opt.zero_grad()
        K.GEMM_TRACE = []
        try:
            logits = m(xb)
            loss = favit.train.cross_entropy(logits, yb)
            loss.backward()
            torch.cuda.synchronize()
            keys = {t[3] for t in K.GEMM_TRACE}
        finally:
            K.GEMM_TRACE = None
        if mode == "bf16":
            assert "bf16_MM_of32_grouped" in keys, f"the grouped weight-gradient kernel did not run: {sorted(keys)}"
        assert rel_l2(logits[:2].detach().cpu(), MD["cfg2/logits"]) < tol_logits
        assert rel_l2(logits[254:].detach().cpu(), MD["cfg2/logits"]) < tol_logits
        assert abs(loss.item() - float(MD["cfg2/loss"])) < tol_logits * abs(float(MD["cfg2/loss"]))
        worst_e, worst_n = 0.0, 0.0
        for k, p in m.named_parameters():
            assert p.grad is not None and torch.isfinite(p.grad).all(), k
            e = rel_l2(p.grad, g2[k])
            worst_e = max(worst_e, e)
            assert e < tol_elem, f"{k}: B=256 vs B=2 gradient rel-L2 {e}"
            r = float(MD[f"cfg2/gnorm/{k}"])
            n = abs(p.grad.norm().item() - r) / max(r, 1e-12)
            worst_n = max(worst_n, n)
            assert n < tol_gn, f"{k}: gradient norm off the reference's by {n}"
        print(f"[{mode}] worst element-wise rel-L2 vs B=2: {worst_e:.2e}; worst gradient-norm deviation vs golden: {worst_n:.2e}")
    finally:
        favit.set_compute_dtype("fp32")
        favit.functional.clear_lp_mirrors()


def test_fused_optimizer_mirror_sees_external_weight_edits(favit):
    """Once a FusedAdamW owns a bf16 mirror of the weights, load_state_dict / nn.init / p.copy_ (version
    counter) and p.data edits + invalidate_weight_cache() (epoch) must reach the next bf16 forward."""
    favit.set_compute_dtype("bf16")
    try:
        torch.manual_seed(0)
        mk = lambda: favit.models.vit_mhla.VisionTransformerMHLA(img_size=32, patch_size=4, num_classes=10, embed_dim=64,
                                                                 depth=2, num_heads=4, use_mhla=True).to(DEV)
        m, other = mk(), mk()
        x = torch.randn(4, 3, 32, 32, device=DEV)
        y = torch.randint(0, 10, (4,), device=DEV)
        opt = favit.train.FusedAdamW(favit.train.param_groups(m, lr=1e-2), lr=1e-2, distributed=False)
        favit.train.train_step(m, x, y, opt)
        m.eval(), other.eval()
        with torch.no_grad():
            want = other(x).float()
            assert rel_l2(m(x).float(), want) > 0.05              # the two models differ
            m.load_state_dict(other.state_dict())                 # copy_ into the flat-buffer views
            assert rel_l2(m(x).float(), want) < 1e-6, "stale bf16 mirror after load_state_dict"
            torch.nn.init.zeros_(m.head.bias)
            torch.nn.init.zeros_(other.head.bias)
            assert rel_l2(m(x).float(), other(x).float()) < 1e-6, "stale bf16 mirror after nn.init"
            m.head.weight.data.mul_(2.0)                          # invisible to the version counter
            other.head.weight.data.mul_(2.0)
            favit.invalidate_weight_cache()
            assert rel_l2(m(x).float(), other(x).float()) < 1e-6, "stale bf16 mirror after invalidate_weight_cache"
        # and training continues from the edited weights
        m.train()
        l0 = favit.train.train_step(m, x, y, opt).item()
        l1 = favit.train.train_step(m, x, y, opt).item()
        assert np.isfinite(l0) and l1 < l0
    finally:
        favit.set_compute_dtype("fp32")
        favit.functional.clear_lp_mirrors()


def test_graphed_step_sees_external_weight_edits(favit):
    """The replayed graph reads the bf16 mirror directly (round 4: no per-step re-cast inside the graph); GraphedStep
    checks the parameters' version counters / the weight epoch on the host before every replay instead.  Weights loaded
    or edited between two replays must be the weights the next replay computes with."""
    favit.set_compute_dtype("bf16")
    try:
        torch.manual_seed(0)
        mk = lambda: favit.models.vit_mhla.VisionTransformerMHLA(img_size=32, patch_size=4, num_classes=10, embed_dim=64,
                                                                 depth=2, num_heads=4, use_mhla=True).to(DEV).train()
        m, other = mk(), mk()
        x = torch.randn(8, 3, 32, 32, device=DEV)
        y = torch.randint(0, 10, (8,), device=DEV)
        opt = favit.train.FusedAdamW(favit.train.param_groups(m, lr=0.0), lr=0.0, weight_decay=0.0, distributed=False)
        oo = favit.train.FusedAdamW(favit.train.param_groups(other, lr=0.0), lr=0.0, weight_decay=0.0, distributed=False)
        step = favit.train.GraphedStep(m, opt, x, y)
        want = favit.train.train_step(other, x, y, oo).item()            # lr = 0: the weights stay what they are
        l_own = step(x, y).item()
        assert abs(l_own - want) > 1e-3                                   # two different models
        m.load_state_dict(other.state_dict())                            # version counters
        assert abs(step(x, y).item() - want) < 1e-5, "replay used a stale bf16 mirror after load_state_dict"
        with torch.no_grad():
            m.head.weight.data.mul_(3.0)                                  # invisible to the counters ...
            other.head.weight.data.mul_(3.0)
        favit.invalidate_weight_cache()                                   # ... hence the explicit invalidation
        want2 = favit.train.train_step(other, x, y, oo).item()
        assert abs(want2 - want) > 1e-4
        assert abs(step(x, y).item() - want2) < 1e-5, "replay used a stale bf16 mirror after invalidate_weight_cache"
    finally:
        favit.set_compute_dtype("fp32")
        favit.functional.clear_lp_mirrors()


def test_cross_entropy_rejects_bad_arguments(favit, K):
    logits = torch.randn(4, 10, device=DEV)
    with pytest.raises(TypeError):
        K.cross_entropy(logits, torch.zeros(4, dtype=torch.int32, device=DEV))
    with pytest.raises(TypeError):
        K.cross_entropy(logits.to(torch.bfloat16), torch.zeros(4, dtype=torch.int64, device=DEV))
    with pytest.raises(TypeError):
        K.cross_entropy(logits, torch.zeros(3, dtype=torch.int64, device=DEV))
    rows, _ = K.cross_entropy(logits, torch.tensor([0, -100, 3, 10], device=DEV), grad_scale=0.25)
    assert torch.isfinite(rows[0]) and torch.isfinite(rows[2]) and torch.isnan(rows[1]) and torch.isnan(rows[3])
