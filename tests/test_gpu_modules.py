"""GPU parity of the nn.Module mirrors against the golden vectors captured from the reference
(tests/golden/*.npz).  fp32 compute mode: forward 1e-4 rel-L2, gradients 5e-4 (north_star:
logits within 1e-3 rel of the reference); bf16 mode: stated tolerance 2e-2 (the reference itself
run in bf16 drifts ~1e-2 from its fp32 result)."""
import numpy as np
import pytest
import torch

from conftest import case, load_golden, rel_l2, sd_of

pytestmark = pytest.mark.gpu
DEV = "cuda"
D, H = 64, 4
FWD_TOL, GRAD_TOL = 1e-4, 5e-4


@pytest.fixture(autouse=True)
def _fp32_mode(favit):
    favit.set_compute_dtype("fp32")
    yield
    favit.set_compute_dtype("fp32")


def _run(module, c, n_in=1, extra=None, fwd_tol=FWD_TOL, grad_tol=GRAD_TOL, check_grads=True):
    module.load_state_dict(sd_of(c))
    module.to(DEV).eval()
    ins = [torch.from_numpy(c[f"in{i}"]).to(DEV).requires_grad_(True) for i in range(n_in)]
    kw = {k: v.to(DEV) for k, v in (extra or {}).items()}
    y = module(*ins, **kw)
    assert rel_l2(y.detach().cpu(), c["out"]) < fwd_tol, "forward"
    if not check_grads:
        return
    (y * torch.from_numpy(c["gout"]).to(DEV)).sum().backward()
    for i, t in enumerate(ins):
        assert rel_l2(t.grad.cpu(), c[f"gin{i}"]) < grad_tol, f"grad input {i}"
    for k, p in module.named_parameters():
        ref = c[f"grad/{k}"]
        g = p.grad.cpu() if p.grad is not None else torch.zeros_like(p).cpu()
        if np.abs(ref).max() < 1e-5:
            assert g.abs().max().item() < 1e-4, k
        else:
            assert rel_l2(g, ref) < grad_tol, f"grad {k}: {rel_l2(g, ref)}"


MHLA = load_golden("mhla.npz")
VP = load_golden("vit_parts.npz")
CR = load_golden("cross.npz")
SP = load_golden("sppp.npz")
MD = load_golden("models.npz")


@pytest.mark.parametrize("name", sorted({k.split("/")[0] for k in MHLA.files if k.startswith("attn_L")}))
def test_mhla_attention(favit, name):
    W = int(name.split("_W")[1])
    _run(favit.models.mhla.MultiHeadLatentAttention(D, H, window_size=W), case(MHLA, name))


@pytest.mark.parametrize("name", sorted({k.split("/")[0] for k in MHLA.files if k.startswith("attn_mask")}))
def test_mhla_attention_masked(favit, name):
    W = int(name.split("_W")[1])
    c = case(MHLA, name)
    _run(favit.models.mhla.MultiHeadLatentAttention(D, H, window_size=W), c,
         extra={"attention_mask": torch.from_numpy(c["attention_mask"])})


@pytest.mark.parametrize("name", sorted({k.split("/")[0] for k in MHLA.files if k.startswith("block_")}))
def test_mhla_block(favit, name):
    W = int(name.split("_W")[1])
    _run(favit.models.mhla.MHLATransformerBlock(D, H, window_size=W), case(MHLA, name))


def test_even_window_is_a_clear_error(favit):
    m = favit.models.mhla.MultiHeadLatentAttention(D, H, window_size=4).to(DEV)
    with pytest.raises(ValueError):
        m(torch.randn(1, 12, D, device=DEV))


@pytest.mark.parametrize("name", ["mha_L5", "mha_L17", "mha_L65"])
def test_dense_mha(favit, name):
    _run(favit.models.vit.MultiHeadAttention(D, H), case(VP, name))


def test_mlp(favit):
    _run(favit.models.vit.MLP(D, 4 * D, D), case(VP, "mlp"))


def test_vit_block(favit):
    _run(favit.models.vit.TransformerBlock(D, H), case(VP, "block_L17"))


def test_patch_embedding(favit):
    _run(favit.models.vit.PatchEmbedding(img_size=32, patch_size=4, in_channels=3, embed_dim=D), case(VP, "patch_embed"))


@pytest.mark.parametrize("use_mhla", [0, 1])
@pytest.mark.parametrize("L", [17, 65])
def test_vit_mhla_block(favit, use_mhla, L):
    _run(favit.models.vit_mhla.TransformerBlock(D, H, window_size=7, use_mhla=bool(use_mhla)),
         case(VP, f"vm_block_mhla{use_mhla}_L{L}"))


@pytest.mark.parametrize("masked", [0, 1])
@pytest.mark.parametrize("kind", ["ca", "mhca"])
def test_cross_attention(favit, kind, masked):
    c = case(CR, f"{kind}_mask{masked}")
    A = favit.models.attention
    m = A.CrossAttention(D) if kind == "ca" else A.MultiHeadCrossAttention(D, H)
    extra = {"attention_mask": torch.from_numpy(c["attention_mask"])} if masked else None
    _run(m, c, n_in=2, extra=extra)


@pytest.mark.parametrize("mh", [0, 1])
def test_cross_block(favit, mh):
    _run(favit.models.attention.CrossAttentionTransformerBlock(D, H, use_multi_head=bool(mh)), case(CR, f"block_mh{mh}"),
         n_in=2)


def test_bf16_mode_block_within_stated_tolerance(favit):
    favit.set_compute_dtype("bf16")
    _run(favit.models.mhla.MHLATransformerBlock(D, H, window_size=7), case(MHLA, "block_L17_W7"), fwd_tol=2e-2,
         grad_tol=4e-2)


@pytest.mark.parametrize("nm", ["grid", "vor16", "vor15"])
def test_sppp_model_logits(favit, nm):
    c = case(SP, nm)
    seg = torch.from_numpy(c["segmap"].astype(np.int64))
    segs = torch.stack([seg, torch.roll(seg, 5, dims=1)]).to(DEV)
    m = favit.models.sppp_mhla.SPPPViTMHLA(img_size=224, patch_size=16, num_classes=10, embed_dim=D, depth=2,
                                          num_heads=H, num_superpixels=16, pooling_type="mean", window_size=7,
                                          use_mhla=True)
    m.load_state_dict(sd_of(c))
    m.to(DEV).eval()
    m.segmentation.set_label_maps(segs)
    x = torch.from_numpy(c["model_x"].astype(np.float32)).to(DEV)
    y = m(x)
    assert rel_l2(y.detach().cpu(), c["logits"]) < 1e-3
    y.sum().backward()          # the whole SPPP graph is differentiable through the kernels
    assert m.patch_embed.projection[1].weight.grad.abs().sum().item() > 0


def test_token_bucketed_runs_batches_with_different_token_counts(favit):
    """models.sppp.TokenBucketed: a batch whose images segment into 4 and 3 superpixel tokens (S and S - 1 for
    num_superpixels = 4: the two counts the reference's positional encoding accepts, models/sppp.py:299).  The plain
    model raises where the reference fails in torch.stack (models/sppp_mhla.py:300); the wrapper's logits per image, and
    the parameter gradients of the batch's mean cross-entropy, equal those of the images run ONE BY ONE through the
    wrapped model (a batch of one always stacks: what the reference computes per image)."""
    torch.manual_seed(21)
    S = 64
    m = favit.models.sppp_mhla.SPPPViTMHLA(img_size=S, patch_size=8, num_classes=10, embed_dim=64, depth=2, num_heads=4,
                                          num_superpixels=4, pooling_type="mean", window_size=3, use_mhla=True).to(DEV).train()
    quad = (torch.arange(S)[:, None] >= S // 2).long() * 2 + (torch.arange(S)[None, :] >= S // 2).long()
    tri = torch.where(quad == 1, torch.zeros_like(quad), quad)                     # top half one region: labels 0, 2, 3
    maps = torch.stack([quad, tri, quad, tri, tri, quad]).contiguous().to(DEV)
    x = torch.randn(6, 3, S, S, device=DEV)
    y = torch.tensor([1, 4, 0, 9, 3, 3], device=DEV)
    m.segmentation.set_label_maps(maps)
    with pytest.raises(ValueError, match="different superpixel-token counts"):
        m(x)
    wrapped = favit.models.sppp.TokenBucketed(m)
    logits = wrapped(x)
    assert m.segmentation._maps is maps and m.assume_num_tokens is None          # restored
    favit.train.cross_entropy(logits, y).backward()
    got = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
    for p in m.parameters():
        p.grad = None
    rows = []
    for i in range(6):
        m.segmentation.set_label_maps(maps[i:i + 1].contiguous())
        li = m(x[i:i + 1])
        rows.append(li.detach())
        (favit.train.cross_entropy(li, y[i:i + 1]) / 6).backward()
    assert rel_l2(logits.detach(), torch.cat(rows)) < 1e-5
    for k, p in m.named_parameters():
        assert p.grad is not None and rel_l2(got[k], p.grad) < 2e-4, k
    # ... and what the ORACLE (the reference's algorithm, oracle/favit_oracle.py) gives for each image alone
    from oracle import favit_oracle as O
    sd = {k: v.detach().float().cpu() for k, v in m.state_dict().items()}
    with torch.no_grad():
        want = torch.cat([O.sppp_vit_mhla_forward(x[i:i + 1].cpu(), maps[i:i + 1].cpu().numpy(), sd, 8, 4, 3, True, S=4, kind="mean")
                          for i in range(6)])
    assert rel_l2(logits.detach().cpu(), want) < 1e-3
    # a batch with one count goes through in one piece
    m.segmentation.set_label_maps(maps[[0, 2, 5]].contiguous())
    assert rel_l2(wrapped(x[[0, 2, 5]]).detach(), torch.cat([rows[0], rows[2], rows[5]])) < 1e-5
    m.segmentation.set_label_maps(None)


def test_sppp_map_derived_tensors_follow_the_installed_maps(favit):
    """The patch -> superpixel mapping and the centroids depend on the label maps alone and are computed once per
    install (models/sppp.py::_MapState; the reference recomputes them in every forward, models/sppp_mhla.py:287-307).
    An in-place edit of the installed maps -- with update_label_maps or behind its back -- is followed: by the next eager
    forward, and by a captured step before its next replay.  Logits / losses equal those of a fresh install."""
    torch.manual_seed(23)
    S = 64
    mk = lambda: favit.models.sppp_mhla.SPPPViTMHLA(img_size=S, patch_size=8, num_classes=10, embed_dim=64, depth=2, num_heads=4,
                                                    num_superpixels=4, pooling_type="mean", window_size=3, use_mhla=True).to(DEV).train()
    m, ref = mk(), mk()
    ref.load_state_dict(m.state_dict())
    ar = torch.arange(S)
    quad = ((ar[:, None] >= S // 2).long() * 2 + (ar[None, :] >= S // 2).long())
    other = ((ar[:, None] >= S // 4).long() * 2 + (ar[None, :] >= 3 * S // 4).long())       # four regions, other borders
    mapsA = quad.expand(4, S, S).contiguous().to(DEV)
    mapsB = other.expand(4, S, S).contiguous().to(DEV)
    x = torch.randn(4, 3, S, S, device=DEV)
    y = torch.tensor([1, 4, 0, 9], device=DEV)
    installed = mapsA.clone()
    m.segmentation.set_label_maps(installed)
    ref.segmentation.set_label_maps(mapsA)
    assert rel_l2(m(x).detach(), ref(x).detach()) < 1e-6
    st = m.segmentation._state
    derived_before = [t.data_ptr() for t in st.get(8, 4)]
    installed.copy_(mapsB)                                           # behind update_label_maps' back
    ref.segmentation.set_label_maps(mapsB)
    assert rel_l2(m(x).detach(), ref(x).detach()) < 1e-6
    assert [t.data_ptr() for t in st.get(8, 4)] == derived_before     # refreshed in place
    m.segmentation.update_label_maps(mapsA)
    ref.segmentation.set_label_maps(mapsA)
    assert rel_l2(m(x).detach(), ref(x).detach()) < 1e-6
    # a captured step: its replays read the derived tensors at fixed addresses
    for mm in (m, ref):
        mm.assume_num_tokens = 4
    opt = favit.train.FusedAdamW(favit.train.param_groups(m, lr=0.0), lr=0.0, weight_decay=0.0, distributed=False)
    ropt = favit.train.FusedAdamW(favit.train.param_groups(ref, lr=0.0), lr=0.0, weight_decay=0.0, distributed=False)
    step = favit.train.GraphedStep(m, opt, x, y)
    for maps, how in ((mapsB, "copy"), (mapsA, "update"), (mapsB, "update")):
        if how == "copy":
            installed.copy_(maps)
        else:
            m.segmentation.update_label_maps(maps)
        ref.segmentation.set_label_maps(maps)
        got = step(x, y).item()
        want = favit.train.train_step(ref, x, y, ropt).item()
        assert abs(got - want) < 1e-5 * max(1.0, abs(want)), (how, got, want)
    # static_inputs: the caller's tensors ARE the captured buffers -- no clone, no copy per call; new batches are written
    # into them in place (bench.py's resident batch)
    xs, ys = x.clone(), y.clone()
    step2 = favit.train.GraphedStep(m, opt, xs, ys, static_inputs=True)
    assert step2.x is xs and step2.y is ys
    for k in range(2):
        xs.copy_(torch.randn_like(xs))
        ys.copy_(torch.randint(0, 10, ys.shape, device=DEV))
        got = step2(xs, ys).item()
        want = favit.train.train_step(ref, xs, ys, ropt).item()
        assert abs(got - want) < 1e-5 * max(1.0, abs(want)), (k, got, want)
    m.segmentation.set_label_maps(None)
    ref.segmentation.set_label_maps(None)
    favit.functional.clear_lp_mirrors()


def test_sppp_reference_dict_api(favit):
    c = case(SP, "vor16")
    seg = torch.from_numpy(c["segmap"].astype(np.int64)).to(DEV)
    S = favit.models.sppp
    mapping = S.PatchToSuperpixelMapper(16).map_patches(seg, 224)
    assert list(mapping.keys()) == c["map_keys"].tolist()
    emb = torch.from_numpy(SP["emb"]).to(DEV)
    for kind in ("mean", "max", "attention"):
        out = S.SuperpixelPooling(kind).pool(emb, mapping)
        assert rel_l2(out.cpu(), c[f"pool_{kind}"]) < 2e-5
    with pytest.raises(ValueError):
        S.SuperpixelPooling("median").pool(emb, mapping)


def _model_check(favit, model, x, y, key, tol_logits, tol_loss, tol_gn):
    model.to(DEV).eval()
    logits = model(x.to(DEV))
    ref = MD[f"{key}/logits"]
    assert rel_l2(logits.detach().cpu(), ref) < tol_logits, rel_l2(logits.detach().cpu(), ref)
    loss = favit.train.cross_entropy(logits, y.to(DEV))
    assert abs(loss.item() - float(MD[f"{key}/loss"])) < tol_loss * abs(float(MD[f"{key}/loss"]))
    loss.backward()
    worst = 0.0
    for k, p in model.named_parameters():
        gn = p.grad.norm().item()
        r = float(MD[f"{key}/gnorm/{k}"])
        worst = max(worst, abs(gn - r) / max(r, 1e-12))
    assert worst < tol_gn, worst


def test_cfg1_vit_tiny_logits_loss_gradnorms(favit):
    """BASELINE.json configs[0]: ViT-Tiny 32x32 patch4, seed-initialised (a24: same RNG order)."""
    torch.manual_seed(1234)
    m = favit.models.vit.VisionTransformer(img_size=32, patch_size=4, num_classes=10, embed_dim=192, depth=12,
                                           num_heads=3)
    _model_check(favit, m, torch.from_numpy(MD["cfg1/x"]), torch.from_numpy(MD["cfg1/y"]), "cfg1", 1e-3, 1e-4, 2e-3)


@pytest.mark.parametrize("mode,tl,tg", [("fp32", 1e-3, 2e-3), ("bf16", 2e-2, 5e-2)])
def test_cfg2_vit_mhla_small_logits_loss_gradnorms(favit, mode, tl, tg):
    """BASELINE.json configs[1]: ViT-MHLA-Small 224/p16, 197 tokens; weights and inputs are
    regenerated from seed 1234 exactly as tests/golden/make_golden.py did."""
    torch.manual_seed(1234)
    m = favit.models.vit_mhla.VisionTransformerMHLA(img_size=224, patch_size=16, num_classes=1000, embed_dim=384,
                                                    depth=12, num_heads=6, window_size=7, use_mhla=True)
    x = torch.randn(2, 3, 224, 224)
    y = torch.randint(0, 1000, (2,))
    assert abs(x.double().sum().item() - float(MD["cfg2/x_sum"])) < 1e-6 and torch.equal(y, torch.from_numpy(MD["cfg2/y"]))
    favit.set_compute_dtype(mode)
    _model_check(favit, m, x, y, "cfg2", tl, tl, tg)


def test_use_mhla_false_fallback_model(favit):
    torch.manual_seed(1234)
    m = favit.models.vit_mhla.VisionTransformerMHLA(img_size=32, patch_size=4, num_classes=10, embed_dim=64, depth=2,
                                                    num_heads=4, use_mhla=False)
    m.to(DEV).eval()
    y = m(torch.from_numpy(MD["fallback/x"]).to(DEV))
    assert rel_l2(y.detach().cpu(), MD["fallback/logits"]) < 1e-3


def test_train_mode_dropout_runs_and_is_stochastic(favit):
    torch.manual_seed(0)
    m = favit.models.vit_mhla.VisionTransformerMHLA(img_size=32, patch_size=4, num_classes=10, embed_dim=64, depth=2,
                                                    num_heads=4, dropout=0.1, attn_dropout=0.1, embed_dropout=0.1,
                                                    use_mhla=True).to(DEV)
    x = torch.randn(4, 3, 32, 32, device=DEV)
    m.train()
    a, b = m(x), m(x)
    assert not torch.allclose(a, b)
    a.sum().backward()
    assert all(torch.isfinite(p.grad).all() for p in m.parameters())
    m.eval()
    assert torch.equal(m(x), m(x))


@pytest.mark.parametrize("use_mhla", [True, False])
def test_train_mode_dropout_gradients_match_finite_differences(favit, use_mhla):
    """Train-mode dropout has no reference fixture (the masks come from this library's own counter-based generator),
    but the seeds are drawn from torch's CPU generator: under a fixed torch.manual_seed the dropped model is a
    deterministic function of its parameters, so the analytic gradients of the whole chain (embedding / attention /
    projection / MLP dropouts recomputed in backward, the mask fused into the LayerNorm backward's low-precision copy)
    must match central finite differences along random directions.  fp32 mode."""
    favit.set_compute_dtype("fp32")
    torch.manual_seed(11)
    m = favit.models.vit_mhla.VisionTransformerMHLA(img_size=32, patch_size=4, num_classes=10, embed_dim=64, depth=3,
                                                    num_heads=4, dropout=0.2, attn_dropout=0.2, embed_dropout=0.2,
                                                    window_size=5, use_mhla=use_mhla).to(DEV).train()
    x = torch.randn(6, 3, 32, 32, device=DEV)
    y = torch.randint(0, 10, (6,), device=DEV)

    def loss():
        torch.manual_seed(1234)                      # same dropout seeds on every evaluation
        return torch.nn.functional.cross_entropy(m(x).double(), y)

    l0 = loss()
    l0.backward()
    assert float(loss().detach()) == float(l0.detach()), "a fixed torch seed must give the same masks"
    params = [p for p in m.parameters()]
    grads = [p.grad.detach().clone().double() for p in params]
    gen = torch.Generator(device=DEV).manual_seed(5)
    for trial in range(4):
        dirs = [torch.randn(p.shape, device=DEV, generator=gen) for p in params]
        if trial == 3:                               # one direction confined to the last block (shallow path)
            names = [n for n, _ in m.named_parameters()]
            dirs = [d if ".2." in n else torch.zeros_like(d) for n, d in zip(names, dirs)]
        analytic = sum(float((g * d.double()).sum()) for g, d in zip(grads, dirs))
        eps = 1e-4                                   # (2e-3 is already 20 % off on this loss surface)
        with torch.no_grad():
            for p, d in zip(params, dirs): p.add_(d, alpha=eps)
            lp = float(loss())
            for p, d in zip(params, dirs): p.add_(d, alpha=-2 * eps)
            lm = float(loss())
            for p, d in zip(params, dirs): p.add_(d, alpha=eps)
        numeric = (lp - lm) / (2 * eps)
        assert abs(numeric - analytic) <= 5e-3 * max(abs(analytic), abs(numeric)) + 2e-3, (trial, numeric, analytic)


@pytest.mark.parametrize("mode,tol", [("fp32", 1e-5), ("bf16", 2e-2)])
def test_full_size_cfg2_batch_independence(favit, mode, tol):
    """BASELINE.json configs[1] at its full size (B=256): every image is independent, so the
    logits of images 0..1 inside the batch of 256 equal the logits of the batch of 2 (bf16: up to
    rounding flips, stated tolerance 2e-2)."""
    favit.set_compute_dtype(mode)
    torch.manual_seed(1234)
    m = favit.models.vit_mhla.VisionTransformerMHLA(img_size=224, patch_size=16, num_classes=1000, embed_dim=384,
                                                    depth=12, num_heads=6, window_size=7, use_mhla=True).to(DEV).eval()
    x = torch.randn(256, 3, 224, 224, device=DEV)
    with torch.no_grad():
        big = m(x)
        small = m(x[:2].contiguous())
    assert torch.isfinite(big).all()
    assert rel_l2(big[:2].cpu(), small.cpu()) < tol


def test_direct_gradient_accumulation_matches_returned_grads(favit):
    """With flat .grad buffers attached (FusedAdamW) the kernels accumulate parameter gradients in
    place and autograd sees None; results must equal the ordinary returned-gradient path, and a
    second backward must accumulate (torch semantics)."""
    torch.manual_seed(0)
    m = favit.models.vit_mhla.VisionTransformerMHLA(img_size=32, patch_size=4, num_classes=10, embed_dim=64, depth=2,
                                                    num_heads=4, window_size=7, use_mhla=True).to(DEV)
    x = torch.randn(4, 3, 32, 32, device=DEV)
    y = torch.randint(0, 10, (4,), device=DEV)
    favit.train.cross_entropy(m(x), y).backward()
    ref = {k: p.grad.clone() for k, p in m.named_parameters()}
    for p in m.parameters():
        p.grad = None
    opt = favit.train.FusedAdamW(favit.train.param_groups(m, lr=1e-4), distributed=False)
    opt.zero_grad()
    favit.train.cross_entropy(m(x), y).backward()
    for k, p in m.named_parameters():
        assert rel_l2(p.grad, ref[k]) < 1e-5, k
    favit.train.cross_entropy(m(x), y).backward()
    for k, p in m.named_parameters():
        assert rel_l2(p.grad, 2 * ref[k]) < 1e-5, k
    w0 = m.head.weight.detach().clone()
    opt.step()
    assert not torch.equal(w0, m.head.weight)


def test_bf16_weight_cache_does_not_leak_between_models(favit):
    """Two models built one after the other usually get the SAME parameter addresses from the caching
    allocator; the compute-dtype weight copies are cached per parameter object, so the second model must
    never be served the first one's bf16 weights (regression: the cache used to be keyed by address)."""
    import gc
    x = torch.randn(2, 3, 32, 32, device=DEV)

    def build(seed):
        torch.manual_seed(seed)
        return favit.models.vit_mhla.VisionTransformerMHLA(img_size=32, patch_size=4, num_classes=10, embed_dim=64,
                                                           depth=2, num_heads=4, use_mhla=True).to(DEV).eval()
    favit.set_compute_dtype("bf16")
    a = build(1)
    with torch.no_grad():
        ya = a(x)
    del a
    gc.collect()
    b = build(2)
    with torch.no_grad():
        yb = b(x)
        favit.set_compute_dtype("fp32")
        yb32 = b(x)
    assert rel_l2(yb.float().cpu(), yb32.cpu()) < 2e-2
    assert rel_l2(ya.float().cpu(), yb32.cpu()) > 0.1          # the two models really differ
    # parameters edited through .data are invisible to the version counter: explicit invalidation
    favit.set_compute_dtype("bf16")
    with torch.no_grad():
        b.head.weight.data.mul_(2.0)
        favit.invalidate_weight_cache()
        y2 = b(x)
    assert rel_l2(y2.float().cpu(), 2 * (yb32.cpu() - b.head.bias.detach().cpu()) + b.head.bias.detach().cpu()) < 2e-2


def test_frozen_layers_flow_of_the_experiments(favit):
    """experiments/mhla_pretrained.py:237-247 freezes everything except 'head' / 'latent_proj' by name:
    frozen parameters get no gradient, trainable ones match the oracle."""
    from oracle import favit_oracle as O
    favit.set_compute_dtype("fp32")
    torch.manual_seed(3)
    m = favit.models.vit_mhla.VisionTransformerMHLA(img_size=32, patch_size=4, num_classes=10, embed_dim=64, depth=2,
                                                    num_heads=4, use_mhla=True).to(DEV).train()
    for n, p in m.named_parameters():
        p.requires_grad = ("head" in n) or ("latent_proj" in n)
    x = torch.randn(3, 3, 32, 32, device=DEV)
    y = torch.randint(0, 10, (3,), device=DEV)
    loss = favit.train.cross_entropy(m(x), y)
    loss.backward()
    sd = {k: v.detach().cpu().clone().requires_grad_(("head" in k) or ("latent_proj" in k)) for k, v in m.state_dict().items()}
    lo = O.cross_entropy(O.vit_mhla_forward(x.cpu(), sd, 4, 4, 7, True), y.cpu())
    lo.backward()
    assert abs(loss.item() - lo.item()) < 1e-5
    for n, p in m.named_parameters():
        if p.requires_grad:
            assert rel_l2(p.grad.cpu(), sd[n].grad) < 1e-4, n
        else:
            assert p.grad is None, n
    # the fused optimizer only sees the trainable groups (5x lr on latent_proj, head lr)
    groups = favit.train.param_groups(m, lr=1e-3, head_lr=1e-2)
    assert sorted(len(g["params"]) for g in groups) == [2, 4]


@pytest.mark.parametrize("mode,tol", [("fp32", 1e-4), ("bf16", 4e-2)])
def test_frozen_layers_with_the_fused_optimizer(favit, mode, tol):
    """The fine-tuning flow as the benchmark's cfg5 runs it: frozen blocks, trainable head / latent_proj, gradients
    written straight into the fused optimizer's flat buffers.  This is the path on which the batched fold backward runs
    with its qkv half switched off, frozen LayerNorms skip their parameter-gradient fold and a block's only
    weight-gradient problem (the folded qkv weight) takes the slab launch -- gradients against the oracle."""
    from oracle import favit_oracle as O
    favit.set_compute_dtype(mode)
    try:
        torch.manual_seed(9)
        m = favit.models.vit_mhla.VisionTransformerMHLA(img_size=32, patch_size=4, num_classes=10, embed_dim=128, depth=3,
                                                        num_heads=2, use_mhla=True).to(DEV).train()
        for n, p in m.named_parameters():
            p.requires_grad = ("head" in n) or ("latent_proj" in n)
        opt = favit.train.FusedAdamW(favit.train.param_groups(m, lr=1e-3, head_lr=1e-2), lr=1e-3, distributed=False)
        x = torch.randn(32, 3, 32, 32, device=DEV)         # 32 x 65 = 2,080 tokens: a multiple of 32 -> the slab launch
        y = torch.randint(0, 10, (32,), device=DEV)
        sd = {k: v.detach().cpu().clone().requires_grad_(("head" in k) or ("latent_proj" in k)) for k, v in m.state_dict().items()}
        lo = O.cross_entropy(O.vit_mhla_forward(x.cpu(), sd, 4, 2, 7, True), y.cpu())
        lo.backward()
        for _ in range(2):                                 # twice: zero_grad must reset what the first pass accumulated
            opt.zero_grad()
            loss = favit.train.cross_entropy(m(x), y)
            loss.backward()
        assert abs(loss.item() - lo.item()) < tol * max(1.0, abs(lo.item()))
        for n, p in m.named_parameters():
            if p.requires_grad:
                gbuf = favit.functional._gt(p)
                assert gbuf is not None, n
                assert rel_l2(gbuf.cpu(), sd[n].grad) < (1e-4 if mode == "fp32" else 5e-2), n
            else:
                assert p.grad is None and favit.functional._gt(p) is None, n
    finally:
        favit.set_compute_dtype("fp32")
        favit.functional.clear_lp_mirrors()
        favit.functional.set_grad_ready_hook(None)


@pytest.mark.parametrize("mode,tol", [("fp32", 1e-4), ("bf16", 3e-2)])
def test_torch_optim_adamw_training_steps(favit, mode, tol):
    """The reference's own loop (torch.optim.AdamW, experiments/mhla_pretrained.py:308-372) over three
    steps: the cached compute-dtype weight copies must follow the optimizer's in-place updates."""
    from oracle import favit_oracle as O
    favit.set_compute_dtype(mode)
    torch.manual_seed(5)
    m = favit.models.vit_mhla.VisionTransformerMHLA(img_size=32, patch_size=4, num_classes=10, embed_dim=64, depth=2,
                                                    num_heads=4, use_mhla=True).to(DEV).train()
    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    names = [k for k, _ in m.named_parameters()]
    opt = torch.optim.AdamW(m.parameters(), lr=1e-2, weight_decay=0.05)
    ropt = torch.optim.AdamW([sd[k] for k in names], lr=1e-2, weight_decay=0.05)
    x = torch.randn(4, 3, 32, 32, device=DEV)
    y = torch.randint(0, 10, (4,), device=DEV)
    for step in range(3):
        opt.zero_grad()
        loss = torch.nn.CrossEntropyLoss()(m(x), y)
        loss.backward()
        opt.step()
        ropt.zero_grad()
        lo = O.cross_entropy(O.vit_mhla_forward(x.cpu(), sd, 4, 4, 7, True), y.cpu())
        lo.backward()
        ropt.step()
        assert abs(loss.item() - lo.item()) < tol * max(1.0, abs(lo.item())), (step, loss.item(), lo.item())
    assert lo.item() < 2.0        # three lr=1e-2 steps on one batch visibly reduce the loss (starts at ~2.3)


def test_cfg2_fp8_mode_within_stated_tolerance(favit):
    """fp8 mode at BASELINE.json configs[1] shapes (B=2): same stated tolerance as the cfg4 fp8 case
    (tests/test_configs_golden.py: logits 0.15, loss 2e-2, gradient norms 25 %)."""
    torch.manual_seed(1234)
    m = favit.models.vit_mhla.VisionTransformerMHLA(img_size=224, patch_size=16, num_classes=1000, embed_dim=384,
                                                    depth=12, num_heads=6, window_size=7, use_mhla=True)
    x = torch.randn(2, 3, 224, 224)
    y = torch.randint(0, 1000, (2,))
    favit.set_compute_dtype("fp8")
    try:
        _model_check(favit, m, x, y, "cfg2", 0.15, 2e-2, 0.25)
    finally:
        favit.set_compute_dtype("fp32")


def test_graphed_step_matches_eager_steps(favit):
    """train.GraphedStep (forward + backward replayed from a captured HIP graph, optimizer eager) walks the
    same trajectory as eager train_step calls: same losses, same weights."""
    favit.set_compute_dtype("bf16")

    def build():
        torch.manual_seed(11)
        m = favit.models.vit_mhla.VisionTransformerMHLA(img_size=32, patch_size=4, num_classes=10, embed_dim=64,
                                                        depth=2, num_heads=4, use_mhla=True).to(DEV).train()
        opt = favit.train.FusedAdamW(favit.train.param_groups(m, lr=1e-3), lr=1e-3, weight_decay=0.05, distributed=False)
        return m, opt
    g = torch.Generator(device=DEV).manual_seed(2)
    xs = [torch.randn(8, 3, 32, 32, device=DEV, generator=g) for _ in range(3)]
    ys = [torch.randint(0, 10, (8,), device=DEV, generator=g) for _ in range(3)]
    m1, o1 = build()
    eager = [favit.train.train_step(m1, x, y, o1).item() for x, y in zip(xs, ys)]
    m2, o2 = build()
    step = favit.train.GraphedStep(m2, o2, xs[0], ys[0])
    graphed = [step(x, y).item() for x, y in zip(xs, ys)]
    # deterministic kernels on this path (no split-K at these sizes would differ): tiny tolerance for atomics
    for a, b in zip(eager, graphed):
        assert abs(a - b) < 2e-3 * max(1.0, abs(a)), (eager, graphed)
    w1 = torch.cat([p.detach().flatten() for p in m1.parameters()])
    w2 = torch.cat([p.detach().flatten() for p in m2.parameters()])
    assert rel_l2(w2.cpu(), w1.cpu()) < 1e-3


@pytest.mark.parametrize("kind", ["vit", "vit_mhla"])
def test_graphed_step_per_parameter_and_optimizer_state_at_bench_like_batch(favit, kind):
    """As above at a batch that takes the chunked embed-prologue backward (B >= 32: zero-fill + atomics for the
    cls_token / pos_embed gradients), compared PER PARAMETER and in the optimizer's moments after four replays.  The
    concatenated-weights comparison above cannot see one small parameter going wrong: with the zero-fill as a captured
    hipMemsetAsync the pos_embed / cls_token gradients of every replay after the first carried a huge constant in every
    fourth element, AdamW's second moment overflowed to inf and both parameters silently stopped training."""
    favit.set_compute_dtype("bf16")

    def build():
        torch.manual_seed(12)
        if kind == "vit":
            m = favit.models.vit.VisionTransformer(img_size=32, patch_size=4, num_classes=10, embed_dim=64, depth=2, num_heads=2)
        else:
            m = favit.models.vit_mhla.VisionTransformerMHLA(img_size=32, patch_size=4, num_classes=10, embed_dim=64,
                                                            depth=2, num_heads=4, use_mhla=True)
        m = m.to(DEV).train()
        opt = favit.train.FusedAdamW(favit.train.param_groups(m, lr=1e-3), lr=1e-3, weight_decay=0.05, distributed=False)
        return m, opt
    g = torch.Generator(device=DEV).manual_seed(3)
    xs = [torch.randn(64, 3, 32, 32, device=DEV, generator=g) for _ in range(4)]
    ys = [torch.randint(0, 10, (64,), device=DEV, generator=g) for _ in range(4)]
    m1, o1 = build()
    for x, y in zip(xs, ys):
        favit.train.train_step(m1, x, y, o1)
    m2, o2 = build()
    step = favit.train.GraphedStep(m2, o2, xs[0], ys[0])
    for x, y in zip(xs, ys):
        step(x, y)
    torch.cuda.synchronize()
    for (n, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        assert bool(torch.isfinite(p2).all()) and bool(torch.isfinite(p2.grad).all()), n
        assert float(p2.grad.abs().max()) < 1e3, n
        if n.endswith("qkv.bias"):
            # the key third of this bias has a mathematically zero gradient (softmax is invariant to a per-query constant):
            # what arrives is rounding noise, AdamW normalises it to +-lr steps, and eager and replayed runs (fp32 atomics
            # in another order) walk different random paths there
            continue
        assert rel_l2(p2, p1) < 2e-3, (n, rel_l2(p2, p1))
        assert rel_l2(p2.grad, p1.grad) < 2e-2, (n, rel_l2(p2.grad, p1.grad))       # the last step's gradients
    for g1, g2 in zip(o1.groups, o2.groups):
        for key in ("m", "v"):
            assert bool(torch.isfinite(g2[key]).all()), key
            assert rel_l2(g2[key], g1[key]) < 2e-2, (key, rel_l2(g2[key], g1[key]))
        assert float(g2["v"].max()) < 1e3


def test_graphed_step_with_dropout_matches_eager_steps(favit, monkeypatch):
    """The reference's training setting (dropout = attn_dropout = embed_dropout = 0.1, main.py:106) through
    train.GraphedStep: dropout seeds are frozen into the captured kernels, a device epoch word (bumped by the graph's
    first node) is mixed in at execution time.  For the SAME by-value seeds and the SAME epoch values the eager path
    draws the same masks: the two trajectories agree, and consecutive replays differ from each other (fresh masks)."""
    favit.set_compute_dtype("bf16")
    F = favit.functional
    counter = {"n": 0}

    def fake_seed():
        counter["n"] += 1
        return 0x5DEECE66D * counter["n"] + 11

    monkeypatch.setattr(F, "_seed", fake_seed)

    def build():
        torch.manual_seed(11)
        m = favit.models.vit_mhla.VisionTransformerMHLA(img_size=32, patch_size=4, num_classes=10, embed_dim=64, depth=2,
                                                        num_heads=4, use_mhla=True, dropout=0.1, attn_dropout=0.1,
                                                        embed_dropout=0.1).to(DEV).train()
        opt = favit.train.FusedAdamW(favit.train.param_groups(m, lr=1e-3), lr=1e-3, weight_decay=0.05, distributed=False)
        return m, opt
    g = torch.Generator(device=DEV).manual_seed(2)
    xs = [torch.randn(8, 3, 32, 32, device=DEV, generator=g) for _ in range(3)]
    ys = [torch.randint(0, 10, (8,), device=DEV, generator=g) for _ in range(3)]
    try:
        # graphed: one warm-up pass (seeds 1..n, epoch -> 1), the capture draws seeds n+1..2n, replays run at epoch 2, 3, 4
        m2, o2 = build()
        counter["n"] = 0
        step = favit.train.GraphedStep(m2, o2, xs[0], ys[0], warmup=1)
        assert step.epoch is not None and F.get_dropout_epoch() is step.epoch
        n = counter["n"] // 2
        assert n >= 9 and counter["n"] == 2 * n           # embed + 2 x (attention, proj, 2 x MLP) sites
        graphed = [step(x, y).item() for x, y in zip(xs, ys)]
        assert int(step.epoch.item()) == 4
        # eager with the capture's seeds and the replays' epoch values
        m1, o1 = build()
        eager = []
        for k, (x, y) in enumerate(zip(xs, ys)):
            counter["n"] = n
            step.epoch.fill_(k + 2)
            eager.append(favit.train.train_step(m1, x, y, o1).item())
        for a, b in zip(eager, graphed):
            assert abs(a - b) < 2e-3 * max(1.0, abs(a)), (eager, graphed)
        w1 = torch.cat([p.detach().flatten() for p in m1.parameters()])
        w2 = torch.cat([p.detach().flatten() for p in m2.parameters()])
        assert rel_l2(w2.cpu(), w1.cpu()) < 1e-3
    finally:
        F.set_dropout_epoch(None)
        favit.set_compute_dtype("fp32")
        favit.functional.clear_lp_mirrors()


def test_graphed_step_backward_segments_match_the_single_graph(favit):
    """segments = 3: the backward captured as three graphs cut at detached boundaries (the form that lets GradSync put
    finished buckets on the wire between replays) walks the same trajectory as the single-graph and the eager step."""
    favit.set_compute_dtype("bf16")

    def build():
        torch.manual_seed(13)
        m = favit.models.vit_mhla.VisionTransformerMHLA(img_size=32, patch_size=4, num_classes=10, embed_dim=64, depth=6,
                                                        num_heads=4, use_mhla=True).to(DEV).train()
        opt = favit.train.FusedAdamW(favit.train.param_groups(m, lr=1e-3), lr=1e-3, weight_decay=0.05, distributed=False)
        return m, opt
    g = torch.Generator(device=DEV).manual_seed(4)
    xs = [torch.randn(8, 3, 32, 32, device=DEV, generator=g) for _ in range(3)]
    ys = [torch.randint(0, 10, (8,), device=DEV, generator=g) for _ in range(3)]
    try:
        m1, o1 = build()
        eager = [favit.train.train_step(m1, x, y, o1).item() for x, y in zip(xs, ys)]
        m3, o3 = build()
        step = favit.train.GraphedStep(m3, o3, xs[0], ys[0], segments=3)
        assert len(step.graphs) == 4 and len(step._ready) == 3
        names = {id(p): k for k, p in m3.named_parameters()}
        first = {names[id(p)] for p in step._ready[0]}
        last = {names[id(p)] for p in step._ready[-1]}
        assert "head.weight" in first and "blocks.5.mlp.fc2.weight" in first and "blocks.0.attn.qkv.weight" not in first
        assert "blocks.0.attn.qkv.weight" in last and "cls_token" in last and "patch_embed.projection.1.weight" in last
        assert sum(len(r) for r in step._ready) == len(names)
        seg = [step(x, y).item() for x, y in zip(xs, ys)]
        for a, b in zip(eager, seg):
            assert abs(a - b) < 2e-3 * max(1.0, abs(a)), (eager, seg)
        w1 = torch.cat([p.detach().flatten() for p in m1.parameters()])
        w3 = torch.cat([p.detach().flatten() for p in m3.parameters()])
        assert rel_l2(w3.cpu(), w1.cpu()) < 1e-3
    finally:
        favit.set_compute_dtype("fp32")
        favit.functional.clear_lp_mirrors()


@pytest.mark.parametrize("mode", ["bf16", "fp8"])
def test_fused_loop_overfits_a_small_batch(favit, mode):
    """End-to-end sanity of the hot loop (zero_grad -> forward -> CE -> backward -> fused AdamW): a small
    ViT-MHLA memorises 16 random images; with the name-based groups of the experiments (5x lr on latent_proj)."""
    favit.set_compute_dtype(mode)
    try:
        torch.manual_seed(21)
        m = favit.models.vit_mhla.VisionTransformerMHLA(img_size=32, patch_size=4, num_classes=10, embed_dim=128, depth=2,
                                                        num_heads=2, use_mhla=True).to(DEV).train()
        opt = favit.train.FusedAdamW(favit.train.param_groups(m, lr=2e-3), lr=2e-3, weight_decay=0.0, distributed=False)
        x = torch.randn(16, 3, 32, 32, device=DEV)
        y = torch.arange(16, device=DEV) % 10
        first = None
        for _ in range(60):
            loss = favit.train.train_step(m, x, y, opt)
            first = loss.item() if first is None else first
        assert first > 2.0 and loss.item() < 0.3, (first, loss.item())
        m.eval()
        with torch.no_grad():
            assert (m(x).argmax(1) == y).float().mean().item() > 0.9
    finally:
        favit.set_compute_dtype("fp32")
