"""Whole-model wrappers: PretrainedViTWithMHLA / PretrainedSPPPViTWithMHLA (models/mhla_models.py:22-395),
SPPPViT (models/sppp.py:303-520), CrossAttentionViT / CrossAttentionSPPPViT (models/attention.py:222-609).

The first two are constructible in the reference with an odd window_size: tests/golden/wrappers.npz holds what the
REFERENCE produced (tests/golden/make_golden.py::gen_wrappers) -- CPU tests pin the oracle and the mirrors' weight
init to it, GPU tests the HIP path.  The other three raise in the reference's own constructors (sppp.py:378 names
a class that does not exist; attention.py:275,454 use nn.Transpose): no reference output can exist, so the HIP
path is checked against the oracle's composition of individually pinned pieces (forward, loss and every
parameter gradient) -- "parity unpinned" as whole models, stated here."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_l2
from oracle import favit_oracle as O

WR = load_golden("wrappers.npz")
DEV = "cuda"


def _pvit(favit):
    torch.manual_seed(77)
    return favit.models.mhla_models.PretrainedViTWithMHLA(img_size=32, patch_size=4, num_classes=10, embed_dim=64, depth=2,
                                                          num_heads=4, window_size=7).eval()


def _psppp(favit, kind):
    torch.manual_seed(78)
    return favit.models.mhla_models.PretrainedSPPPViTWithMHLA(img_size=64, patch_size=16, num_classes=10, embed_dim=64,
                                                              depth=2, num_heads=4, window_size=3, num_superpixels=4,
                                                              pooling_type=kind).eval()


def _sd(m):
    return {k: v.detach().cpu().clone().requires_grad_(v.is_floating_point()) for k, v in m.state_dict().items()}


def _gn_worst(named_grads, key):
    worst = 0.0
    for k, g in named_grads:
        r = float(WR[f"{key}/gnorm/{k}"])
        gn = 0.0 if g is None else g.norm().item()
        worst = max(worst, abs(gn - r) / max(r, 1e-10))
    return worst


# ------------------------------------------------------------------ CPU: mirrors' init + the oracle vs the reference
def test_pretrained_vit_mhla_mirror_and_oracle_match_reference(favit):
    m = _pvit(favit)
    assert abs(sum(p.double().sum().item() for p in m.parameters()) - float(WR["pvit/param_sum"])) < 1e-6
    assert m.get_num_parameters() == int(WR["pvit/n_params"])
    assert list(m.state_dict().keys()) == [str(k) for k in WR["pvit/sd_keys"]]
    sd = _sd(m)
    x, y = torch.from_numpy(WR["pvit/x"]), torch.from_numpy(WR["pvit/y"])
    logits = O.pretrained_vit_mhla_forward(x, sd, 4, 4, 7)
    assert rel_l2(logits, WR["pvit/logits"]) < 2e-5
    loss = O.cross_entropy(logits, y)
    assert abs(loss.item() - float(WR["pvit/loss"])) < 1e-5
    loss.backward()
    assert _gn_worst(((k, sd[k].grad) for k, _ in m.named_parameters()), "pvit") < 1e-3


@pytest.mark.parametrize("kind", ["mean", "max", "attention"])
def test_pretrained_sppp_vit_mhla_mirror_and_oracle_match_reference(favit, kind):
    m = _psppp(favit, kind)
    key = f"psppp_{kind}"
    assert abs(sum(p.double().sum().item() for p in m.parameters()) - float(WR[f"{key}/param_sum"])) < 1e-6
    sd = _sd(m)
    x, y = torch.from_numpy(WR[f"{key}/x"]), torch.from_numpy(WR[f"{key}/y"])
    segs = WR["psppp/segmaps"].astype(np.int64)
    logits = O.pretrained_sppp_vit_mhla_forward(x, segs, sd, 16, 4, 3, S=4, kind=kind)
    assert rel_l2(logits, WR[f"{key}/logits"]) < 2e-5
    loss = O.cross_entropy(logits, y)
    loss.backward()
    assert _gn_worst(((k, sd[k].grad) for k, _ in m.named_parameters()), key) < 1e-3


@pytest.mark.gpu
def test_even_window_raises_instead_of_crashing(favit):
    """The reference's default window_size = 4 crashes in torch.stack at the first forward (mhla.py:83)."""
    m = favit.models.mhla_models.PretrainedViTWithMHLA(img_size=32, patch_size=4, num_classes=10, embed_dim=64, depth=1,
                                                       num_heads=4)          # window_size = 4, the reference default
    with pytest.raises(ValueError):
        m.to(DEV)(torch.randn(1, 3, 32, 32, device=DEV))


# ------------------------------------------------------------------ GPU: the HIP path vs the reference's outputs
@pytest.mark.gpu
@pytest.mark.parametrize("mode,tl,tg", [("fp32", 1e-3, 2e-3), ("bf16", 2e-2, 5e-2)])
def test_gpu_pretrained_vit_mhla(favit, mode, tl, tg):
    favit.set_compute_dtype(mode)
    try:
        m = _pvit(favit).to(DEV)
        x, y = torch.from_numpy(WR["pvit/x"]).to(DEV), torch.from_numpy(WR["pvit/y"]).to(DEV)
        logits = m(x)
        assert rel_l2(logits.detach().float().cpu(), WR["pvit/logits"]) < tl
        loss = favit.train.cross_entropy(logits, y)
        assert abs(loss.item() - float(WR["pvit/loss"])) < tl * abs(float(WR["pvit/loss"]))
        loss.backward()
        assert _gn_worst(((k, p.grad) for k, p in m.named_parameters()), "pvit") < tg
    finally:
        favit.set_compute_dtype("fp32")


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["mean", "max", "attention"])
@pytest.mark.parametrize("mode,tl,tg", [("fp32", 1e-3, 2e-3), ("bf16", 2e-2, 6e-2)])
def test_gpu_pretrained_sppp_vit_mhla(favit, kind, mode, tl, tg):
    favit.set_compute_dtype(mode)
    try:
        key = f"psppp_{kind}"
        m = _psppp(favit, kind).to(DEV)
        m.segmentation.set_label_maps(torch.from_numpy(WR["psppp/segmaps"].astype(np.int64)).to(DEV))
        x, y = torch.from_numpy(WR[f"{key}/x"]).to(DEV), torch.from_numpy(WR[f"{key}/y"]).to(DEV)
        logits = m(x)
        assert rel_l2(logits.detach().float().cpu(), WR[f"{key}/logits"]) < tl
        loss = favit.train.cross_entropy(logits, y)
        loss.backward()
        assert _gn_worst(((k, p.grad) for k, p in m.named_parameters()), key) < tg
    finally:
        favit.set_compute_dtype("fp32")


# ------------------------------------------------------------------ GPU: the three classes the reference cannot build
def _vs_oracle(favit, m, x, y, ref_fn, tol_logits=1e-3, tol_grad=2e-3):
    sd = _sd(m)
    ref_logits = ref_fn(sd)
    ref_loss = O.cross_entropy(ref_logits, y)
    ref_loss.backward()
    m.to(DEV)
    logits = m(x.to(DEV))
    assert torch.isfinite(logits).all()
    assert rel_l2(logits.detach().cpu(), ref_logits.detach()) < tol_logits
    loss = favit.train.cross_entropy(logits, y.to(DEV))
    assert abs(loss.item() - ref_loss.item()) < tol_logits * max(1.0, abs(ref_loss.item()))
    loss.backward()
    for k, p in m.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
        r = sd[k].grad
        assert (p.grad.cpu() - r).norm() <= tol_grad * r.norm() + 1e-6, k


def _small_segs():
    return WR["psppp/segmaps"].astype(np.int64)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["mean", "attention"])
def test_gpu_sppp_vit_dense_blocks_vs_oracle_composition(favit, kind):
    favit.set_compute_dtype("fp32")
    torch.manual_seed(5)
    m = favit.models.sppp.SPPPViT(img_size=64, patch_size=16, num_classes=10, embed_dim=64, depth=2, num_heads=4,
                                  num_superpixels=4, pooling_type=kind).eval()
    segs = _small_segs()
    x, y = torch.randn(2, 3, 64, 64), torch.randint(0, 10, (2,))
    m.segmentation.set_label_maps(torch.from_numpy(segs).to(DEV))
    _vs_oracle(favit, m, x, y, lambda sd: O.sppp_vit_forward(x, segs, sd, 16, 4, S=4, kind=kind))


@pytest.mark.gpu
@pytest.mark.parametrize("multi_head", [False, True])
def test_gpu_cross_attention_vit_vs_oracle_composition(favit, multi_head):
    favit.set_compute_dtype("fp32")
    torch.manual_seed(6)
    m = favit.models.attention.CrossAttentionViT(img_size=32, patch_size=4, num_classes=10, embed_dim=64, depth=2,
                                                 num_heads=4, use_multi_head=multi_head).eval()
    x, y = torch.randn(3, 3, 32, 32), torch.randint(0, 10, (3,))
    _vs_oracle(favit, m, x, y, lambda sd: O.cross_vit_forward(x, sd, 4, 4, multi_head))


@pytest.mark.gpu
@pytest.mark.parametrize("multi_head", [False, True])
def test_gpu_cross_attention_sppp_vit_vs_oracle_composition(favit, multi_head):
    favit.set_compute_dtype("fp32")
    torch.manual_seed(7)
    m = favit.models.attention.CrossAttentionSPPPViT(img_size=64, patch_size=16, num_classes=10, embed_dim=64, depth=2,
                                                     num_heads=4, num_superpixels=4, pooling_type="mean",
                                                     use_multi_head=multi_head).eval()
    segs = _small_segs()
    x, y = torch.randn(2, 3, 64, 64), torch.randint(0, 10, (2,))
    m.segmentation.set_label_maps(torch.from_numpy(segs).to(DEV))
    _vs_oracle(favit, m, x, y, lambda sd: O.cross_sppp_vit_forward(x, segs, sd, 16, 4, multi_head, S=4))
