"""Pin the CPU oracle (oracle/favit_oracle.py) to the golden vectors that were
captured from the reference's own modules (tests/golden/make_golden.py).
fp32 everywhere; tolerance 2e-5 rel-L2 (reference fp32-vs-fp64 noise is ~1e-6)."""
import numpy as np
import pytest
import torch

from conftest import case, load_golden, rel_l2, sd_of
from oracle import favit_oracle as O

TOL = 2e-5
D, H = 64, 4


def _fwd_bwd(c, fn, n_in=1):
    sd = {k: v.clone().requires_grad_(True) for k, v in sd_of(c).items()}
    ins = [torch.from_numpy(c[f"in{i}"]).clone().requires_grad_(True) for i in range(n_in)]
    y = fn(sd, *ins)
    assert rel_l2(y, c["out"]) < TOL, "forward"
    (y * torch.from_numpy(c["gout"])).sum().backward()
    for i, t in enumerate(ins):
        assert rel_l2(t.grad, c[f"gin{i}"]) < 5 * TOL, f"grad input {i}"
    for k, v in sd.items():
        g = v.grad if v.grad is not None else torch.zeros_like(v)
        ref = c[f"grad/{k}"]
        if np.abs(ref).max() < 1e-5:     # analytically-zero grads (e.g. key bias): fp32 noise only
            assert g.abs().max().item() < 1e-5, k
        else:
            assert rel_l2(g, ref) < 5 * TOL, f"grad {k}"


def test_window_indices_match_reference_tables():
    z = load_golden("windows.npz")
    for key in z.files:
        L, W = (int(s[1:]) for s in key.split("_"))
        np.testing.assert_array_equal(O.window_indices(L, W), z[key], err_msg=key)


def test_window_indices_probed_rows():
    # SURVEY 8a row a6 (probed on the reference)
    idx = O.window_indices(12, 7)
    assert idx[0].tolist() == [0, 1, 2, 3, 11, 11, 11]
    assert idx[11].tolist() == [0, 0, 0, 8, 9, 10, 11]
    idx = O.window_indices(5, 7)
    assert idx[3].tolist() == [0, 1, 2, 3, 4, 4, 4] and idx[4].tolist() == [0, 0, 0, 1, 2, 3, 4]
    with pytest.raises(ValueError):
        O.window_indices(12, 4)


MHLA = load_golden("mhla.npz")


@pytest.mark.parametrize("name", sorted({k.split("/")[0] for k in MHLA.files if k.startswith("attn_L")}))
def test_mhla_attention(name):
    W = int(name.split("_W")[1])
    _fwd_bwd(case(MHLA, name), lambda sd, x: O.mhla_attention(x, sd, "", H, W))


@pytest.mark.parametrize("name", sorted({k.split("/")[0] for k in MHLA.files if k.startswith("attn_mask")}))
def test_mhla_attention_masked(name):
    W = int(name.split("_W")[1])
    c = case(MHLA, name)
    mask = torch.from_numpy(c["attention_mask"])
    _fwd_bwd(c, lambda sd, x: O.mhla_attention(x, sd, "", H, W, mask))


@pytest.mark.parametrize("name", sorted({k.split("/")[0] for k in MHLA.files if k.startswith("block_")}))
def test_mhla_block(name):
    W = int(name.split("_W")[1])
    _fwd_bwd(case(MHLA, name), lambda sd, x: O.mhla_block(x, sd, "", H, W))


VP = load_golden("vit_parts.npz")


@pytest.mark.parametrize("name", ["mha_L5", "mha_L17", "mha_L65"])
def test_dense_mha(name):
    _fwd_bwd(case(VP, name), lambda sd, x: O.dense_mha(x, sd, "", H))


def test_mlp():
    _fwd_bwd(case(VP, "mlp"), lambda sd, x: O.mlp(x, sd, ""))


def test_vit_block():
    _fwd_bwd(case(VP, "block_L17"), lambda sd, x: O.vit_block(x, sd, "", H))


def test_patch_embed():
    _fwd_bwd(case(VP, "patch_embed"), lambda sd, x: O.patch_embed(x, sd, "", 4))


@pytest.mark.parametrize("use_mhla", [0, 1])
@pytest.mark.parametrize("L", [17, 65])
def test_vit_mhla_block(use_mhla, L):
    _fwd_bwd(case(VP, f"vm_block_mhla{use_mhla}_L{L}"),
             lambda sd, x: O.vit_mhla_block(x, sd, "", H, 7, bool(use_mhla)))


CR = load_golden("cross.npz")


@pytest.mark.parametrize("masked", [0, 1])
def test_cross_attention(masked):
    c = case(CR, f"ca_mask{masked}")
    m = torch.from_numpy(c["attention_mask"]) if masked else None
    _fwd_bwd(c, lambda sd, q, kv: O.cross_attention(q, kv, sd, "", m), n_in=2)


@pytest.mark.parametrize("masked", [0, 1])
def test_multihead_cross_attention(masked):
    c = case(CR, f"mhca_mask{masked}")
    m = torch.from_numpy(c["attention_mask"]) if masked else None
    _fwd_bwd(c, lambda sd, q, kv: O.multihead_cross_attention(q, kv, sd, "", H, m), n_in=2)


@pytest.mark.parametrize("mh", [0, 1])
def test_cross_block(mh):
    _fwd_bwd(case(CR, f"block_mh{mh}"), lambda sd, q, kv: O.cross_block(q, kv, sd, "", H, bool(mh)), n_in=2)


SP = load_golden("sppp.npz")


@pytest.mark.parametrize("nm", ["grid", "vor16", "vor15"])
def test_sppp_map_pool_centroid_posenc(nm):
    c = case(SP, nm)
    seg = c["segmap"].astype(np.int64)
    mapping = O.map_patches(seg, 224, 16)
    assert list(mapping.keys()) == c["map_keys"].tolist()
    rank = np.full(196, -1, dtype=np.int64)
    for r, (_, idx) in enumerate(mapping.items()):
        rank[idx] = r
    np.testing.assert_array_equal(rank, c["patch_rank"])
    for kind in ("mean", "max", "attention"):
        e = torch.from_numpy(SP["emb"]).clone().requires_grad_(True)
        p = O.pool(e, mapping, kind)
        assert rel_l2(p, c[f"pool_{kind}"]) < TOL
        (p * torch.from_numpy(c[f"pool_{kind}_gout"])).sum().backward()
        assert rel_l2(e.grad, c[f"pool_{kind}_gin"]) < TOL
    segs = np.stack([seg, np.roll(seg, 5, axis=1)])
    cent = O.superpixel_centroids(segs, 16)
    assert rel_l2(cent, c["centroids"]) < TOL
    pe = O.dynamic_posenc(torch.from_numpy(c["posenc_in"]), cent)
    assert rel_l2(pe, c["posenc_out"]) < TOL


def test_posenc_without_centroids():
    pe = O.dynamic_posenc(torch.from_numpy(SP["posenc_nocentroid_in"]), None)
    assert rel_l2(pe, SP["posenc_nocentroid_out"]) < TOL


@pytest.mark.parametrize("nm", ["grid", "vor16", "vor15"])
def test_sppp_model(nm):
    c = case(SP, nm)
    seg = c["segmap"].astype(np.int64)
    segs = np.stack([seg, np.roll(seg, 5, axis=1)])
    x = torch.from_numpy(c["model_x"].astype(np.float32))
    with torch.no_grad():
        y = O.sppp_vit_mhla_forward(x, segs, sd_of(c), 16, H, 7, True, S=16, kind="mean")
    assert rel_l2(y, c["logits"]) < 5 * TOL
