"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol include/favit.h
declares, the nn.Module mirrors keep the reference's constructor signatures / attribute names /
state_dict keys / weight-init RNG order, and the product refuses to run without a GPU (there is no
CPU fallback)."""
import ctypes
import importlib
import inspect

import numpy as np
import pytest
import torch

from conftest import load_golden

MD = load_golden("models.npz")
WIN = load_golden("windows.npz")


def test_library_loads_and_exports_every_declared_symbol(favit):
    lib = favit._abi.lib()
    declared = favit._abi.declared_symbols()
    assert len(declared) >= 25
    for s in declared:
        assert hasattr(lib, s), f"libfavit.so does not export {s}"
    assert set(favit._abi._SIGS) == set(declared), "ctypes signature table out of sync with include/favit.h"
    assert lib.favit_abi_version() == 8
    assert lib.favit_strerror(-2).decode().startswith("unsupported")


def test_gemm_descriptor_layout_matches_header(favit):
    # 8 pointers + 21 int64 + 9 int32 + 3 x 4-byte + uint64, naturally aligned
    assert ctypes.sizeof(favit._abi.GemmDesc) == 8 * 8 + 15 * 8 + 9 * 4 + 4 + 4 + 4 + 8 + 2 * 8      # ABI v2: + scale_a, scale_b
    assert favit._abi.GemmDesc.dropout_seed.offset % 8 == 0
    # and against the C compiler's view of include/favit.h, field by field
    import os, shutil, subprocess, tempfile
    if shutil.which("gcc"):
        fields = [f[0] for f in favit._abi.GemmDesc._fields_]
        src = "#include <stdio.h>\n#include <stddef.h>\n#include \"favit.h\"\nint main(){printf(\"%zu\", sizeof(favit_gemm_t));" + \
              "".join(f'printf(" %zu", offsetof(favit_gemm_t, {f}));' for f in fields) + "return 0;}"
        with tempfile.TemporaryDirectory() as d:
            with open(os.path.join(d, "t.c"), "w") as fh:
                fh.write(src)
            inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
            subprocess.check_call(["gcc", "-I", inc, os.path.join(d, "t.c"), "-o", os.path.join(d, "t")])
            out = subprocess.check_output([os.path.join(d, "t")], text=True).split()
        assert int(out[0]) == ctypes.sizeof(favit._abi.GemmDesc)
        for f, off in zip(fields, out[1:]):
            assert getattr(favit._abi.GemmDesc, f).offset == int(off), f


def _wgrad_descs(favit, T, D, nblocks):
    """favit_gemm_t descriptors of the weight-gradient problems of `nblocks` blocks (fc2, fc1, proj, folded qkv); the
    pointers are never dereferenced by the planning query (any aligned non-null value)."""
    G = favit._abi.GemmDesc
    shapes = [(D, 4 * D), (4 * D, D), (D, D), (3 * D, D)]
    arr = (G * (4 * nblocks))()
    i = 0
    for _ in range(nblocks):
        for N, Kd in shapes:
            d = arr[i]
            d.A = d.B = d.C = 0x10000
            d.M, d.N, d.K = N, Kd, T
            d.lda, d.ldb, d.ldc = N, Kd, Kd
            d.batch = d.batch_inner = 1
            d.in_dtype, d.out_dtype = favit._abi.BF16, favit._abi.F32
            d.alpha = 1.0
            d.accumulate = 1
            i += 1
    return arr, sum(n * k + n for n, k in shapes) * nblocks


@pytest.mark.parametrize("name,T,D,nblocks,want_splits", [
    ("cfg2, one block per launch", 256 * 197, 384, 1, 8),
    ("cfg2, four blocks per launch", 256 * 197, 384, 4, 8),      # splits of <= 6,400 tokens: long splits drift
    ("cfg4, one block per launch", 64 * 577, 768, 1, 8),
    ("cfg3, the whole encoder", 128 * 17, 384, 12, 1),           # 756 tiles: no split, no slabs, no reduction kernel
    ("cfg3, a graph segment of four blocks", 128 * 17, 384, 4, None),
    ("cfg5 second bucket, the whole encoder", 128 * 16, 384, 12, 1),
])
def test_grouped_weight_gradient_plan_on_the_host(favit, name, T, D, nblocks, want_splits):
    """The split count of the grouped weight-gradient launch is host logic (csrc/gemm.hip: grouped_plan -- chunks of
    <= 64 tiles, units packed into eight per-XCD queues, a cost model): favit_gemm_grouped_tn_workspace returns
    nsplit x slab bytes, or 0 when one split suffices and the tiles write dW themselves.  No GPU needed."""
    lib = favit._abi.lib()
    arr, out_floats = _wgrad_descs(favit, T, D, nblocks)
    ws = int(lib.favit_gemm_grouped_tn_workspace(arr, len(arr)))
    slab = (out_floats + 63) // 64 * 64 * 4                      # per problem [M, N] + [M] rounded to 4, the total to 64 floats
    assert ws % slab == 0, (name, ws, slab)
    splits = ws // slab if ws else 1
    assert 1 <= splits <= 32
    if want_splits is not None:
        assert splits == want_splits, (name, splits)
    assert int(lib.favit_gemm_grouped_tn_workspace(arr, 49)) == 0           # more than 48 problems: not one launch


def test_cfg2_model_init_parity_and_state_dict_keys(favit):
    """Same RNG call order as the reference: same seed -> same weights."""
    torch.manual_seed(1234)
    m = favit.models.vit_mhla.VisionTransformerMHLA(img_size=224, patch_size=16, num_classes=1000, embed_dim=384,
                                                    depth=12, num_heads=6, window_size=7, use_mhla=True)
    assert m.get_num_parameters() == int(MD["cfg2/n_params"]) == 22100584
    assert list(m.state_dict().keys()) == MD["cfg2/sd_keys"].tolist()
    assert abs(sum(p.double().sum().item() for p in m.parameters()) - float(MD["cfg2/param_sum"])) < 1e-9
    # attribute paths the reference's experiments touch (experiments/mhla_pretrained.py:188-234)
    assert m.patch_embed.projection[1].weight.shape == (384, 768)
    b = m.blocks[0]
    for path in ("norm1", "norm2", "mlp.fc1", "mlp.fc2", "attn.qkv", "attn.proj", "attn.latent_proj"):
        obj = b
        for part in path.split("."):
            obj = getattr(obj, part)
        assert isinstance(obj.weight, torch.nn.Parameter)
    assert m.embed_dim == 384 and m.cls_token.shape == (1, 1, 384) and m.pos_embed.shape == (1, 197, 384)


def test_cfg1_and_fallback_init_parity(favit):
    torch.manual_seed(1234)
    m = favit.models.vit.VisionTransformer(img_size=32, patch_size=4, num_classes=10, embed_dim=192, depth=12,
                                           num_heads=3)
    assert m.get_num_parameters() == int(MD["cfg1/n_params"]) == 5362762
    assert abs(sum(p.double().sum().item() for p in m.parameters()) - float(MD["cfg1/param_sum"])) < 1e-9
    torch.manual_seed(1234)
    m = favit.models.vit_mhla.VisionTransformerMHLA(img_size=32, patch_size=4, num_classes=10, embed_dim=64, depth=2,
                                                    num_heads=4, use_mhla=False)
    assert abs(sum(p.double().sum().item() for p in m.parameters()) - float(MD["fallback/param_sum"])) < 1e-9
    assert "blocks.0.attn.in_proj_weight" in m.state_dict() and "blocks.0.attn.out_proj.weight" in m.state_dict()


def test_sppp_model_has_no_pos_embed_parameter(favit):
    m = favit.models.sppp_mhla.SPPPViTMHLA(img_size=32, patch_size=4, num_classes=10, embed_dim=64, depth=1,
                                           num_heads=4, use_mhla=True)
    keys = list(m.state_dict().keys())
    assert "pos_embed" not in keys and "cls_token" in keys and "patch_embed.projection.1.weight" in keys


@pytest.mark.parametrize("cls,kw", [
    ("vit.PatchEmbedding", dict(img_size=224, patch_size=16, in_channels=3, embed_dim=768)),
    ("vit.MultiHeadAttention", dict(embed_dim=None, num_heads=None, dropout=0.0)),
    ("vit.MLP", dict(in_features=None, hidden_features=None, out_features=None, dropout=0.0)),
    ("vit.TransformerBlock", dict(embed_dim=None, num_heads=None, mlp_ratio=4.0, dropout=0.0, attn_dropout=0.0)),
    ("mhla.MultiHeadLatentAttention", dict(embed_dim=None, num_heads=None, window_size=7, dropout=0.0)),
    ("mhla.MHLATransformerBlock", dict(embed_dim=None, num_heads=None, window_size=7, mlp_ratio=4.0, dropout=0.0,
                                       attn_dropout=0.0)),
    ("vit_mhla.TransformerBlock", dict(embed_dim=None, num_heads=None, mlp_ratio=4.0, dropout=0.0, attn_dropout=0.0,
                                       window_size=7, use_mhla=False)),
    ("attention.CrossAttention", dict(embed_dim=None, dropout=0.0)),
    ("attention.MultiHeadCrossAttention", dict(embed_dim=None, num_heads=None, dropout=0.0)),
    ("attention.CrossAttentionTransformerBlock", dict(embed_dim=None, num_heads=None, mlp_ratio=4.0, dropout=0.0,
                                                      attn_dropout=0.0, use_multi_head=False)),
    ("sppp.DynamicPositionalEncoding", dict(embed_dim=None, dropout=0.0)),
])
def test_constructor_signatures_match_reference(favit, cls, kw):
    """Parameter names, order and defaults of SURVEY 8b (None = required positional)."""
    mod, name = cls.split(".")
    sig = inspect.signature(getattr(getattr(favit.models, mod), name).__init__)
    got = [(k, v.default) for k, v in list(sig.parameters.items())[1:]]
    want = [(k, inspect.Parameter.empty if v is None else v) for k, v in kw.items()]
    assert got == want


def test_model_constructor_kwargs(favit):
    names = list(inspect.signature(favit.models.vit_mhla.VisionTransformerMHLA.__init__).parameters)[1:]
    assert names == ["img_size", "patch_size", "in_channels", "num_classes", "embed_dim", "depth", "num_heads",
                     "mlp_ratio", "dropout", "attn_dropout", "embed_dropout", "window_size", "use_mhla"]
    names = list(inspect.signature(favit.models.sppp_mhla.SPPPViTMHLA.__init__).parameters)[1:]
    assert names[-5:] == ["num_superpixels", "compactness", "pooling_type", "window_size", "use_mhla"]


def test_window_index_table_matches_reference(favit):
    for key in WIN.files:
        L, W = (int(s[1:]) for s in key.split("_"))
        m = favit.models.mhla.MultiHeadLatentAttention(64, 4, window_size=W)
        np.testing.assert_array_equal(m._get_window_indices(L).numpy(), WIN[key], err_msg=key)
    with pytest.raises(ValueError):
        favit.models.mhla.MultiHeadLatentAttention(64, 4, window_size=4)._get_window_indices(12)


def test_no_cpu_fallback(favit):
    m = favit.models.vit.MLP(16, 32, 16)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.randn(2, 16))
    blk = favit.models.mhla.MHLATransformerBlock(16, 2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        blk(torch.randn(1, 5, 16))


def test_drop_in_import_as_top_level_models(tmp_path):
    """INTEGRATION.md mode: the package directory on sys.path gives `models.vit` etc."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); from models.vit_mhla import VisionTransformerMHLA; "
            "from models.mhla import MultiHeadLatentAttention; from models.sppp_mhla import SPPPViTMHLA; "
            "from models.attention import CrossAttentionTransformerBlock; from models.vit import VisionTransformer; "
            "m = VisionTransformerMHLA(img_size=32, patch_size=4, embed_dim=64, depth=1, num_heads=4, use_mhla=True); "
            "print(m.get_num_parameters())") % os.path.join(root, "focused-attention-vit_amd")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr
    assert int(out.stdout.strip()) > 0


def test_param_groups_follow_reference_names(favit):
    m = favit.models.vit_mhla.VisionTransformerMHLA(img_size=32, patch_size=4, num_classes=10, embed_dim=64, depth=2,
                                                    num_heads=4, use_mhla=True)
    groups = favit.train.param_groups(m, lr=1e-4, head_lr=1e-3)
    assert [g["lr"] for g in groups] == [1e-4, 5e-4, 1e-3]
    assert len(groups[1]["params"]) == 2 * 2 and len(groups[2]["params"]) == 2      # latent_proj w/b per block; head w/b


def test_flat_buffers_keep_parameter_semantics(favit):
    lin = torch.nn.Linear(5, 3)
    w0 = lin.weight.detach().clone()
    flat = favit.dp.FlatBuffers(lin.parameters())
    assert torch.equal(lin.weight, w0) and lin.weight.shape == (3, 5)
    lin.weight.data.copy_(torch.ones(3, 5))
    idx = next(i for i, p in enumerate(flat.params) if p is lin.weight)
    assert flat.flat_p[flat.offsets[idx]].item() == 1.0
    (lin(torch.ones(2, 5)).sum()).backward()
    assert flat.flat_g.abs().sum().item() > 0
    flat.zero_grad()
    assert flat.flat_g.abs().sum().item() == 0 and lin.weight.grad is not None
