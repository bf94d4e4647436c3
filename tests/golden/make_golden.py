#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/*.npz by IMPORTING THE REFERENCE.

Run once in the build container (the reference lives at /root/reference and never
travels to the GPU box; only the .npz data files produced here are committed):

    python tests/golden/make_golden.py

What is recorded is data only: seeds, inputs, state_dicts, outputs and gradients of
the reference's own nn.Modules executed on CPU in fp32.

skimage is not installed in the image and the reference pins no version of it.  The
only symbol the reference takes from it is ``slic`` (models/sppp.py:19,64,72).  To let
``import models.sppp`` succeed we register an EMPTY placeholder whose ``slic`` raises
if called; SLIC is never executed, the label map is an input of every SPPP fixture
(SLIC itself stays "parity unpinned").
"""
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("FAVIT_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(HERE, "..", ".."))


def _placeholder_skimage():
    def slic(*a, **k):  # pragma: no cover
        raise RuntimeError("SLIC is out of scope: label maps are inputs")
    sk = types.ModuleType("skimage")
    seg = types.ModuleType("skimage.segmentation")
    seg.slic = slic
    sk.segmentation = seg
    sys.modules.setdefault("skimage", sk)
    sys.modules.setdefault("skimage.segmentation", seg)


_placeholder_skimage()

from models import vit as rvit                      # noqa: E402
from models import mhla as rmhla                    # noqa: E402
from models import vit_mhla as rvm                  # noqa: E402
from models import attention as ratt                # noqa: E402
from models import sppp as rsppp                    # noqa: E402
from models import sppp_mhla as rsm                 # noqa: E402
from models import mhla_models as rmm               # noqa: E402

from oracle.favit_oracle import voronoi_labels      # noqa: E402  (input generator only)

D, H, B = 64, 4, 2


def np32(t):
    return t.detach().cpu().numpy().astype(np.float32)


def run_case(out, name, module, inputs, seed, extra=None, call=None):
    """fwd + bwd of a reference module; loss = sum(out * gout)."""
    g = torch.Generator().manual_seed(seed + 7)
    ins = [t.clone().requires_grad_(t.is_floating_point()) for t in inputs]
    y = (call or module)(*ins, **(extra or {}))
    gout = torch.randn(y.shape, generator=g)
    (y * gout).sum().backward()
    for k, v in module.state_dict().items():
        out[f"{name}/sd/{k}"] = np32(v)
    for k, p in module.named_parameters():
        out[f"{name}/grad/{k}"] = np32(p.grad if p.grad is not None else torch.zeros_like(p))
    for i, t in enumerate(ins):
        out[f"{name}/in{i}"] = t.detach().numpy()
        if t.is_floating_point():
            out[f"{name}/gin{i}"] = np32(t.grad)
    for k, v in (extra or {}).items():
        if torch.is_tensor(v):
            out[f"{name}/{k}"] = v.numpy()
    out[f"{name}/out"] = np32(y)
    out[f"{name}/gout"] = np32(gout)


def randomize(module, seed):
    """Reference modules built standalone keep nn.Linear default init; perturb every
    parameter (incl. LayerNorm affine, biases) so no term is trivially 0/1."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in module.named_parameters():
            if p.dim() == 1 and "norm" in n and n.endswith("weight"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            elif p.dim() == 1:
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(torch.randn(p.shape, generator=g) / (p.shape[-1] ** 0.5))


def save(fname, d):
    path = os.path.join(HERE, fname)
    np.savez_compressed(path, **d)
    print(f"{fname}: {len(d)} arrays, {os.path.getsize(path) / 1e6:.2f} MB")


def gen_windows():
    out = {}
    for L in (1, 3, 5, 7, 8, 12, 17, 65, 197, 577):
        for W in (3, 5, 7):
            m = rmhla.MultiHeadLatentAttention(64, 4, window_size=W)
            out[f"L{L}_W{W}"] = m._get_window_indices(L).numpy()
    save("windows.npz", out)


def gen_mhla():
    out = {}
    seed = 100
    for L, W in ((5, 3), (5, 7), (7, 7), (12, 3), (12, 7), (17, 7), (65, 3), (65, 7), (197, 7)):
        seed += 1
        torch.manual_seed(seed)
        m = rmhla.MultiHeadLatentAttention(D, H, window_size=W)
        randomize(m, seed)
        x = torch.randn(B, L, D)
        run_case(out, f"attn_L{L}_W{W}", m, [x], seed)
    for L, W in ((12, 7), (17, 3)):
        seed += 1
        torch.manual_seed(seed)
        m = rmhla.MultiHeadLatentAttention(D, H, window_size=W)
        randomize(m, seed)
        x = torch.randn(B, L, D)
        mask = (torch.rand(B, L, L) > 0.4).float()
        mask[:, torch.arange(L), torch.arange(L)] = 1.0       # every row keeps its own key
        run_case(out, f"attn_mask_L{L}_W{W}", m, [x], seed, extra={"attention_mask": mask})
    for L, W in ((17, 7), (65, 3)):
        seed += 1
        torch.manual_seed(seed)
        m = rmhla.MHLATransformerBlock(D, H, window_size=W)
        randomize(m, seed)
        run_case(out, f"block_L{L}_W{W}", m, [torch.randn(B, L, D)], seed)
    save("mhla.npz", out)


def gen_vit_parts():
    out = {}
    seed = 200
    for L in (5, 17, 65):
        seed += 1
        torch.manual_seed(seed)
        m = rvit.MultiHeadAttention(D, H)
        randomize(m, seed)
        run_case(out, f"mha_L{L}", m, [torch.randn(B, L, D)], seed)
    seed += 1
    torch.manual_seed(seed)
    m = rvit.MLP(D, 4 * D, D)
    randomize(m, seed)
    run_case(out, "mlp", m, [torch.randn(B, 17, D)], seed)
    seed += 1
    torch.manual_seed(seed)
    m = rvit.TransformerBlock(D, H)
    randomize(m, seed)
    run_case(out, "block_L17", m, [torch.randn(B, 17, D)], seed)
    seed += 1
    torch.manual_seed(seed)
    m = rvit.PatchEmbedding(img_size=32, patch_size=4, in_channels=3, embed_dim=D)
    randomize(m, seed)
    run_case(out, "patch_embed", m, [torch.randn(B, 3, 32, 32)], seed)
    for use_mhla in (True, False):
        for L in (17, 65):
            seed += 1
            torch.manual_seed(seed)
            m = rvm.TransformerBlock(D, H, window_size=7, use_mhla=use_mhla)
            randomize(m, seed)
            m.eval()
            run_case(out, f"vm_block_mhla{int(use_mhla)}_L{L}", m, [torch.randn(B, L, D)], seed)
    save("vit_parts.npz", out)


def gen_cross():
    out = {}
    seed = 300
    for masked in (False, True):
        for cls, nm in ((ratt.CrossAttention, "ca"), (ratt.MultiHeadCrossAttention, "mhca")):
            seed += 1
            torch.manual_seed(seed)
            m = cls(D) if nm == "ca" else cls(D, H)
            randomize(m, seed)
            q, kv = torch.randn(B, 7, D), torch.randn(B, 12, D)
            extra = None
            if masked:
                mask = (torch.rand(B, 7, 12) > 0.4).float()
                mask[:, :, 0] = 1.0
                extra = {"attention_mask": mask}
            run_case(out, f"{nm}_mask{int(masked)}", m, [q, kv], seed, extra=extra)
    for mh in (False, True):
        seed += 1
        torch.manual_seed(seed)
        m = ratt.CrossAttentionTransformerBlock(D, H, use_multi_head=mh)
        randomize(m, seed)
        run_case(out, f"block_mh{int(mh)}", m, [torch.randn(B, 9, D), torch.randn(B, 17, D)], seed)
    save("cross.npz", out)


def gen_sppp():
    out = {}
    img, P, S = 224, 16, 16
    g = img // P
    grid = np.repeat(np.repeat(np.arange(16).reshape(4, 4), img // 4, axis=0), img // 4, axis=1).astype(np.int64)
    maps = {"grid": grid, "vor16": voronoi_labels(img, 16, seed=3), "vor15": voronoi_labels(img, 15, seed=5)}
    mapper = rsppp.PatchToSuperpixelMapper(P)
    gen = torch.Generator().manual_seed(400)
    emb = torch.randn(g * g, D, generator=gen)
    out["emb"] = np32(emb)
    for nm, sm in maps.items():
        out[f"{nm}/segmap"] = sm.astype(np.uint8)
        mapping = mapper.map_patches(torch.from_numpy(sm), img)
        out[f"{nm}/map_keys"] = np.asarray(list(mapping.keys()), dtype=np.int64)
        rank = np.full(g * g, -1, dtype=np.int64)
        for r, (_, idx) in enumerate(mapping.items()):
            rank[idx] = r
        out[f"{nm}/patch_rank"] = rank
        for kind in ("mean", "max", "attention"):
            e = emb.clone().requires_grad_(True)
            pooled = rsppp.SuperpixelPooling(kind).pool(e, mapping)
            gout = torch.randn(pooled.shape, generator=gen)
            (pooled * gout).sum().backward()
            out[f"{nm}/pool_{kind}"] = np32(pooled)
            out[f"{nm}/pool_{kind}_gout"] = np32(gout)
            out[f"{nm}/pool_{kind}_gin"] = np32(e.grad)
    # centroids + positional encoding + whole SPPPViTMHLA (tiny config)
    for nm in ("grid", "vor16", "vor15"):
        torch.manual_seed(410)
        model = rsm.SPPPViTMHLA(img_size=img, patch_size=P, num_classes=10, embed_dim=D, depth=2, num_heads=H,
                                num_superpixels=S, pooling_type="mean", window_size=7, use_mhla=True)
        model.eval()
        segs = torch.from_numpy(np.stack([maps[nm], np.roll(maps[nm], 5, axis=1)]))
        model.segmentation.segment = lambda x, _s=segs: _s        # label maps are inputs
        cent = model._calculate_superpixel_centroids(segs)
        out[f"{nm}/centroids"] = np32(cent)
        R = len(mapper.map_patches(segs[0], img))
        tok = torch.randn(2, R + 1, D, generator=gen)
        out[f"{nm}/posenc_in"] = np32(tok)
        out[f"{nm}/posenc_out"] = np32(rsppp.DynamicPositionalEncoding(D)(tok, cent))
        x = torch.randn(2, 3, img, img, generator=gen)
        out[f"{nm}/model_x_sum"] = np.float64(x.double().sum().item())
        out[f"{nm}/model_x_seed"] = np.int64(0)
        out[f"{nm}/model_x"] = x.numpy().astype(np.float16)        # fp16-rounded inputs keep the file small
        x = torch.from_numpy(out[f"{nm}/model_x"].astype(np.float32))
        for k, v in model.state_dict().items():
            out[f"{nm}/sd/{k}"] = np32(v)
        out[f"{nm}/logits"] = np32(model(x))
    tok = torch.randn(2, 9, D, generator=gen)
    out["posenc_nocentroid_in"] = np32(tok)
    out["posenc_nocentroid_out"] = np32(rsppp.DynamicPositionalEncoding(D)(tok, None))
    save("sppp.npz", out)


def gen_models():
    out = {}
    # cfg1: ViT-Tiny dense, CIFAR shape (BASELINE.json configs[0])
    torch.manual_seed(1234)
    m = rvit.VisionTransformer(img_size=32, patch_size=4, num_classes=10, embed_dim=192, depth=12, num_heads=3)
    m.eval()
    x = torch.randn(4, 3, 32, 32)
    y = torch.randint(0, 10, (4,))
    logits = m(x)
    loss = torch.nn.CrossEntropyLoss()(logits, y)
    loss.backward()
    out["cfg1/x"], out["cfg1/y"] = x.numpy(), y.numpy()
    out["cfg1/logits"], out["cfg1/loss"] = np32(logits), np32(loss)
    out["cfg1/param_sum"] = np.float64(sum(p.double().sum().item() for p in m.parameters()))
    out["cfg1/n_params"] = np.int64(m.get_num_parameters())
    for k, p in m.named_parameters():
        out[f"cfg1/gnorm/{k}"] = np.float32(p.grad.norm().item())
    # cfg2: ViT-MHLA-Small 224/p16 (BASELINE.json configs[1]), B=2
    torch.manual_seed(1234)
    m = rvm.VisionTransformerMHLA(img_size=224, patch_size=16, num_classes=1000, embed_dim=384, depth=12,
                                  num_heads=6, window_size=7, use_mhla=True)
    m.eval()
    x = torch.randn(2, 3, 224, 224)
    y = torch.randint(0, 1000, (2,))
    logits = m(x)
    loss = torch.nn.CrossEntropyLoss()(logits, y)
    loss.backward()
    out["cfg2/x_sum"] = np.float64(x.double().sum().item())
    out["cfg2/y"] = y.numpy()
    out["cfg2/logits"], out["cfg2/loss"] = np32(logits), np32(loss)
    out["cfg2/param_sum"] = np.float64(sum(p.double().sum().item() for p in m.parameters()))
    out["cfg2/n_params"] = np.int64(m.get_num_parameters())
    for k, p in m.named_parameters():
        out[f"cfg2/gnorm/{k}"] = np.float32(p.grad.norm().item())
    out["cfg2/sd_keys"] = np.asarray(list(m.state_dict().keys()))
    # use_mhla=False (nn.MultiheadAttention fallback) small model
    torch.manual_seed(1234)
    m = rvm.VisionTransformerMHLA(img_size=32, patch_size=4, num_classes=10, embed_dim=64, depth=2,
                                  num_heads=4, use_mhla=False)
    m.eval()
    x = torch.randn(2, 3, 32, 32)
    out["fallback/x"] = x.numpy()
    out["fallback/logits"] = np32(m(x))
    out["fallback/param_sum"] = np.float64(sum(p.double().sum().item() for p in m.parameters()))
    save("models.npz", out)


def _gnorms(out, key, m):
    for k, p in m.named_parameters():
        out[f"{key}/gnorm/{k}"] = np.float32(0.0 if p.grad is None else p.grad.norm().item())


def _maps_with_R(img, P, n_regions, want_R, count, seed0):
    """`count` seeded Voronoi label maps whose reference patch->superpixel dict has `want_R` keys."""
    mapper = rsppp.PatchToSuperpixelMapper(P)
    maps, seed = [], seed0
    while len(maps) < count:
        sm = voronoi_labels(img, n_regions, seed=seed)
        if len(mapper.map_patches(torch.from_numpy(sm), img)) == want_R:
            maps.append((seed, sm))
        seed += 1
    return maps


def gen_configs():
    """Whole-model fixtures at BASELINE.json configs[2..4] (weights and inputs are regenerated from the
    seed by the tests: only seeds, label maps, logits, loss and gradient norms are stored)."""
    out = {}
    ce = torch.nn.CrossEntropyLoss()
    # cfg3: SPPP+MHLA Small, 224/p16, 16 superpixels (R = 16 -> 17 tokens), mean pooling
    maps16 = _maps_with_R(224, 16, 16, 16, 4, 100)
    out["cfg3/map_seeds"] = np.asarray([s for s, _ in maps16], dtype=np.int64)
    segs = torch.from_numpy(np.stack([m for _, m in maps16]))
    out["cfg3/segmaps"] = segs.numpy().astype(np.uint8)
    torch.manual_seed(1234)
    m = rsm.SPPPViTMHLA(img_size=224, patch_size=16, num_classes=1000, embed_dim=384, depth=12, num_heads=6,
                        num_superpixels=16, pooling_type="mean", window_size=7, use_mhla=True)
    m.eval()
    m.segmentation.segment = lambda x, _s=segs: _s
    x = torch.randn(4, 3, 224, 224)
    y = torch.randint(0, 1000, (4,))
    logits = m(x)
    loss = ce(logits, y)
    loss.backward()
    out["cfg3/x_sum"], out["cfg3/y"] = np.float64(x.double().sum().item()), y.numpy()
    out["cfg3/logits"], out["cfg3/loss"] = np32(logits), np32(loss)
    out["cfg3/param_sum"] = np.float64(sum(p.double().sum().item() for p in m.parameters()))
    _gnorms(out, "cfg3", m)
    # cfg5: the fine-tune setup of experiments/sppp_mhla_pretrained.py:236-237 (identity latent_proj,
    # zero bias) with "mixed superpixel counts": one bucket with R = 16 and one with R = 15
    maps15 = _maps_with_R(224, 16, 15, 15, 2, 200)
    out["cfg5/map_seeds15"] = np.asarray([s for s, _ in maps15], dtype=np.int64)
    segs15 = torch.from_numpy(np.stack([m_ for _, m_ in maps15]))
    out["cfg5/segmaps15"] = segs15.numpy().astype(np.uint8)
    torch.manual_seed(4321)
    m = rsm.SPPPViTMHLA(img_size=224, patch_size=16, num_classes=1000, embed_dim=384, depth=12, num_heads=6,
                        num_superpixels=16, pooling_type="mean", window_size=7, use_mhla=True)
    for blk in m.blocks:
        torch.nn.init.eye_(blk.attn.latent_proj.weight)
        torch.nn.init.zeros_(blk.attn.latent_proj.bias)
    m.eval()
    x16 = torch.randn(2, 3, 224, 224)
    x15 = torch.randn(2, 3, 224, 224)
    y16 = torch.randint(0, 1000, (2,))
    m.segmentation.segment = lambda x, _s=segs[:2]: _s
    lg16 = m(x16)
    loss = ce(lg16, y16)
    loss.backward()
    out["cfg5/x16_sum"], out["cfg5/x15_sum"] = np.float64(x16.double().sum().item()), np.float64(x15.double().sum().item())
    out["cfg5/y16"] = y16.numpy()
    out["cfg5/logits16"], out["cfg5/loss16"] = np32(lg16), np32(loss)
    _gnorms(out, "cfg5", m)
    m.segmentation.segment = lambda x, _s=segs15: _s
    with torch.no_grad():
        out["cfg5/logits15"] = np32(m(x15))
    # cfg4: ViT-MHLA-Base 384/p16 (577 tokens), one image
    torch.manual_seed(1234)
    m = rvm.VisionTransformerMHLA(img_size=384, patch_size=16, num_classes=1000, embed_dim=768, depth=12,
                                  num_heads=12, window_size=7, use_mhla=True)
    m.eval()
    x = torch.randn(1, 3, 384, 384)
    y = torch.randint(0, 1000, (1,))
    logits = m(x)
    loss = ce(logits, y)
    loss.backward()
    out["cfg4/x_sum"], out["cfg4/y"] = np.float64(x.double().sum().item()), y.numpy()
    out["cfg4/logits"], out["cfg4/loss"] = np32(logits), np32(loss)
    out["cfg4/param_sum"] = np.float64(sum(p.double().sum().item() for p in m.parameters()))
    out["cfg4/n_params"] = np.int64(m.get_num_parameters())
    _gnorms(out, "cfg4", m)
    save("configs.npz", out)


def gen_wrappers():
    """The two whole-model wrappers of models/mhla_models.py that the reference CAN construct (with an odd
    window_size; its default 4 crashes at the first forward).  SPPPViT, CrossAttentionViT and CrossAttentionSPPPViT
    raise in their constructors in the reference (sppp.py:378, attention.py:275,454): no fixture is possible."""
    out = {}
    ce = torch.nn.CrossEntropyLoss()
    torch.manual_seed(77)
    m = rmm.PretrainedViTWithMHLA(img_size=32, patch_size=4, num_classes=10, embed_dim=64, depth=2, num_heads=4,
                                  window_size=7)
    m.eval()
    x = torch.randn(3, 3, 32, 32)
    y = torch.randint(0, 10, (3,))
    logits = m(x)
    loss = ce(logits, y)
    loss.backward()
    out["pvit/x"], out["pvit/y"] = x.numpy(), y.numpy()
    out["pvit/logits"], out["pvit/loss"] = np32(logits), np32(loss)
    out["pvit/param_sum"] = np.float64(sum(p.double().sum().item() for p in m.parameters()))
    out["pvit/n_params"] = np.int64(m.get_num_parameters())
    out["pvit/sd_keys"] = np.asarray(list(m.state_dict().keys()))
    _gnorms(out, "pvit", m)
    # SPPP wrapper: 64x64 images, 16x16 patches (4x4 grid), 4 superpixels; label maps are inputs
    maps = _maps_with_R(64, 16, 4, 4, 2, 300)
    segs = torch.from_numpy(np.stack([sm for _, sm in maps]))
    out["psppp/segmaps"] = segs.numpy().astype(np.uint8)
    for kind in ("mean", "max", "attention"):
        torch.manual_seed(78)
        m = rmm.PretrainedSPPPViTWithMHLA(img_size=64, patch_size=16, num_classes=10, embed_dim=64, depth=2, num_heads=4,
                                          window_size=3, num_superpixels=4, pooling_type=kind)
        m.eval()
        m.segmentation.segment = lambda x, _s=segs: _s
        x = torch.randn(2, 3, 64, 64)
        y = torch.randint(0, 10, (2,))
        logits = m(x)
        loss = ce(logits, y)
        loss.backward()
        out[f"psppp_{kind}/x"], out[f"psppp_{kind}/y"] = x.numpy(), y.numpy()
        out[f"psppp_{kind}/logits"], out[f"psppp_{kind}/loss"] = np32(logits), np32(loss)
        out[f"psppp_{kind}/param_sum"] = np.float64(sum(p.double().sum().item() for p in m.parameters()))
        _gnorms(out, f"psppp_{kind}", m)
    save("wrappers.npz", out)


if __name__ == "__main__":
    torch.set_num_threads(8)
    gen_windows()
    gen_mhla()
    gen_vit_parts()
    gen_cross()
    gen_sppp()
    gen_models()
    gen_configs()
    gen_wrappers()
