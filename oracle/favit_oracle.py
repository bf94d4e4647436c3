"""CPU oracle: a plain fp32 PyTorch/numpy restatement of the reference hot path.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Every function is a
*functional* restatement (state_dict in, tensor out) of one reference function and
cites the reference file:line it follows (paths relative to the reference repo
``zser092/Focused-Attention-ViT``).  Backward passes come from autograd over these
restatements.

Pinning: ``tests/test_oracle_golden.py`` checks every function here against the
golden vectors in ``tests/golden/*.npz``, which were produced by importing the
reference's own modules in the build container (``tests/golden/make_golden.py``).
SLIC (``skimage.segmentation.slic``, version unpinned by the reference and absent
from the image) is NOT restated: the label map is an input -> "parity unpinned"
for SLIC itself, pinned for everything downstream of the label map.

The algorithms are deliberately written differently from the reference where that
gives an independent check (MHLA is evaluated as dense attention with a
``log(multiplicity)`` bias instead of the reference's gather of windows).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# --------------------------------------------------------------------------- #
# primitives
# --------------------------------------------------------------------------- #
def linear(x: Tensor, w: Tensor, b: Optional[Tensor]) -> Tensor:
    """nn.Linear: y = x W^T + b  (models/vit.py:40,72,74,119,121,252)."""
    y = x @ w.t()
    return y if b is None else y + b


def layer_norm(x: Tensor, w: Tensor, b: Tensor, eps: float = 1e-5) -> Tensor:
    """nn.LayerNorm over the last dim, biased variance, eps=1e-5
    (models/vit.py:155,157,251; models/vit_mhla.py:45,64,188)."""
    mu = x.mean(dim=-1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(dim=-1, keepdim=True)
    return xc * torch.rsqrt(var + eps) * w + b


def gelu(x: Tensor) -> Tensor:
    """nn.GELU() default = exact erf form (models/vit.py:120)."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def mlp(x: Tensor, sd: SD, p: str, fc1: str = "fc1", fc2: str = "fc2") -> Tensor:
    """MLP.forward fc1 -> GELU -> fc2, dropout p=0 (models/vit.py:124-139).
    The nn.Sequential twins (models/mhla.py:197-203, models/attention.py:186-192)
    use fc1='0', fc2='3'."""
    h = gelu(linear(x, sd[f"{p}{fc1}.weight"], sd[f"{p}{fc1}.bias"]))
    return linear(h, sd[f"{p}{fc2}.weight"], sd[f"{p}{fc2}.bias"])


def patch_rearrange(x: Tensor, P: int) -> Tensor:
    """einops 'b c (h p1) (w p2) -> b (h w) (p1 p2 c)' (models/vit.py:38-39):
    patch vector order is (row, col, channel) with channel fastest."""
    B, C, H, W = x.shape
    gh, gw = H // P, W // P
    x = x.reshape(B, C, gh, P, gw, P)          # b c h p1 w p2
    x = x.permute(0, 2, 4, 3, 5, 1)            # b h w p1 p2 c
    return x.reshape(B, gh * gw, P * P * C)


def patch_embed(x: Tensor, sd: SD, p: str, P: int) -> Tensor:
    """PatchEmbedding.forward (models/vit.py:43-53)."""
    return linear(patch_rearrange(x, P), sd[f"{p}projection.1.weight"], sd[f"{p}projection.1.bias"])


# --------------------------------------------------------------------------- #
# MHLA: windowed "latent" attention  (models/mhla.py)
# --------------------------------------------------------------------------- #
def window_indices(L: int, W: int) -> np.ndarray:
    """Closed form of MultiHeadLatentAttention._get_window_indices
    (models/mhla.py:46-83).  Row i: window [max(0,i-h), min(L,i+h+1)); a short
    window that starts at 0 is padded at the END with L-1, any other short window
    is padded at the FRONT with 0.  Even W makes interior rows W+1 long and the
    reference's torch.stack raises (models/mhla.py:83) -> ValueError here."""
    if W % 2 == 0:
        raise ValueError("window_size must be odd (reference crashes on even sizes, models/mhla.py:83)")
    h = W // 2
    out = np.empty((L, W), dtype=np.int64)
    for i in range(L):
        lo, hi = max(0, i - h), min(L, i + h + 1)
        n = hi - lo
        if n == W:
            out[i] = np.arange(lo, hi)
        elif lo == 0:
            out[i, :n] = np.arange(lo, hi)
            out[i, n:] = L - 1
        else:
            out[i, : W - n] = 0
            out[i, W - n:] = np.arange(lo, hi)
    return out


def window_multiplicity(L: int, W: int) -> np.ndarray:
    """mult[i, j] = number of times key j appears in row i's window (duplicates
    from the padding rule take part in the reference softmax, models/mhla.py:146)."""
    idx = window_indices(L, W)
    m = np.zeros((L, L), dtype=np.float32)
    for i in range(L):
        for j in idx[i]:
            m[i, j] += 1.0
    return m


def mhla_attention(x: Tensor, sd: SD, p: str, H: int, W: int, mask: Optional[Tensor] = None) -> Tensor:
    """MultiHeadLatentAttention.forward (models/mhla.py:85-161), dropout p=0.
    Restated as dense attention with additive bias log(mult[i,j]) (-inf outside the
    window): softmax over a window with duplicates == softmax(s + log mult) over
    the distinct keys.  mask[B,L,L]==0 entries are -inf (models/mhla.py:136-143)."""
    B, L, D = x.shape
    hd = D // H
    qkv = linear(x, sd[f"{p}qkv.weight"], sd[f"{p}qkv.bias"]).reshape(B, L, 3, H, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]                                   # models/mhla.py:100-102
    wl, bl = sd[f"{p}latent_proj.weight"], sd[f"{p}latent_proj.bias"]
    k = linear(k, wl, bl)                                              # models/mhla.py:105
    v = linear(v, wl, bl)                                              # models/mhla.py:106
    s = (q @ k.transpose(-2, -1)) / (hd ** 0.5)                        # models/mhla.py:130-133
    mult = torch.from_numpy(window_multiplicity(L, W)).to(x.dtype)
    bias = torch.where(mult > 0, torch.log(mult.clamp_min(1.0)), torch.full_like(mult, float("-inf")))
    s = s + bias
    if mask is not None:
        s = s.masked_fill(mask[:, None, :, :] == 0, float("-inf"))
    a = torch.softmax(s, dim=-1)
    o = (a @ v).transpose(1, 2).reshape(B, L, D)                       # models/mhla.py:151-157
    return linear(o, sd[f"{p}proj.weight"], sd[f"{p}proj.bias"])       # models/mhla.py:158


# --------------------------------------------------------------------------- #
# dense attention variants
# --------------------------------------------------------------------------- #
def dense_mha(x: Tensor, sd: SD, p: str, H: int) -> Tensor:
    """vit.MultiHeadAttention.forward (models/vit.py:77-104): softmax((q k^T) * hd**-0.5) v."""
    B, L, D = x.shape
    hd = D // H
    qkv = linear(x, sd[f"{p}qkv.weight"], sd[f"{p}qkv.bias"]).reshape(B, L, 3, H, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    a = torch.softmax((q @ k.transpose(-2, -1)) * (hd ** -0.5), dim=-1)
    o = (a @ v).transpose(1, 2).reshape(B, L, D)
    return linear(o, sd[f"{p}proj.weight"], sd[f"{p}proj.bias"])


def torch_mha(x: Tensor, sd: SD, p: str, H: int, key_keep: Optional[Tensor] = None) -> Tensor:
    """nn.MultiheadAttention(batch_first=True) self-attention as used by the
    use_mhla=False branch (models/vit_mhla.py:57-62,96-101).  key_keep[B,L] bool is
    the reference's ``attention_mask`` (key_padding_mask = ~attention_mask)."""
    B, L, D = x.shape
    hd = D // H
    qkv = linear(x, sd[f"{p}in_proj_weight"], sd[f"{p}in_proj_bias"]).reshape(B, L, 3, H, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    s = (q * (hd ** -0.5)) @ k.transpose(-2, -1)
    if key_keep is not None:
        s = s.masked_fill(~key_keep[:, None, None, :], float("-inf"))
    o = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B, L, D)
    return linear(o, sd[f"{p}out_proj.weight"], sd[f"{p}out_proj.bias"])


def cross_attention(q_in: Tensor, kv_in: Tensor, sd: SD, p: str, mask: Optional[Tensor] = None) -> Tensor:
    """CrossAttention.forward (models/attention.py:37-78): ONE head, scores divided
    by embed_dim**0.5 (attention.py:64), no dropout after out_proj."""
    D = q_in.shape[-1]
    q = linear(q_in, sd[f"{p}q_proj.weight"], sd[f"{p}q_proj.bias"])
    k = linear(kv_in, sd[f"{p}k_proj.weight"], sd[f"{p}k_proj.bias"])
    v = linear(kv_in, sd[f"{p}v_proj.weight"], sd[f"{p}v_proj.bias"])
    s = (q @ k.transpose(1, 2)) / (D ** 0.5)
    if mask is not None:
        s = s.masked_fill(mask == 0, float("-inf"))
    return linear(torch.softmax(s, dim=-1) @ v, sd[f"{p}out_proj.weight"], sd[f"{p}out_proj.bias"])


def multihead_cross_attention(q_in: Tensor, kv_in: Tensor, sd: SD, p: str, H: int,
                              mask: Optional[Tensor] = None) -> Tensor:
    """MultiHeadCrossAttention.forward (models/attention.py:105-148): / hd**0.5,
    mask [B,Lq,Lk] broadcast over heads (attention.py:134-137)."""
    B, Lq, D = q_in.shape
    Lk = kv_in.shape[1]
    hd = D // H
    q = linear(q_in, sd[f"{p}q_proj.weight"], sd[f"{p}q_proj.bias"]).reshape(B, Lq, H, hd).permute(0, 2, 1, 3)
    k = linear(kv_in, sd[f"{p}k_proj.weight"], sd[f"{p}k_proj.bias"]).reshape(B, Lk, H, hd).permute(0, 2, 1, 3)
    v = linear(kv_in, sd[f"{p}v_proj.weight"], sd[f"{p}v_proj.bias"]).reshape(B, Lk, H, hd).permute(0, 2, 1, 3)
    s = (q @ k.transpose(-2, -1)) / (hd ** 0.5)
    if mask is not None:
        s = s.masked_fill(mask[:, None] == 0, float("-inf"))
    o = (torch.softmax(s, dim=-1) @ v).permute(0, 2, 1, 3).reshape(B, Lq, D)
    return linear(o, sd[f"{p}out_proj.weight"], sd[f"{p}out_proj.bias"])


# --------------------------------------------------------------------------- #
# blocks
# --------------------------------------------------------------------------- #
def vit_block(x: Tensor, sd: SD, p: str, H: int) -> Tensor:
    """vit.TransformerBlock.forward, pre-LN residual (models/vit.py:165-179)."""
    x = x + dense_mha(layer_norm(x, sd[f"{p}norm1.weight"], sd[f"{p}norm1.bias"]), sd, f"{p}attn.", H)
    return x + mlp(layer_norm(x, sd[f"{p}norm2.weight"], sd[f"{p}norm2.bias"]), sd, f"{p}mlp.")


def vit_mhla_block(x: Tensor, sd: SD, p: str, H: int, W: int, use_mhla: bool,
                   mask: Optional[Tensor] = None) -> Tensor:
    """vit_mhla.TransformerBlock.forward (models/vit_mhla.py:77-109; byte-identical
    logic in models/sppp_mhla.py:78-110)."""
    xn = layer_norm(x, sd[f"{p}norm1.weight"], sd[f"{p}norm1.bias"])
    if use_mhla:
        a = mhla_attention(xn, sd, f"{p}attn.", H, W, mask)
    else:
        a = torch_mha(xn, sd, f"{p}attn.", H, mask)
    x = x + a
    return x + mlp(layer_norm(x, sd[f"{p}norm2.weight"], sd[f"{p}norm2.bias"]), sd, f"{p}mlp.")


def mhla_block(x: Tensor, sd: SD, p: str, H: int, W: int, mask: Optional[Tensor] = None) -> Tensor:
    """mhla.MHLATransformerBlock.forward (models/mhla.py:205-222); mlp is an
    nn.Sequential with Linear at index 0 and 3 (mhla.py:197-203)."""
    x = x + mhla_attention(layer_norm(x, sd[f"{p}norm1.weight"], sd[f"{p}norm1.bias"]), sd, f"{p}attn.", H, W, mask)
    return x + mlp(layer_norm(x, sd[f"{p}norm2.weight"], sd[f"{p}norm2.bias"]), sd, f"{p}mlp.", "0", "3")


def cross_block(q: Tensor, kv: Tensor, sd: SD, p: str, H: int, multi_head: bool,
                mask: Optional[Tensor] = None) -> Tensor:
    """CrossAttentionTransformerBlock.forward (models/attention.py:194-219): two
    separate LayerNorms for query and key/value (attention.py:173-174)."""
    qn = layer_norm(q, sd[f"{p}norm1_query.weight"], sd[f"{p}norm1_query.bias"])
    kn = layer_norm(kv, sd[f"{p}norm1_kv.weight"], sd[f"{p}norm1_kv.bias"])
    if multi_head:
        a = multihead_cross_attention(qn, kn, sd, f"{p}attn.", H, mask)
    else:
        a = cross_attention(qn, kn, sd, f"{p}attn.", mask)
    q = q + a
    return q + mlp(layer_norm(q, sd[f"{p}norm2.weight"], sd[f"{p}norm2.bias"]), sd, f"{p}mlp.", "0", "3")


# --------------------------------------------------------------------------- #
# whole models
# --------------------------------------------------------------------------- #
def _depth(sd: SD) -> int:
    return 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))


def vit_forward(x: Tensor, sd: SD, P: int, H: int) -> Tensor:
    """VisionTransformer.forward (models/vit.py:276-322)."""
    B = x.shape[0]
    t = patch_embed(x, sd, "patch_embed.", P)
    t = torch.cat((sd["cls_token"].expand(B, -1, -1), t), dim=1) + sd["pos_embed"]
    for i in range(_depth(sd)):
        t = vit_block(t, sd, f"blocks.{i}.", H)
    t = layer_norm(t, sd["norm.weight"], sd["norm.bias"])[:, 0]
    return linear(t, sd["head.weight"], sd["head.bias"])


def vit_mhla_forward(x: Tensor, sd: SD, P: int, H: int, W: int, use_mhla: bool) -> Tensor:
    """VisionTransformerMHLA.forward (models/vit_mhla.py:213-259)."""
    B = x.shape[0]
    t = patch_embed(x, sd, "patch_embed.", P)
    t = torch.cat((sd["cls_token"].expand(B, -1, -1), t), dim=1) + sd["pos_embed"]
    for i in range(_depth(sd)):
        t = vit_mhla_block(t, sd, f"blocks.{i}.", H, W, use_mhla)
    t = layer_norm(t, sd["norm.weight"], sd["norm.bias"])[:, 0]
    return linear(t, sd["head.weight"], sd["head.bias"])


# --------------------------------------------------------------------------- #
# SPPP: superpixel patch pooling (models/sppp.py, models/sppp_mhla.py).
# The label map [H,W] int64 is an INPUT (SLIC is out of scope / unpinned).
# --------------------------------------------------------------------------- #
def map_patches(segmap: np.ndarray, img_size: int, P: int) -> Dict[int, List[int]]:
    """PatchToSuperpixelMapper.map_patches (models/sppp.py:91-128).  Dominant label
    of a patch = sorted-unique label with the largest count, ties -> smallest label
    (sppp.py:117-120); dict keys appear in first-seen raster order (sppp.py:124-126)."""
    g = img_size // P
    out: Dict[int, List[int]] = {}
    for i in range(g):
        for j in range(g):
            u, c = np.unique(segmap[i * P:(i + 1) * P, j * P:(j + 1) * P], return_counts=True)
            dom = int(u[int(np.argmax(c))])
            out.setdefault(dom, []).append(i * g + j)
    return out


def pool(emb: Tensor, mapping: Dict[int, List[int]], kind: str = "mean") -> Tensor:
    """SuperpixelPooling.pool, 2-D branch (models/sppp.py:192-223).  Token r is the
    r-th dict entry (insertion rank, not the label)."""
    rows = []
    for _, idx in mapping.items():
        e = emb[idx, :]
        if kind == "mean":
            rows.append(e.mean(dim=0))
        elif kind == "max":
            rows.append(e.max(dim=0)[0])
        elif kind == "attention":
            w = torch.softmax(e.sum(dim=-1), dim=-1)
            rows.append((e * w[:, None]).sum(dim=0))
        else:
            raise ValueError(f"Unsupported pooling type: {kind}")
    return torch.stack(rows)


def superpixel_centroids(segmaps: np.ndarray, S: int) -> Tensor:
    """SPPPViTMHLA._calculate_superpixel_centroids (models/sppp_mhla.py:226-262):
    per LABEL s < S the mean of x/w and y/h over its pixels, empty -> (0.5, 0.5);
    out[...,0] = x, out[...,1] = y."""
    B, Hh, Ww = segmaps.shape
    out = torch.zeros(B, S, 2)
    ys = (torch.arange(Hh).float() / Hh)[:, None].expand(Hh, Ww)
    xs = (torch.arange(Ww).float() / Ww)[None, :].expand(Hh, Ww)
    for b in range(B):
        sm = torch.from_numpy(segmaps[b])
        for s in range(S):
            m = (sm == s).float()
            n = m.sum()
            if n > 0:
                out[b, s, 0] = (xs * m).sum() / n
                out[b, s, 1] = (ys * m).sum() / n
            else:
                out[b, s] = 0.5
    return out


def dynamic_posenc(x: Tensor, centroids: Optional[Tensor]) -> Tensor:
    """DynamicPositionalEncoding.forward (models/sppp.py:243-300), dropout p=0."""
    B, L, D = x.shape
    if centroids is None:
        pos = torch.arange(L, dtype=torch.float)[:, None]
        div = torch.exp(torch.arange(0, D, 2, dtype=torch.float) * (-math.log(10000.0) / D))
        pe = torch.zeros(L, D)
        pe[:, 0::2] = torch.sin(pos * div)
        pe[:, 1::2] = torch.cos(pos * div)
        return x + pe[None]
    c = centroids
    if c.shape[1] < L:                                                  # sppp.py:271-274
        c = torch.cat([torch.full((B, 1, 2), 0.5), c], dim=1)
    f = torch.exp(torch.arange(0, D // 2, dtype=torch.float) * (-math.log(10000.0) / (D // 2)))
    pe = torch.cat([torch.sin(c[:, :, 0:1] * f), torch.cos(c[:, :, 1:2] * f)], dim=-1)
    return x + pe


def sppp_vit_mhla_forward(x: Tensor, segmaps: np.ndarray, sd: SD, P: int, H: int, W: int, use_mhla: bool,
                          S: int = 16, kind: str = "mean") -> Tensor:
    """SPPPViTMHLA.forward (models/sppp_mhla.py:264-325) with the label maps given."""
    B, _, img, _ = x.shape
    emb = patch_embed(x, sd, "patch_embed.", P)
    pooled = torch.stack([pool(emb[b], map_patches(segmaps[b], img, P), kind) for b in range(B)])
    t = torch.cat((sd["cls_token"].expand(B, -1, -1), pooled), dim=1)
    t = dynamic_posenc(t, superpixel_centroids(segmaps, S))
    for i in range(_depth(sd)):
        t = vit_mhla_block(t, sd, f"blocks.{i}.", H, W, use_mhla)
    t = layer_norm(t, sd["norm.weight"], sd["norm.bias"])[:, 0]
    return linear(t, sd["head.weight"], sd["head.bias"])


def _sppp_front(x: Tensor, segmaps: np.ndarray, sd: SD, P: int, S: int, kind: str, emb: Tensor) -> Tensor:
    """Steps 3-6 shared by every SPPP model (models/sppp.py:452-484, models/sppp_mhla.py:286-304,
    models/mhla_models.py:226-253, models/attention.py:565-590): pool per image, prepend CLS, centroid pos-enc."""
    B, _, img, _ = x.shape
    pooled = torch.stack([pool(emb[b], map_patches(segmaps[b], img, P), kind) for b in range(B)])
    t = torch.cat((sd["cls_token"].expand(B, -1, -1), pooled), dim=1)
    return dynamic_posenc(t, superpixel_centroids(segmaps, S))


def _head(t: Tensor, sd: SD) -> Tensor:
    return linear(layer_norm(t, sd["norm.weight"], sd["norm.bias"])[:, 0], sd["head.weight"], sd["head.bias"])


def pretrained_vit_mhla_forward(x: Tensor, sd: SD, P: int, H: int, W: int) -> Tensor:
    """PretrainedViTWithMHLA.forward (models/mhla_models.py:120-166): the ViT skeleton over
    MHLATransformerBlock (mlp = nn.Sequential, keys mlp.0 / mlp.3)."""
    B = x.shape[0]
    t = patch_embed(x, sd, "patch_embed.", P)
    t = torch.cat((sd["cls_token"].expand(B, -1, -1), t), dim=1) + sd["pos_embed"]
    for i in range(_depth(sd)):
        t = mhla_block(t, sd, f"blocks.{i}.", H, W)
    return _head(t, sd)


def pretrained_sppp_vit_mhla_forward(x: Tensor, segmaps: np.ndarray, sd: SD, P: int, H: int, W: int, S: int = 16,
                                     kind: str = "mean") -> Tensor:
    """PretrainedSPPPViTWithMHLA.forward (models/mhla_models.py:207-268) with the label maps given."""
    t = _sppp_front(x, segmaps, sd, P, S, kind, patch_embed(x, sd, "patch_embed.", P))
    for i in range(_depth(sd)):
        t = mhla_block(t, sd, f"blocks.{i}.", H, W)
    return _head(t, sd)


def sppp_vit_forward(x: Tensor, segmaps: np.ndarray, sd: SD, P: int, H: int, S: int = 16, kind: str = "mean") -> Tensor:
    """SPPPViT.forward (models/sppp.py:430-500) over vit.TransformerBlock -- the reference's constructor raises
    (sppp.py:378 names a class that does not exist), so this restates the evident intent: NO reference output
    exists for it; every piece it is composed of is pinned on its own."""
    t = _sppp_front(x, segmaps, sd, P, S, kind, patch_embed(x, sd, "patch_embed.", P))
    for i in range(_depth(sd)):
        t = vit_block(t, sd, f"blocks.{i}.", H)
    return _head(t, sd)


def conv_patch_embed(x: Tensor, sd: SD, p: str, P: int) -> Tensor:
    """nn.Conv2d(C, D, kernel = stride = P) -> flatten(2) -> transpose(1, 2) (models/attention.py:271-276,450-455)."""
    return torch.nn.functional.conv2d(x, sd[f"{p}0.weight"], sd[f"{p}0.bias"], stride=P).flatten(2).transpose(1, 2)


def cross_vit_forward(x: Tensor, sd: SD, P: int, H: int, multi_head: bool) -> Tensor:
    """CrossAttentionViT.forward (models/attention.py:325-372; block(x, x), :350).  Constructor raises in the
    reference (nn.Transpose, attention.py:275): evident intent, no reference output."""
    B = x.shape[0]
    t = conv_patch_embed(x, sd, "patch_embed.", P)
    t = torch.cat((sd["cls_token"].expand(B, -1, -1), t), dim=1) + sd["pos_embed"]
    for i in range(_depth(sd)):
        t = cross_block(t, t, sd, f"blocks.{i}.", H, multi_head)
    return _head(t, sd)


def cross_sppp_vit_forward(x: Tensor, segmaps: np.ndarray, sd: SD, P: int, H: int, multi_head: bool, S: int = 16,
                           kind: str = "mean") -> Tensor:
    """CrossAttentionSPPPViT.forward (models/attention.py:540-609); same caveat as cross_vit_forward."""
    t = _sppp_front(x, segmaps, sd, P, S, kind, conv_patch_embed(x, sd, "patch_embed.", P))
    for i in range(_depth(sd)):
        t = cross_block(t, t, sd, f"blocks.{i}.", H, multi_head)
    return _head(t, sd)


def cross_entropy(logits: Tensor, labels: Tensor) -> Tensor:
    """nn.CrossEntropyLoss() mean reduction (experiments/mhla_pretrained.py:329,365)."""
    lse = torch.logsumexp(logits, dim=-1)
    return (lse - logits.gather(1, labels[:, None]).squeeze(1)).mean()


# --------------------------------------------------------------------------- #
# synthetic label maps (inputs for the SPPP path; SLIC itself is out of scope)
# --------------------------------------------------------------------------- #
def voronoi_labels(img: int, n_regions: int, seed: int, grid_jitter: float = 0.25) -> np.ndarray:
    """Seeded jittered-Voronoi label map [img,img] int64 with labels 0..n_regions-1
    (cfg3 inputs).  Seeds sit on a jittered sqrt(n) x sqrt(n) grid so every
    region dominates at least one 16x16 patch at img=224."""
    rng = np.random.RandomState(seed)
    g = int(math.ceil(math.sqrt(n_regions)))
    cell = img / g
    pts = []
    for r in range(g):
        for c in range(g):
            if len(pts) < n_regions:
                pts.append(((r + 0.5 + rng.uniform(-grid_jitter, grid_jitter)) * cell,
                            (c + 0.5 + rng.uniform(-grid_jitter, grid_jitter)) * cell))
    pts = np.asarray(pts, dtype=np.float32)
    yy, xx = np.meshgrid(np.arange(img, dtype=np.float32), np.arange(img, dtype=np.float32), indexing="ij")
    d = (yy[..., None] - pts[:, 0]) ** 2 + (xx[..., None] - pts[:, 1]) ** 2
    return d.argmin(axis=-1).astype(np.int64)
