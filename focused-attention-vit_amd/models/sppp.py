"""Mirror of the reference's models/sppp.py (SuperpixelSegmentation, PatchToSuperpixelMapper,
SuperpixelPooling, DynamicPositionalEncoding, SPPPViT).

The reference runs mapping / pooling / centroids as per-image, per-patch Python loops with a
device sync per patch; here they are batched device kernels (csrc/sppp.hip).  The dict-based
per-image API of the reference is kept (``map_patches`` returns the same ordered dict) and a
batched device API is added for the model fast path.  ``SuperpixelSegmentation.segment`` runs SLIC on the
device for GPU images (csrc/slic.hip: the published algorithm as scikit-image >= 0.19 parametrises it, including
its per-image min-max rescale of the input; parity with scikit-image itself is UNPINNED -- it is not importable here
and the reference holds no label-map fixture), returns installed label maps if there are any
(``model.segmentation.set_label_maps(...)``, the way to reproduce segmentations computed elsewhere), and calls
scikit-image for CPU images like the reference does."""
import math
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from ._backend import F, K, params
from .vit import PatchEmbedding, TransformerBlock as _VitBlock, run_encoder


class _MapState:
    """Installed label maps + what depends on them alone: the patch -> superpixel mapping (rank, token counts, permutation,
    offsets: `PatchToSuperpixelMapper.map_patches_batched`) and the superpixel centroids, per (patch size, superpixel
    count) that a model asked for.  The reference recomputes both inside every forward
    (models/sppp_mhla.py:287-291, 307); neither depends on a weight, so here they are computed once per INSTALL of the
    maps (by data.DeviceLoader on its preparation stream, under the previous step) instead of inside every step
    (three launches, ~68 us of a 2.4-ms cfg3 step).  `version` is the maps tensor's version the derived tensors belong
    to: an in-place edit of the maps that bypasses update_label_maps is noticed by the next eager forward and by
    train.GraphedStep before a replay, and the derived tensors are refreshed IN PLACE (a captured graph holds their
    addresses)."""

    def __init__(self, maps: torch.Tensor, derived=None):
        self.maps = maps
        self.derived = {k: list(v) for k, v in (derived or {}).items()}
        self.version = maps._version

    @staticmethod
    def derive(maps: torch.Tensor, P: int, S: int):
        rank, ntok, perm, offs, _ = K.sppp_map_patches(maps, P)
        return [rank, ntok, perm, offs, K.sppp_centroids(maps, S)]

    def refresh_if_stale(self, derived=None):
        if self.maps._version == self.version and derived is None:
            return
        for key, dst in self.derived.items():
            src = derived.get(key) if derived else None
            if src is None:
                src = self.derive(self.maps, *key)
            for d, s_ in zip(dst, src):
                d.copy_(s_, non_blocking=True)
        self.version = self.maps._version

    def get(self, P: int, S: int):
        key = (int(P), int(S))
        if key not in self.derived:
            if self.maps._version != self.version:
                self.refresh_if_stale()
            self.derived[key] = self.derive(self.maps, *key)
        elif self.maps._version != self.version:
            self.refresh_if_stale()
        return self.derived[key]


class SuperpixelSegmentation:
    """reference models/sppp.py:26-74"""

    def __init__(self, num_segments: int = 16, compactness: float = 0.1, sigma: float = 1.0):
        self.num_segments = num_segments
        self.compactness = compactness
        self.sigma = sigma
        self._state: Optional[_MapState] = None
        self._keys = set()             # (patch size, superpixel count) pairs models have asked the derived tensors for
        self._captured = False         # a train.GraphedStep has captured the ADDRESS of the installed maps
        self.rescale_input = True      # scikit-image >= 0.19 rescales every image to [0, 1] first; False = < 0.19

    @property
    def _maps(self) -> Optional[torch.Tensor]:
        return None if self._state is None else self._state.maps

    def set_label_maps(self, maps: Optional[torch.Tensor], derived=None):
        """Install precomputed label maps [B,H,W] (int64) returned by the next segment() calls.  A train.GraphedStep
        captured earlier keeps reading (and keeps alive) the tensors that were installed at ITS capture: to change what
        such a step sees, use update_label_maps.  derived: what `derive_for(maps)` returned (data.DeviceLoader computes
        it on its own stream); otherwise the first forward computes it."""
        self._state = None if maps is None else _MapState(maps, derived)

    def update_label_maps(self, maps: torch.Tensor, derived=None):
        """Copy new label maps INTO the installed tensor (same shape): the form a captured training step needs, whose
        replayed kernels read the installed tensor's address (data.DeviceLoader does this per batch once a step has
        been captured).  The tensors derived from the maps are refreshed in place as well (from `derived` if given)."""
        st = self._state
        if st is None or tuple(maps.shape) != tuple(st.maps.shape) or maps.device != st.maps.device:
            raise RuntimeError("update_label_maps: install maps of this shape with set_label_maps first")
        if maps is not st.maps:
            st.maps.copy_(maps.to(st.maps.dtype), non_blocking=True)
        st.refresh_if_stale(derived if derived else None)

    def derive_for(self, maps: torch.Tensor) -> dict:
        """The patch mapping / centroid tensors of `maps` for every (patch size, superpixel count) a model has asked this
        object for so far, on the current stream (data.DeviceLoader: its preparation stream)."""
        return {k: _MapState.derive(maps, *k) for k in sorted(self._keys)}

    def segment_device(self, images: torch.Tensor) -> torch.Tensor:
        """Device SLIC of a batch [B,3,H,W] on the current stream, ignoring installed maps (data.DeviceLoader calls it
        for the NEXT batch on its preparation stream and installs the result when that batch is yielded)."""
        return K.slic(images.float(), n_segments=self.num_segments, compactness=self.compactness, sigma=self.sigma,
                      rescale=self.rescale_input)

    def segment(self, image: torch.Tensor) -> torch.Tensor:
        """Label maps [B,H,W] (or [H,W]) int64.  Installed maps win; images on the GPU run the device SLIC
        (csrc/slic.hip: no D2H -> skimage -> H2D hop per image; parity with scikit-image is unpinned, the
        algorithm and its checked properties are in DESIGN.md); CPU images go to scikit-image like the reference."""
        if self._maps is not None:
            maps = self._maps
            return maps if image.dim() == 4 else maps[0]
        batch_mode = image.dim() == 4
        imgs = image if batch_mode else image[None]
        if imgs.is_cuda:
            seg = K.slic(imgs.float(), n_segments=self.num_segments, compactness=self.compactness, sigma=self.sigma,
                         rescale=self.rescale_input)
            return seg if batch_mode else seg[0]
        try:
            from skimage.segmentation import slic
        except ImportError as e:
            raise ImportError(
                "SuperpixelSegmentation.segment on CPU tensors needs scikit-image's SLIC (third-party, not in this "
                "image); move the images to the GPU (device SLIC) or provide label maps via set_label_maps().") from e
        out = []
        for i in range(imgs.shape[0]):
            img = imgs[i].permute(1, 2, 0).cpu().numpy()
            seg = slic(img, n_segments=self.num_segments, compactness=self.compactness, sigma=self.sigma,
                       start_label=0)
            out.append(torch.from_numpy(seg).to(image.device))
        return torch.stack(out) if batch_mode else out[0]


class PatchToSuperpixelMapper:
    """reference models/sppp.py:77-128"""

    def __init__(self, patch_size: int):
        self.patch_size = patch_size

    def map_patches_batched(self, segmentation_maps: torch.Tensor):
        """Device API: label maps [B,H,W] -> (patch_rank[B,N], n_tokens[B], perm[B,N], offs[B,N+1], dom[B,N])."""
        return K.sppp_map_patches(segmentation_maps, self.patch_size)

    def map_patches(self, segmentation_map: torch.Tensor, img_size: int) -> Dict[int, List[int]]:
        """Reference API: ordered dict {dominant label: [patch idx]} in first-seen raster order."""
        rank, ntok, perm, offs, dom = K.sppp_map_patches(segmentation_map[None], self.patch_size)
        rank, dom = rank[0].tolist(), dom[0].tolist()
        out: Dict[int, List[int]] = {}
        order = sorted(set(rank))
        keys = {}
        for p, r in enumerate(rank):
            keys.setdefault(r, dom[p])
        for r in order:
            out[int(keys[r])] = [p for p, rr in enumerate(rank) if rr == r]
        return out


class SuperpixelPooling:
    """reference models/sppp.py:131-223"""

    def __init__(self, pooling_type: str = 'mean'):
        self.pooling_type = pooling_type

    def _kind(self):
        try:
            return {"mean": 0, "max": 1, "attention": 2}[self.pooling_type]
        except KeyError:
            raise ValueError(f"Unsupported pooling type: {self.pooling_type}") from None

    def pool_batched(self, patch_embeddings: torch.Tensor, perm, offs, R: int) -> torch.Tensor:
        return F.run(F.PoolOp(self._kind(), perm, offs, R), [patch_embeddings], [])

    def pool(self, patch_embeddings: torch.Tensor, superpixel_to_patches: Dict[int, List[int]]) -> torch.Tensor:
        """Reference API ([N,D] or [B,N,D] embeddings + one mapping dict shared by the batch)."""
        kind = self._kind()
        squeeze = patch_embeddings.dim() == 2
        emb = patch_embeddings[None] if squeeze else patch_embeddings
        B, N, D = emb.shape
        groups = list(superpixel_to_patches.values())
        R = len(groups)
        perm_l, offs_l = [], [0]
        for g in groups:
            perm_l += list(g)
            offs_l.append(len(perm_l))
        perm_l += [0] * (N - len(perm_l))
        offs_l += [offs_l[-1]] * (N + 1 - len(offs_l))
        dev = emb.device
        perm = torch.tensor(perm_l, dtype=torch.int32, device=dev)[None].expand(B, N).contiguous()
        offs = torch.tensor(offs_l, dtype=torch.int32, device=dev)[None].expand(B, N + 1).contiguous()
        out = F.run(F.PoolOp(kind, perm, offs, R), [emb], [])
        return out[0] if squeeze else out


class DynamicPositionalEncoding(nn.Module):
    """reference models/sppp.py:226-300"""

    def __init__(self, embed_dim: int, dropout: float = 0.0):
        super().__init__()
        self.embed_dim = embed_dim
        self.dropout = nn.Dropout(dropout)

    def forward(self, x: torch.Tensor, superpixel_centroids: Optional[torch.Tensor] = None) -> torch.Tensor:
        if superpixel_centroids is not None:
            L, n = x.shape[1], superpixel_centroids.shape[1]
            if n != L and n != L - 1:
                raise ValueError(f"centroids for {n} superpixels cannot be broadcast onto {L} tokens "
                                 "(the reference fails here with a broadcast error, models/sppp.py:299)")
            cent = superpixel_centroids.to(torch.float32).contiguous()
        else:
            cent = None
        x = F.run(F.PosEncOp(cent), [x], [])
        if self.training and self.dropout.p > 0:
            x = F.run(F.DropoutOp(self.dropout.p), [x], [])
        return x


def sppp_tokens(model, x: torch.Tensor) -> torch.Tensor:
    """Shared front end of the SPPP models (models/sppp_mhla.py:274-310): label maps -> patch
    embedding -> device-side patch->superpixel mapping + pooling -> CLS -> centroid pos-enc."""
    sg = model.segmentation
    seg = sg.segment(x)
    if seg.device != x.device:
        seg = seg.to(x.device)
    tok = model.patch_embed(x)
    st = sg._state
    if st is not None and seg is st.maps and x.dim() == 4:          # installed maps: mapping + centroids once per install
        rank, ntok, perm, offs, cent_pre = st.get(model.patch_mapper.patch_size, model.num_superpixels)
        sg._keys.add((int(model.patch_mapper.patch_size), int(model.num_superpixels)))
    else:
        rank, ntok, perm, offs, _ = model.patch_mapper.map_patches_batched(seg)
        cent_pre = None
    R = getattr(model, "assume_num_tokens", None)
    if R is None:
        lo, hi = int(ntok.min().item()), int(ntok.max().item())      # one host sync; skip via assume_num_tokens
        if lo != hi:
            raise ValueError(f"images in the batch produced different superpixel-token counts ({lo}..{hi}); the "
                             "reference fails here in torch.stack (models/sppp_mhla.py:300): bucket by count")
        R = lo
    pooled = model.pooling.pool_batched(tok, perm, offs, R)
    t = F.run(F.PrologueOp(False), [pooled], [model.cls_token])
    cent = cent_pre if cent_pre is not None else K.sppp_centroids(seg, model.num_superpixels)
    return model.pos_embed(t, cent)


class TokenBucketed(nn.Module):
    """Runs an SPPP model on batches whose images segment into DIFFERENT numbers of superpixel tokens.

    The reference stacks the per-image token lists (``torch.stack``, models/sppp_mhla.py:300; models/sppp.py:446), so a
    batch trains there only when every image happens to yield the same count -- SLIC on photographs does not promise
    that.  (Its positional encoding, models/sppp.py:299, accepts exactly two counts per model: ``num_superpixels`` and
    ``num_superpixels - 1``; other counts fail there for a batch of one as well, and they do here.)
    This wrapper (an extension: nothing in the reference corresponds to it) groups the images of a batch by count,
    runs the wrapped model once per group on that group's images and label maps, and returns the logits in the batch's
    order; the pieces are joined with autograd-visible ops, so ONE ``loss.backward()`` on the whole batch's loss gives
    the wrapped model's parameters exactly the gradients of that loss.  Per image the result is what the wrapped model
    (and the reference) gives for that image alone.  One host sync per forward (the counts); eager steps only --
    ``train.GraphedStep`` needs fixed shapes (bench.py's cfg5 captures one step per count instead).

    ``parameters()``, ``state_dict()`` (prefix ``model.``), ``train()`` / ``eval()`` behave as for any module; hand the
    WRAPPED model to ``train.param_groups`` / checkpoints if the reference's key names matter."""

    def __init__(self, model: nn.Module):
        super().__init__()
        if not hasattr(model, "segmentation") or not hasattr(model, "patch_mapper"):
            raise TypeError("TokenBucketed wraps the SPPP models (SPPPViT, SPPPViTMHLA, ...)")
        self.model = model

    @property
    def segmentation(self):
        return self.model.segmentation

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        m = self.model
        seg_host = m.segmentation
        maps = seg_host.segment(x)
        if maps.device != x.device:
            maps = maps.to(x.device)
        counts = m.patch_mapper.map_patches_batched(maps)[1].tolist()           # the host sync
        groups = {}
        for i, c in enumerate(counts):
            groups.setdefault(int(c), []).append(i)
        installed, assumed = seg_host._state, getattr(m, "assume_num_tokens", None)
        if seg_host._captured:
            raise RuntimeError("TokenBucketed: the wrapped model's label maps are captured by a GraphedStep")
        try:
            if len(groups) == 1:
                seg_host._state, m.assume_num_tokens = _MapState(maps), next(iter(groups))
                return m(x)
            order, outs = [], []
            for c in sorted(groups):
                idx = torch.tensor(groups[c], device=x.device, dtype=torch.int64)
                seg_host._state, m.assume_num_tokens = _MapState(maps.index_select(0, idx)), c
                outs.append(m(x.index_select(0, idx)))
                order += groups[c]
            inv = torch.empty(len(order), dtype=torch.int64)
            inv[torch.tensor(order)] = torch.arange(len(order))
            return torch.cat(outs, 0).index_select(0, inv.to(x.device))
        finally:
            seg_host._state, m.assume_num_tokens = installed, assumed


def calculate_superpixel_centroids(model, segmentation_maps: torch.Tensor) -> torch.Tensor:
    return K.sppp_centroids(segmentation_maps, model.num_superpixels)


class SPPPViT(nn.Module):
    """reference models/sppp.py:303-520.  The reference constructor raises
    (``VisionTransformer.TransformerBlock``, sppp.py:378); this keeps the signature and implements
    the evident intent with vit.TransformerBlock."""

    def __init__(self, img_size: int = 224, patch_size: int = 4, in_channels: int = 3, num_classes: int = 1000,
                 embed_dim: int = 768, depth: int = 12, num_heads: int = 12, mlp_ratio: float = 4.0,
                 dropout: float = 0.0, attn_dropout: float = 0.0, embed_dropout: float = 0.0,
                 num_superpixels: int = 16, compactness: float = 0.1, pooling_type: str = 'mean'):
        super().__init__()
        self.img_size = img_size
        self.patch_size = patch_size
        self.in_channels = in_channels
        self.num_classes = num_classes
        self.embed_dim = embed_dim
        self.depth = depth
        self.num_heads = num_heads
        self.num_superpixels = num_superpixels
        self.segmentation = SuperpixelSegmentation(num_segments=num_superpixels, compactness=compactness)
        self.patch_mapper = PatchToSuperpixelMapper(patch_size=patch_size)
        self.pooling = SuperpixelPooling(pooling_type=pooling_type)
        self.patch_embed = PatchEmbedding(img_size=img_size, patch_size=patch_size, in_channels=in_channels,
                                          embed_dim=embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = DynamicPositionalEncoding(embed_dim, embed_dropout)
        self.blocks = nn.ModuleList([
            _VitBlock(embed_dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, dropout=dropout,
                      attn_dropout=attn_dropout) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim)
        self.head = nn.Linear(embed_dim, num_classes)
        self._init_weights()

    def _init_weights(self):
        nn.init.normal_(self.cls_token, std=0.02)
        self.apply(self._init_weights_recursive)

    def _init_weights_recursive(self, m):
        if isinstance(m, nn.Linear):
            nn.init.normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, nn.LayerNorm):
            nn.init.ones_(m.weight)
            nn.init.zeros_(m.bias)

    def _calculate_superpixel_centroids(self, segmentation_maps):
        return calculate_superpixel_centroids(self, segmentation_maps)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        t = sppp_tokens(self, x)
        t = run_encoder(self.blocks, t, None, self.training)
        t = F.run(F.FinalNormOp(), [t], [self.norm.weight, self.norm.bias])
        return F.run(F.LinearOp(), [t], [self.head.weight, self.head.bias])

    def get_num_parameters(self) -> int:
        return sum(p.numel() for p in self.parameters() if p.requires_grad)
