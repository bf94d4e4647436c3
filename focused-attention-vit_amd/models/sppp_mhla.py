"""Mirror of the reference's models/sppp_mhla.py (TransformerBlock, SPPPViTMHLA)."""
import torch
import torch.nn as nn

from ._backend import F
from .vit import PatchEmbedding, run_encoder
from .vit_mhla import TransformerBlock          # byte-identical logic in the reference (sppp_mhla.py:21-110)
from .sppp import (SuperpixelSegmentation, PatchToSuperpixelMapper, SuperpixelPooling, DynamicPositionalEncoding,
                   calculate_superpixel_centroids, sppp_tokens)

__all__ = ["TransformerBlock", "SPPPViTMHLA"]


class SPPPViTMHLA(nn.Module):
    """reference models/sppp_mhla.py:113-334.  No ``pos_embed`` Parameter (DynamicPositionalEncoding
    instead).  ``assume_num_tokens`` (optional int attribute) skips the per-forward host check that
    every image of the batch produced the same number of superpixel tokens."""

    def __init__(self, img_size: int = 224, patch_size: int = 4, in_channels: int = 3, num_classes: int = 1000,
                 embed_dim: int = 768, depth: int = 12, num_heads: int = 12, mlp_ratio: float = 4.0,
                 dropout: float = 0.0, attn_dropout: float = 0.0, embed_dropout: float = 0.0,
                 num_superpixels: int = 16, compactness: float = 0.1, pooling_type: str = 'mean',
                 window_size: int = 7, use_mhla: bool = False):
        super().__init__()
        self.img_size = img_size
        self.patch_size = patch_size
        self.in_channels = in_channels
        self.num_classes = num_classes
        self.embed_dim = embed_dim
        self.depth = depth
        self.num_heads = num_heads
        self.num_superpixels = num_superpixels
        self.use_mhla = use_mhla
        self.segmentation = SuperpixelSegmentation(num_segments=num_superpixels, compactness=compactness)
        self.patch_mapper = PatchToSuperpixelMapper(patch_size=patch_size)
        self.pooling = SuperpixelPooling(pooling_type=pooling_type)
        self.patch_embed = PatchEmbedding(img_size=img_size, patch_size=patch_size, in_channels=in_channels,
                                          embed_dim=embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = DynamicPositionalEncoding(embed_dim, embed_dropout)
        self.blocks = nn.ModuleList([
            TransformerBlock(embed_dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, dropout=dropout,
                             attn_dropout=attn_dropout, window_size=window_size, use_mhla=use_mhla)
            for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim)
        self.head = nn.Linear(embed_dim, num_classes)
        self.assume_num_tokens = None
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
        t = run_encoder(self.blocks, t, None, self.training)       # blocks get no mask (sppp_mhla.py:313-314)
        t = F.run(F.FinalNormOp(), [t], [self.norm.weight, self.norm.bias])
        return F.run(F.LinearOp(), [t], [self.head.weight, self.head.bias])

    def get_num_parameters(self) -> int:
        return sum(p.numel() for p in self.parameters() if p.requires_grad)
