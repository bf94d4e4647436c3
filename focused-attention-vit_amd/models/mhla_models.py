"""Mirror of the reference's models/mhla_models.py (PretrainedViTWithMHLA,
PretrainedSPPPViTWithMHLA): same structure as vit_mhla / sppp_mhla but the blocks are
MHLATransformerBlock.  NOTE: the reference default ``window_size=4`` (mhla_models.py:49,208) is
even and crashes the reference at the first forward; here an even window raises a clear
ValueError at forward time."""
import torch
import torch.nn as nn

from ._backend import F
from .vit import PatchEmbedding, embed_dropout, run_encoder
from .sppp import (SuperpixelSegmentation, PatchToSuperpixelMapper, SuperpixelPooling, DynamicPositionalEncoding,
                   calculate_superpixel_centroids, sppp_tokens)
from .mhla import MHLATransformerBlock


def _init_recursive(m):
    if isinstance(m, nn.Linear):
        nn.init.normal_(m.weight, std=0.02)
        if m.bias is not None:
            nn.init.zeros_(m.bias)
    elif isinstance(m, nn.LayerNorm):
        nn.init.ones_(m.weight)
        nn.init.zeros_(m.bias)


class PretrainedViTWithMHLA(nn.Module):
    """reference models/mhla_models.py:22-175"""

    def __init__(self, img_size: int = 224, patch_size: int = 4, in_channels: int = 3, num_classes: int = 1000,
                 embed_dim: int = 768, depth: int = 12, num_heads: int = 12, window_size: int = 4,
                 mlp_ratio: float = 4.0, dropout: float = 0.0, attn_dropout: float = 0.0,
                 embed_dropout: float = 0.0):
        super().__init__()
        self.img_size = img_size
        self.patch_size = patch_size
        self.in_channels = in_channels
        self.num_classes = num_classes
        self.embed_dim = embed_dim
        self.depth = depth
        self.num_heads = num_heads
        self.window_size = window_size
        self.patch_embed = PatchEmbedding(img_size=img_size, patch_size=patch_size, in_channels=in_channels,
                                          embed_dim=embed_dim)
        num_patches = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, embed_dim))
        self.pos_drop = nn.Dropout(embed_dropout)
        self.blocks = nn.ModuleList([
            MHLATransformerBlock(embed_dim=embed_dim, num_heads=num_heads, window_size=window_size,
                                 mlp_ratio=mlp_ratio, dropout=dropout, attn_dropout=attn_dropout)
            for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim)
        self.head = nn.Linear(embed_dim, num_classes)
        nn.init.normal_(self.cls_token, std=0.02)
        nn.init.normal_(self.pos_embed, std=0.02)
        self.apply(_init_recursive)

    def forward_features(self, x: torch.Tensor) -> torch.Tensor:
        tok = self.patch_embed(x)
        x = F.run(F.PrologueOp(True), [tok], [self.cls_token, self.pos_embed])
        x = embed_dropout(x, self.pos_drop.p, self.training)
        x = run_encoder(self.blocks, x, None, self.training)
        return F.run(F.FinalNormOp(), [x], [self.norm.weight, self.norm.bias])

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return F.run(F.LinearOp(), [self.forward_features(x)], [self.head.weight, self.head.bias])

    def get_num_parameters(self) -> int:
        return sum(p.numel() for p in self.parameters() if p.requires_grad)


class PretrainedSPPPViTWithMHLA(nn.Module):
    """reference models/mhla_models.py:178-395"""

    def __init__(self, img_size: int = 224, patch_size: int = 4, in_channels: int = 3, num_classes: int = 1000,
                 embed_dim: int = 768, depth: int = 12, num_heads: int = 12, window_size: int = 4,
                 mlp_ratio: float = 4.0, dropout: float = 0.0, attn_dropout: float = 0.0,
                 embed_dropout: float = 0.0, num_superpixels: int = 16, compactness: float = 0.1,
                 pooling_type: str = 'mean'):
        super().__init__()
        self.img_size = img_size
        self.patch_size = patch_size
        self.in_channels = in_channels
        self.num_classes = num_classes
        self.embed_dim = embed_dim
        self.depth = depth
        self.num_heads = num_heads
        self.window_size = window_size
        self.num_superpixels = num_superpixels
        self.segmentation = SuperpixelSegmentation(num_segments=num_superpixels, compactness=compactness)
        self.patch_mapper = PatchToSuperpixelMapper(patch_size=patch_size)
        self.pooling = SuperpixelPooling(pooling_type=pooling_type)
        self.patch_embed = PatchEmbedding(img_size=img_size, patch_size=patch_size, in_channels=in_channels,
                                          embed_dim=embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = DynamicPositionalEncoding(embed_dim, embed_dropout)
        self.blocks = nn.ModuleList([
            MHLATransformerBlock(embed_dim=embed_dim, num_heads=num_heads, window_size=window_size,
                                 mlp_ratio=mlp_ratio, dropout=dropout, attn_dropout=attn_dropout)
            for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim)
        self.head = nn.Linear(embed_dim, num_classes)
        self.assume_num_tokens = None
        nn.init.normal_(self.cls_token, std=0.02)
        self.apply(_init_recursive)

    def _calculate_superpixel_centroids(self, segmentation_maps):
        return calculate_superpixel_centroids(self, segmentation_maps)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        t = sppp_tokens(self, x)
        t = run_encoder(self.blocks, t, None, self.training)
        t = F.run(F.FinalNormOp(), [t], [self.norm.weight, self.norm.bias])
        return F.run(F.LinearOp(), [t], [self.head.weight, self.head.bias])

    def get_num_parameters(self) -> int:
        return sum(p.numel() for p in self.parameters() if p.requires_grad)
