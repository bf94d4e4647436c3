"""Mirror of the reference's models/vit_mhla.py (TransformerBlock, VisionTransformerMHLA)."""
from typing import Optional

import torch
import torch.nn as nn

from ._backend import F, params
from .vit import PatchEmbedding, MLP, embed_dropout, run_encoder
from .mhla import MultiHeadLatentAttention


class TransformerBlock(nn.Module):
    """Encoder block with MHLA or nn.MultiheadAttention, reference models/vit_mhla.py:20-109."""

    def __init__(self, embed_dim: int, num_heads: int, mlp_ratio: float = 4.0, dropout: float = 0.0,
                 attn_dropout: float = 0.0, window_size: int = 7, use_mhla: bool = False):
        super().__init__()
        self.norm1 = nn.LayerNorm(embed_dim)
        if use_mhla:
            self.attn = MultiHeadLatentAttention(embed_dim=embed_dim, num_heads=num_heads, window_size=window_size,
                                                 dropout=attn_dropout)
        else:
            # parameter container only (same names / init as the reference); the math runs in DenseChain
            self.attn = nn.MultiheadAttention(embed_dim=embed_dim, num_heads=num_heads, dropout=attn_dropout,
                                              batch_first=True)
        self.norm2 = nn.LayerNorm(embed_dim)
        self.mlp = MLP(in_features=embed_dim, hidden_features=int(embed_dim * mlp_ratio), out_features=embed_dim,
                       dropout=dropout)
        self.use_mhla = use_mhla
        self._num_heads = num_heads

    def _spec(self):
        if self.use_mhla:
            attn = self.attn._chain()
        else:
            attn = F.DenseChain(self._num_heads, self.attn.dropout, 0.0, torch_mha=True)
        return F.BlockSpec(attn, self.mlp._chain())

    def forward(self, x: torch.Tensor, attention_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        spec = self._spec()
        return F.run(F.EncoderOp([spec], F._mask_u8(attention_mask), self.training), [x], params(self, spec.names))


class VisionTransformerMHLA(nn.Module):
    """reference models/vit_mhla.py:112-268"""

    def __init__(self, img_size: int = 224, patch_size: int = 4, in_channels: int = 3, num_classes: int = 1000,
                 embed_dim: int = 768, depth: int = 12, num_heads: int = 12, mlp_ratio: float = 4.0,
                 dropout: float = 0.0, attn_dropout: float = 0.0, embed_dropout: float = 0.0, window_size: int = 7,
                 use_mhla: bool = False):
        super().__init__()
        self.img_size = img_size
        self.patch_size = patch_size
        self.in_channels = in_channels
        self.num_classes = num_classes
        self.embed_dim = embed_dim
        self.depth = depth
        self.num_heads = num_heads
        self.use_mhla = use_mhla
        self.patch_embed = PatchEmbedding(img_size=img_size, patch_size=patch_size, in_channels=in_channels,
                                          embed_dim=embed_dim)
        num_patches = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, embed_dim))
        self.pos_drop = nn.Dropout(embed_dropout)
        self.blocks = nn.ModuleList([
            TransformerBlock(embed_dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, dropout=dropout,
                             attn_dropout=attn_dropout, window_size=window_size, use_mhla=use_mhla)
            for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim)
        self.head = nn.Linear(embed_dim, num_classes)
        self._init_weights()

    def _init_weights(self):
        nn.init.normal_(self.cls_token, std=0.02)
        nn.init.normal_(self.pos_embed, std=0.02)
        self.apply(self._init_weights_recursive)

    def _init_weights_recursive(self, m):
        if isinstance(m, nn.Linear):
            nn.init.normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, nn.LayerNorm):
            nn.init.ones_(m.weight)
            nn.init.zeros_(m.bias)

    def forward_features(self, x: torch.Tensor) -> torch.Tensor:
        tok = self.patch_embed(x)
        x = F.run(F.PrologueOp(True), [tok], [self.cls_token, self.pos_embed])
        x = embed_dropout(x, self.pos_drop.p, self.training)
        x = run_encoder(self.blocks, x, None, self.training)
        return F.run(F.FinalNormOp(), [x], [self.norm.weight, self.norm.bias])

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = self.forward_features(x)
        return F.run(F.LinearOp(), [x], [self.head.weight, self.head.bias])

    def get_num_parameters(self) -> int:
        return sum(p.numel() for p in self.parameters() if p.requires_grad)
