"""Mirror of the reference's models/mhla.py (MultiHeadLatentAttention, MHLATransformerBlock)."""
from typing import Optional

import torch
import torch.nn as nn

from ._backend import F, params


class MultiHeadLatentAttention(nn.Module):
    """Window-based "latent" attention, reference models/mhla.py:17-161.  ``latent_proj`` stays an
    ordinary nn.Linear(head_dim, head_dim) parameter (callers eye_-init it and give it its own LR
    group, experiments/mhla_pretrained.py:224,324); the kernels fold it into the qkv projection."""

    def __init__(self, embed_dim: int, num_heads: int, window_size: int = 7, dropout: float = 0.0):
        super().__init__()
        self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.window_size = window_size
        self.head_dim = embed_dim // num_heads
        assert self.head_dim * num_heads == embed_dim, "embed_dim must be divisible by num_heads"
        self.qkv = nn.Linear(embed_dim, embed_dim * 3)
        self.proj = nn.Linear(embed_dim, embed_dim)
        self.latent_proj = nn.Linear(self.head_dim, self.head_dim)
        self.attn_dropout = nn.Dropout(dropout)
        self.proj_dropout = nn.Dropout(dropout)

    def _get_window_indices(self, seq_len: int) -> torch.Tensor:
        """[seq_len, window_size] int64 table with the reference's padding rule
        (models/mhla.py:46-83); the kernels evaluate the same closed form on the fly."""
        W, h = self.window_size, self.window_size // 2
        if W % 2 == 0:
            raise ValueError("window_size must be odd: the reference crashes on even sizes (models/mhla.py:83)")
        rows = []
        for i in range(seq_len):
            lo, hi = max(0, i - h), min(seq_len, i + h + 1)
            win = list(range(lo, hi))
            pad = W - len(win)
            if pad > 0:
                win = win + [seq_len - 1] * pad if lo == 0 else [0] * pad + win
            rows.append(win)
        return torch.tensor(rows, dtype=torch.long)

    def _chain(self):
        return F.MHLAChain(self.num_heads, self.window_size, self.attn_dropout.p, self.proj_dropout.p)

    def forward(self, x: torch.Tensor, attention_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        ch = self._chain()
        return F.run(F.AttnOp(ch, F._mask_u8(attention_mask), self.training), [x], params(self, ch.names))


class MHLATransformerBlock(nn.Module):
    """reference models/mhla.py:164-222 (mlp is an nn.Sequential with Linear at index 0 and 3)"""

    def __init__(self, embed_dim: int, num_heads: int, window_size: int = 7, mlp_ratio: float = 4.0,
                 dropout: float = 0.0, attn_dropout: float = 0.0):
        super().__init__()
        self.norm1 = nn.LayerNorm(embed_dim)
        self.attn = MultiHeadLatentAttention(embed_dim=embed_dim, num_heads=num_heads, window_size=window_size,
                                             dropout=attn_dropout)
        self.norm2 = nn.LayerNorm(embed_dim)
        mlp_hidden_dim = int(embed_dim * mlp_ratio)
        self.mlp = nn.Sequential(nn.Linear(embed_dim, mlp_hidden_dim), nn.GELU(), nn.Dropout(dropout),
                                 nn.Linear(mlp_hidden_dim, embed_dim), nn.Dropout(dropout))

    def _spec(self):
        return F.BlockSpec(self.attn._chain(), F.MLPChain(self.mlp[2].p, "0", "3"))

    def forward(self, x: torch.Tensor, attention_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        spec = self._spec()
        return F.run(F.EncoderOp([spec], F._mask_u8(attention_mask), self.training), [x], params(self, spec.names))
