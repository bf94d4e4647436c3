"""Mirror of the reference's models/vit.py (PatchEmbedding, MultiHeadAttention, MLP,
TransformerBlock, VisionTransformer): identical constructors, forward signatures,
attribute names, state_dict keys and weight-init RNG order; the math runs in the HIP
kernels of libfavit (no aten compute ops on the path)."""
import torch
import torch.nn as nn

from ._backend import F, params


class PatchRearrange(nn.Module):
    """Parameter-free placeholder for einops' Rearrange('b c (h p1) (w p2) -> b (h w) (p1 p2 c)')
    at index 0 of ``PatchEmbedding.projection`` (reference models/vit.py:38-39); the gather is
    fused into favit_patchify_fwd."""

    def __init__(self, patch_size):
        super().__init__()
        self.patch_size = patch_size

    def extra_repr(self):
        return f"'b c (h p1) (w p2) -> b (h w) (p1 p2 c)', p1=p2={self.patch_size}"

    def forward(self, x):  # pragma: no cover - the fused path never calls this
        raise RuntimeError("PatchRearrange is fused into PatchEmbedding.forward")


class PatchEmbedding(nn.Module):
    """reference models/vit.py:19-53"""

    def __init__(self, img_size=224, patch_size=16, in_channels=3, embed_dim=768):
        super().__init__()
        self.img_size = img_size
        self.patch_size = patch_size
        self.num_patches = (img_size // patch_size) ** 2
        self.projection = nn.Sequential(
            PatchRearrange(patch_size),
            nn.Linear(patch_size * patch_size * in_channels, embed_dim),
        )

    def forward(self, x):
        lin = self.projection[1]
        return F.run(F.PatchEmbedOp(self.patch_size), [x], [lin.weight, lin.bias])


class MultiHeadAttention(nn.Module):
    """Dense multi-head self-attention, reference models/vit.py:56-104."""

    def __init__(self, embed_dim, num_heads, dropout=0.0):
        super().__init__()
        self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.head_dim = embed_dim // num_heads
        assert self.head_dim * num_heads == embed_dim, "embed_dim must be divisible by num_heads"
        self.qkv = nn.Linear(embed_dim, embed_dim * 3)
        self.attn_dropout = nn.Dropout(dropout)
        self.proj = nn.Linear(embed_dim, embed_dim)
        self.proj_dropout = nn.Dropout(dropout)

    def _chain(self):
        return F.DenseChain(self.num_heads, self.attn_dropout.p, self.proj_dropout.p)

    def forward(self, x):
        ch = self._chain()
        return F.run(F.AttnOp(ch, None, self.training), [x], params(self, ch.names))


class MLP(nn.Module):
    """reference models/vit.py:107-139"""

    def __init__(self, in_features, hidden_features, out_features, dropout=0.0):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.dropout = nn.Dropout(dropout)

    def _chain(self):
        return F.MLPChain(self.dropout.p)

    def forward(self, x):
        ch = self._chain()
        return F.run(F.MLPOp(ch, self.training), [x], params(self, ch.names))


class TransformerBlock(nn.Module):
    """Pre-LN encoder block, reference models/vit.py:142-179."""

    def __init__(self, embed_dim, num_heads, mlp_ratio=4.0, dropout=0.0, attn_dropout=0.0):
        super().__init__()
        self.norm1 = nn.LayerNorm(embed_dim)
        self.attn = MultiHeadAttention(embed_dim, num_heads, attn_dropout)
        self.norm2 = nn.LayerNorm(embed_dim)
        self.mlp = MLP(in_features=embed_dim, hidden_features=int(embed_dim * mlp_ratio), out_features=embed_dim,
                       dropout=dropout)

    def _spec(self):
        return F.BlockSpec(self.attn._chain(), self.mlp._chain())

    def forward(self, x):
        spec = self._spec()
        return F.run(F.EncoderOp([spec], None, self.training), [x], params(self, spec.names))


def run_encoder(blocks, x, mask, training):
    """All blocks of a model as ONE autograd node (no per-block fp32<->bf16 gradient casts) -- or, while
    functional.encoder_segments(n) is active (train.GraphedStep with backward segments), as n consecutive nodes cut
    apart at detached boundary tensors, so that backward can be run, and captured, segment by segment."""
    blocks = list(blocks)
    nseg = max(1, min(F.get_encoder_segments(), len(blocks)))
    m = F._mask_u8(mask)
    per = (len(blocks) + nseg - 1) // nseg
    for i in range(0, len(blocks), per):
        grp = blocks[i:i + per]
        specs = [b._spec() for b in grp]
        prm = []
        for b, s in zip(grp, specs):
            prm += params(b, s.names)
        if i:
            x = F.note_segment_boundary(x)        # the next node starts a fresh autograd graph at a leaf copy of x
        x = F.run(F.EncoderOp(specs, m, training), [x], prm)
    return x


def embed_dropout(x, p, training):
    if not training or p <= 0.0:
        return x
    return F.run(F.DropoutOp(p), [x], [])


class VisionTransformer(nn.Module):
    """reference models/vit.py:182-331"""

    def __init__(self, img_size=224, patch_size=4, in_channels=3, num_classes=1000, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4.0, dropout=0.0, attn_dropout=0.0, embed_dropout=0.0):
        super().__init__()
        self.img_size = img_size
        self.patch_size = patch_size
        self.in_channels = in_channels
        self.num_classes = num_classes
        self.embed_dim = embed_dim
        self.depth = depth
        self.num_heads = num_heads
        self.patch_embed = PatchEmbedding(img_size=img_size, patch_size=patch_size, in_channels=in_channels,
                                          embed_dim=embed_dim)
        num_patches = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, embed_dim))
        self.pos_drop = nn.Dropout(embed_dropout)
        self.blocks = nn.ModuleList([
            TransformerBlock(embed_dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, dropout=dropout,
                             attn_dropout=attn_dropout) for _ in range(depth)])
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

    def forward_features(self, x):
        tok = self.patch_embed(x)
        x = F.run(F.PrologueOp(True), [tok], [self.cls_token, self.pos_embed])
        x = embed_dropout(x, self.pos_drop.p, self.training)
        x = run_encoder(self.blocks, x, None, self.training)
        return F.run(F.FinalNormOp(), [x], [self.norm.weight, self.norm.bias])

    def forward(self, x):
        x = self.forward_features(x)
        return F.run(F.LinearOp(), [x], [self.head.weight, self.head.bias])

    def get_num_parameters(self):
        return sum(p.numel() for p in self.parameters() if p.requires_grad)
