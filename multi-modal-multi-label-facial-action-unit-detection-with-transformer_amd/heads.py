"""Token producers / consumers either side of the transformer block, host-side mirrors of the
reference classes (same constructor arguments, parameter names and return values) built on the
HIP ``Transformer``:

* ``AU_former``        - reference models/heads.py:258-339
* ``tformer_AU_head``  - reference models/tformer.py:362-403; ``former_AU_head`` is the same structure and
                         stands in for the class models/avformer.py:19,87 imports but the reference never defines
* ``TFormer``          - reference models/vformer.py:270-293 (= tformer.py:271-294)
* ``ResFormerTokens``  - the token section of ``ResFormer.forward``, reference models/sformer.py:313-327
                         (= vformer.py:245-259, tformer.py:246-260); the conv stages around it are out of scope

The glue outside the block (BatchNorm1d, the 12 small projections, the 12 per-token dots, cls/pos
assembly) is plain PyTorch on the GPU in this version ("next"); the 24
``nn.Linear`` holders keep the reference's parameter names so its checkpoints load.
"""
from __future__ import annotations

import torch
from torch import nn

from .transformer import Transformer


class _AUHeadBase(nn.Module):
    def _make_last(self, emb_dim):
        for i in range(1, 13):
            setattr(self, f"AU_linear_last{i}", nn.Linear(emb_dim, 1, bias=False))

    def _last_logits(self, tokens):
        # token i -> bias-free Linear(emb,1) number i+1  == batched row-dot [B,12,E] . [12,E]
        w = torch.cat([getattr(self, f"AU_linear_last{i}").weight for i in range(1, 13)], dim=0)  # [12, E]
        return (tokens * w.unsqueeze(0)).sum(dim=-1)


class AU_former(_AUHeadBase):
    def __init__(self, input_dim=512, emb_dim=128, dropout=0.0, compute_dtype="bf16"):
        super().__init__()
        self.emb_dim = input_dim
        self.AU_BN1 = nn.BatchNorm1d(self.emb_dim)
        for i in range(1, 13):
            setattr(self, f"AU_linear_p{i}", nn.Linear(self.emb_dim, emb_dim))
        self.pos_embedding = nn.Parameter(torch.randn(1, 12, emb_dim))
        self.corr_transformer = Transformer(emb_dim, depth=2, heads=8, mlp_dim=256, dim_head=32, dropout=dropout,
                                            compute_dtype=compute_dtype)
        self._make_last(emb_dim)

    def forward(self, emb):
        bs = emb.shape[0]
        emb = self.AU_BN1(emb)
        w = torch.cat([getattr(self, f"AU_linear_p{i}").weight for i in range(1, 13)], dim=0)  # [12*E, in]
        b = torch.cat([getattr(self, f"AU_linear_p{i}").bias for i in range(1, 13)], dim=0)
        tokens = torch.nn.functional.linear(emb, w, b).view(bs, 12, -1)  # heads.py:318-319
        tokens = tokens + self.pos_embedding[:, :12]
        out = self.corr_transformer(tokens)
        return self._last_logits(out), out


class tformer_AU_head(_AUHeadBase):
    def __init__(self, emb_dim=128, dropout=0.0, compute_dtype="bf16"):
        super().__init__()
        self.pos_embedding = nn.Parameter(torch.randn(1, 12, emb_dim))
        self.corr_transformer = Transformer(emb_dim, depth=3, heads=8, mlp_dim=256, dim_head=32, dropout=dropout,
                                            compute_dtype=compute_dtype)
        self._make_last(emb_dim)

    def forward(self, input):
        bs = input.shape[0]
        tokens = input.reshape(bs, 12, -1)
        tokens = tokens + self.pos_embedding[:, :12]
        out = self.corr_transformer(tokens)
        return self._last_logits(out)


class former_AU_head(tformer_AU_head):
    """The head ``models/avformer.py:87`` instantiates as ``former_AU_head(emb_dim=256, dropout=0.2)``; the
    reference imports it from ``.heads`` (avformer.py:19) but never defines it - its closest surviving
    definition is ``tformer_AU_head``, whose structure this class takes."""


class TFormer(nn.Module):
    def __init__(self, num_patches=16, dim=512, depth=3, heads=8, mlp_dim=1024, dim_head=64, dropout=0.0,
                 compute_dtype="bf16"):
        super().__init__()
        self.num_patches = num_patches
        self.dim = dim
        self.cls_token = nn.Parameter(torch.randn(1, 1, dim))
        self.pos_embedding = nn.Parameter(torch.randn(1, num_patches + 1, dim))
        self.spatial_transformer = Transformer(dim, depth, heads, dim_head, mlp_dim, dropout,
                                               compute_dtype=compute_dtype)

    def forward(self, x):
        x = x.contiguous().view(-1, self.num_patches, self.dim)
        b, n, _ = x.shape
        x = torch.cat((self.cls_token.expand(b, -1, -1), x), dim=1)
        x = x + self.pos_embedding[:, :(n + 1)]
        x = self.spatial_transformer(x)
        return x[:, 0]


class ResFormerTokens(nn.Module):
    """The transformer section in the middle of the reference's ``ResFormer`` (S-Former after ResNet stage 3): the
    stage-3 feature map [B', C, h, w] becomes h*w tokens of width C, gets the learned positional embedding, runs through
    ``spatial_transformer`` and is folded back to [B', C, h, w] for stage 4.  Parameter names (``pos_embedding``,
    ``spatial_transformer.*``) are ResFormer's, so its checkpoints load with ``strict=False``; defaults are its ctor's
    (``sformer.py:240``: 49 patches, dim 256, depth 1, 8 heads of 32, mlp 512)."""

    def __init__(self, num_patches=7 * 7, dim=256, depth=1, heads=8, mlp_dim=512, dim_head=32, dropout=0.0,
                 compute_dtype="bf16"):
        super().__init__()
        self.pos_embedding = nn.Parameter(torch.randn(1, num_patches, dim))
        self.spatial_transformer = Transformer(dim, depth, heads, dim_head, mlp_dim, dropout,
                                               compute_dtype=compute_dtype)

    def forward(self, x):
        b_l, c, h, w = x.shape
        t = x.reshape((b_l, c, h * w)).permute(0, 2, 1)
        t = t + self.pos_embedding[:, :t.shape[1]]
        t = self.spatial_transformer(t)
        return t.permute(0, 2, 1).reshape((b_l, c, h, w))
