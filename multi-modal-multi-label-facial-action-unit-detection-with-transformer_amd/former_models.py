"""The other ``*former`` entries of the reference's model registry (train.py:292-303): ``sformer`` (SpatialFormer,
sformer.py:338-382), ``vformer`` (VisualFormer, vformer.py:295-387) and ``tformer`` (SpatialTemporalFormer,
tformer.py:296-436), assembled from the token sections this package builds on the HIP path - ``ResFormerTokens``
(sformer.py:313-327), ``TFormer`` (vformer.py:270-293), ``AU_former`` / ``tformer_AU_head`` - around a caller-supplied
CNN backbone.

The ResNet stages of the reference's ``ResFormer`` are conv-bound and out of scope: ``backbone=``
takes them as two modules, ``stem`` (frames [B', C, H, W] -> stage-3 feature map [B', 256, 7, 7]) and ``tail`` (feature map
-> pooled features [B', 512]); ``ResFormerShell`` runs the token section between them exactly where ``ResFormer.forward``
does.  Without a backbone the models consume what the backbone would produce (a [B', 256, 7, 7] map, ``tail`` = global
average pool + a fixed channel tiling to 512 features), which is enough to drive the token path, the heads and the
[B, 21] output contract.  Module / parameter names follow the reference so that its checkpoints load with strict=False.
The small ``fc`` heads (BatchNorm1d / Linear) are plain PyTorch modules - plumbing either side of the hot path."""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import nn

from .heads import AU_former, ResFormerTokens, TFormer, VA_former, tformer_AU_head
from .loss import AULoss
from .models import _TaskLossMixin


class _PoolTile(nn.Module):
    """stand-in for ResNet stage 4 + avgpool when no backbone is given: [B', C, h, w] -> [B', 512]"""

    def __init__(self, out_features=512):
        super().__init__()
        self.out_features = out_features

    def forward(self, x):
        f = x.mean(dim=(2, 3))
        rep = -(-self.out_features // f.shape[1])
        return f.repeat(1, rep)[:, :self.out_features]


class ResFormerShell(ResFormerTokens):
    """``ResFormer.forward`` (sformer.py:300-337) with the conv stages supplied by the caller:
    ``x[B, T, C, H, W] -> view(-1, C, H, W) -> stem -> token section (HIP) -> tail -> [B*T, 512]``.  Subclass of the token
    section, so its parameters carry the reference's names (``base_model.pos_embedding``, ``base_model.spatial_transformer.*``)."""

    def __init__(self, backbone: Optional[Tuple[nn.Module, nn.Module]] = None, dropout=0.0, compute_dtype="bf16"):
        super().__init__(dropout=dropout, compute_dtype=compute_dtype)
        stem, tail = backbone if backbone is not None else (nn.Identity(), _PoolTile())
        self.stem, self.tail = stem, tail

    def forward(self, x):
        if x.dim() == 5:
            x = x.contiguous().view(-1, *x.shape[2:])        # sformer.py:301-302
        return self.tail(super().forward(self.stem(x)))


def _fc_head(in_features):
    return nn.Sequential(nn.BatchNorm1d(in_features), nn.Linear(in_features, 256), nn.BatchNorm1d(256), nn.Linear(256, 12 + 7 + 2))


class _FormerTask(nn.Module, _TaskLossMixin):
    num_channels = 3

    def _config(self, modality, task):
        """``config_modality`` (sformer.py:384-391, same in vformer / tformer): RGB + mask -> the last 4 channels of the
        clip, mask only -> the last one, otherwise the 3 RGB channels (the caller's stem must take that many: the reference
        rebuilds ``conv1`` there, which belongs to the backbone this package does not ship).  Tasks as in the reference: AU, EX
        and VA columns come out of the fc head, with ``sformer`` overwriting the AU / VA columns by its ``AU_former`` /
        ``VA_former`` heads (sformer.py:375-380); ``vformer`` and ``tformer`` have no VA head (their VA columns are the fc head's)."""
        if 'M' in modality:
            self.num_channels = 4 if 'V' in modality else 1

    def _select(self, x):
        clip = x['clip']
        if clip.dim() == 5 and self.has_backbone:             # [B, C, T, H, W] -> [B, T, C, H, W] (sformer.py:374-377)
            clip = clip[:, -self.num_channels:].permute(0, 2, 1, 3, 4)
        return clip


class SpatialFormerModel(_FormerTask):
    """registry name ``sformer`` (sformer.py:338-382): per-frame features -> fc head, AU logits from ``AU_former``"""

    def __init__(self, modality='A;V;M', video_pretrained=True, task='EX', backbone=None, compute_dtype="bf16"):
        super().__init__()
        self.has_backbone = backbone is not None
        self.base_model = ResFormerShell(backbone, dropout=0.2, compute_dtype=compute_dtype)
        self._config(modality, task)
        self.task, self.modes = task, ["clip"]
        self.fc = _fc_head(512)
        self.au_head = AU_former(dropout=0.2, compute_dtype=compute_dtype)
        self.va_head = VA_former(dropout=0.2, compute_dtype=compute_dtype)   # sformer.py:358
        # DEVIATION, on purpose: the reference's SpatialFormer trains AU with DiceAULoss (multi-label Dice + 5 x weighted BCE,
        # sformer.py:362, loss.py:149-176); this entry uses the AULoss of the path SURVEY.md section 8 scopes (loss.py:63-103,
        # what avformer / vformer / tformer use).  INTEGRATION.md section 4 says so.
        self.loss_AU = AULoss()

    def forward(self, x):
        features = self.base_model(self._select(x))
        out = self.fc(features)
        if self.task == 'AU':
            au_out, _ = self.au_head(features)                 # sformer.py:382-384
            out = torch.cat([au_out[:, :12].to(out.dtype), out[:, 12:]], dim=1)
        if self.task == 'VA':
            va_out, _ = self.va_head(features)                 # sformer.py:378-380: out[:, -2:] = va_out
            out = torch.cat([out[:, :-2], va_out.to(out.dtype)], dim=1)
        return out


class _VideoModel(nn.Module):
    def __init__(self, backbone, temporal_dim, au_tokens, compute_dtype):
        super().__init__()
        self.s_former = ResFormerShell(backbone, compute_dtype=compute_dtype)
        self.au_head = AU_former(dropout=0.2, compute_dtype=compute_dtype) if au_tokens else None
        self.t_former = TFormer(dim=temporal_dim, compute_dtype=compute_dtype)

    def forward(self, x):
        x = self.s_former(x)                                   # [B*16, 512]
        if self.au_head is not None:                           # tformer.py:310-313: the 12 AU tokens of every frame, flattened
            _, tok = self.au_head(x)
            x = tok.reshape(tok.shape[0], -1)                  # [B*16, 12*128]
        return self.t_former(x)                                # view(-1, 16, dim) -> cls token [B, dim]


class VisualFormerModel(_FormerTask):
    """registry name ``vformer`` (vformer.py:358-387): S-Former frames -> ``TFormer`` over 16 frames -> fc head"""

    def __init__(self, modality='A;V;M', video_pretrained=True, task='EX', backbone=None, compute_dtype="bf16"):
        super().__init__()
        self.has_backbone = backbone is not None
        self.video_model = _VideoModel(backbone, 512, False, compute_dtype)
        self._config(modality, task)
        self.task, self.modes = task, ["clip"]
        self.fc = _fc_head(512)
        self.loss_AU = AULoss()

    def forward(self, x):
        return self.fc(self.video_model(self._select(x)))


class SpatialTemporalFormerModel(_FormerTask):
    """registry name ``tformer`` (tformer.py:405-436): per-frame AU tokens -> ``TFormer(dim=1536)`` -> fc head, AU logits
    from ``tformer_AU_head`` on the [B, 12, 128] view of the temporal feature"""

    def __init__(self, modality='A;V;M', video_pretrained=True, task='EX', backbone=None, compute_dtype="bf16"):
        super().__init__()
        self.has_backbone = backbone is not None
        self.video_model = _VideoModel(backbone, 128 * 12, True, compute_dtype)
        self._config(modality, task)
        self.task, self.modes = task, ["clip"]
        self.au_head = tformer_AU_head(dropout=0.2, compute_dtype=compute_dtype)
        self.fc = _fc_head(128 * 12)
        self.loss_AU = AULoss()

    def forward(self, x):
        f = self.video_model(self._select(x))
        out = self.fc(f)
        au = self.au_head(f)                                   # tformer.py:432-433: out[:, :12] = au_out
        return torch.cat([au[:, :12].to(out.dtype), out[:, 12:]], dim=1)
