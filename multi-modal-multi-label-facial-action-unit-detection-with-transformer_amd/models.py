"""Task models and the model registry around the HIP transformer (reference train.py:292-315,
models/avformer.py:37-123).

``TwoStreamAuralVisualFormer`` keeps the reference's constructor signature, ``.modes`` / ``.task``
attributes, ``forward(dict) -> [B,21]`` layout ([0:12] AU logits, [12:19] EX, [19:21] VA) and
``get_*_loss`` methods, so the step loop of the reference's ``train.py:206-237`` runs on it
unchanged.  Its CNN feature extractors (ResNet18 on mel-spectrograms, Former-DFER video model) are
out of scope: ``AudioFormer`` / ``VisualFormer`` take a ``backbone``
module and default to the identity, i.e. they consume per-clip feature vectors [B, 512].

``SyntheticAVFormer`` is the BASELINE.json scale-up of the same block: video tokens [B,T_v,D] and
audio tokens [B,T_a,D] fused on the SEQUENCE axis, + positional embedding -> Transformer(D, L) ->
mean-pooled -> 12 AU logits -> AULoss.
"""
from __future__ import annotations

import os
from collections import OrderedDict

import torch
from torch import nn

from . import ops
from .heads import AU_former, TFormer, former_AU_head, tformer_AU_head  # noqa: F401
from .loss import AULoss
from .transformer import Transformer


def load_pretrain(model, weight_path):
    """reference avformer.py:28-35 - strips 'module.' prefixes, strict=False; a missing file is skipped
    (the reference hard-codes K:\\ paths and would crash).  See checkpoint.py for the other loaders."""
    from .checkpoint import load_pretrain as _load
    return _load(model, weight_path) is not None


class AudioFormer(nn.Module):
    def __init__(self, modality='A', audio_pretrained=False, task='EX', backbone=None, compute_dtype="bf16"):
        super().__init__()
        self.audio_model = backbone if backbone is not None else nn.Identity()
        self.task = task
        self.modes = ['audio_features']
        self.au_head = AU_former(dropout=0.2, compute_dtype=compute_dtype)

    def forward(self, x):
        return self.au_head.tokens(self.audio_model(x))  # avformer.py:96: only the tokens of this head are used


class VisualFormer(nn.Module):
    def __init__(self, modality='A;V', video_pretrained=True, task='EX', backbone=None, in_features=512,
                 compute_dtype="bf16"):
        super().__init__()
        self.video_model = backbone if backbone is not None else nn.Identity()
        self.task = task
        self.modes = ["clip"]
        self.au_head = AU_former(input_dim=in_features, compute_dtype=compute_dtype)

    def forward(self, x):
        return self.au_head.tokens(self.video_model(x))  # avformer.py:99


class _TaskLossMixin:
    def get_au_loss(self, y_pred, y_true):
        # loss on the AU slots 0..11 of the [B,21] row (train.py:136-138 / the reference models' get_au_loss)
        if y_true.dim() == 2 and y_true.shape[1] == 12 and hasattr(self.loss_AU, "forward_rows"):
            return self.loss_AU.forward_rows(y_pred, y_true)
        return self.loss_AU(y_pred[:, :12], y_true)

    def get_ex_loss(self, y_pred, y_true):
        # EX / VA tasks are outside the hot path; plain PyTorch equivalents
        return nn.functional.cross_entropy(y_pred[:, 12:19], y_true.view(-1), ignore_index=7)

    def get_va_loss(self, y_pred, y_true):
        def ccc_loss(p, t):
            pm, tm = p.mean(), t.mean()
            cov = ((p - pm) * (t - tm)).mean()
            return 1 - 2 * cov / (p.var(unbiased=False) + t.var(unbiased=False) + (pm - tm) ** 2 + 1e-8)
        v, a = torch.tanh(y_pred[:, 19]), torch.tanh(y_pred[:, 20])
        return 2 * ccc_loss(v, y_true[:, 0]) + ccc_loss(a, y_true[:, 1])


class TwoStreamAuralVisualFormer(nn.Module, _TaskLossMixin):
    def __init__(self, modality='A;V;M', video_pretrained=True, audio_pretrained=True, task='EX',
                 video_weights=None, audio_weights=None, compute_dtype="bf16"):
        super().__init__()
        self.audio_model = AudioFormer(compute_dtype=compute_dtype)
        self.video_model = VisualFormer(compute_dtype=compute_dtype)
        if video_pretrained and load_pretrain(self.video_model, video_weights):
            for p in self.video_model.parameters():
                p.requires_grad = False
        if audio_pretrained and load_pretrain(self.audio_model, audio_weights):
            for p in self.audio_model.parameters():
                p.requires_grad = False
        self.task = task
        self.au_head = former_AU_head(emb_dim=256, dropout=0.2, compute_dtype=compute_dtype)
        self.modes = ['clip', 'audio_features']
        self.loss_AU = AULoss()
        self.concurrent_streams = True
        self._side_streams = {}

    def forward(self, x):
        a_in, v_in = x['audio_features'], x['clip']
        if self.concurrent_streams and a_in.is_cuda and torch.cuda.is_current_stream_capturing():
            # the two streams of the model are independent up to the fusion: while a hipGraph is being captured the aural
            # one is recorded on a side HIP stream (autograd replays each branch's backward on its forward stream), so the
            # replayed graph runs both branches concurrently: 1.06 vs 1.18 ms per step at B=64.  Eager execution is
            # host-bound at these shapes and the extra stream bookkeeping costs 7 % there, so it stays on one stream.
            cur = torch.cuda.current_stream(a_in.device)
            side = self._side_streams.get(a_in.device)
            if side is None:
                side = self._side_streams[a_in.device] = torch.cuda.Stream(a_in.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                audio_tok = self.audio_model(a_in)
            video_tok = self.video_model(v_in)
            cur.wait_stream(side)
            audio_tok.record_stream(cur)
        else:
            audio_tok = self.audio_model(a_in)
            video_tok = self.video_model(v_in)
        if self.task == 'AU':
            # fusion on the FEATURE axis (avformer.py:100) + the head's positional add in one pass; the 12 logits land in
            # a zero-padded [B, 21] row (avformer.py:101-105)
            return self.au_head.forward_fused(audio_tok, video_tok, pad_to=21)
        return torch.zeros(audio_tok.shape[0], 21, device=audio_tok.device, dtype=torch.float32)


class _LinearPadFn(torch.autograd.Function):
    """out[:, :O] = x W^T + b, out[:, O:width] = 0 - the AU logits Linear written straight into the reference's [B,21] row"""

    @staticmethod
    def forward(ctx, x, w, b, width):
        x = x.detach().float().contiguous()
        O = w.shape[0]
        ctx.small = x.shape[1] % 4 == 0 and x.shape[1] <= 4096 and w.is_contiguous()
        if ctx.small:  # one launch (a few rows x a dozen outputs: the GEMM path is a split-K launch, a fold and a zero fill)
            out = ops.linear_pad_fwd(x, w.detach(), b.detach(), width)
        else:
            out = torch.empty((x.shape[0], width), dtype=torch.float32, device=x.device)
            ops.gemm(x, w.detach(), bias=b.detach(), out=out[:, :O])  # the fp32 kernel adds a non-null bias in every epilogue
            ops.zero_cols(out, O, width)
        ctx.save_for_backward(x, w.detach())
        ctx.O = O
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w = ctx.saved_tensors
        if ctx.small and dout.stride(1) == 1 and dout.dtype == torch.float32:
            dx, dw, db = ops.linear_pad_bwd(dout, x, w, *ctx.needs_input_grad[:3])
            return dx, dw, db, None
        dl = dout[:, :ctx.O]                                        # row stride `width`: read in place
        dw = ops.gemm(dl, x, trans_a=True, trans_b=False) if ctx.needs_input_grad[1] else None
        db = ops.colsum(dl) if ctx.needs_input_grad[2] else None
        dx = ops.gemm(dl, w, trans_b=False) if ctx.needs_input_grad[0] else None
        return dx, dw, db, None


class SyntheticAVFormer(nn.Module, _TaskLossMixin):
    """BASELINE.json configs C2-C5: one Transformer(dim, depth, heads, dim_head, mlp_dim) over the fused
    [B, T_v + T_a, dim] token sequence, mean pooling, 12 AU logits in the reference's [B,21] layout."""

    def __init__(self, dim=512, depth=6, heads=8, dim_head=64, mlp_dim=1024, t_video=196, t_audio=128, task='AU',
                 compute_dtype="bf16", residual_dtype="f32", dropout=0.0):
        super().__init__()
        self.task = task
        self.modes = ['clip', 'audio_features']
        self.t_video, self.t_audio = t_video, t_audio
        self.pos_embedding = nn.Parameter(torch.randn(1, t_video + t_audio, dim) * 0.02)
        # dropout: 0 in BASELINE's configs; the reference's real stacks train at 0.2 (heads.py:277) - bench.py --dropout times it
        self.transformer = Transformer(dim, depth, heads, dim_head, mlp_dim, dropout, compute_dtype=compute_dtype,
                                       residual_dtype=residual_dtype)
        self.au_fc = nn.Linear(dim, 12)
        self.loss_AU = AULoss()

    def forward(self, x):
        # fusion on the SEQUENCE axis + positional embedding: cat([clip, audio], 1) + pos_embedding, then the stack and
        # y.mean(dim=1).  Widths that are a multiple of 4 take the one-pass library kernels for both ends.
        if x['clip'].shape[-1] % 4 == 0:
            # (one library pass builds the sequence in the residual stream's storage type; its backward returns d clip,
            # d audio and d pos_embedding from the fp32 gradient of the sequence)
            pooled = self.transformer(x['clip'], pool='mean', fuse=(x['audio_features'], self.pos_embedding))
        else:
            tokens = torch.cat([x['clip'], x['audio_features']], dim=1)
            pooled = self.transformer(tokens + self.pos_embedding[:, :tokens.shape[1]]).mean(dim=1)
        # the reference's [B,21] layout: AU logits in slots 0..11, the rest zero
        return _LinearPadFn.apply(pooled, self.au_fc.weight, self.au_fc.bias, 21)


# name -> class, as the if/elif chain in the reference's train.py:292-315 does for --model_name
MODEL_REGISTRY = {
    'avformer': TwoStreamAuralVisualFormer,
    'avformer_synthetic': SyntheticAVFormer,
}


def _register_former_models():
    # sformer / vformer / tformer (train.py:292-303): the token sections on the HIP path around a caller's CNN backbone
    from . import former_models as fm
    MODEL_REGISTRY.update({'sformer': fm.SpatialFormerModel, 'vformer': fm.VisualFormerModel,
                           'tformer': fm.SpatialTemporalFormerModel})


def build_model(model_name: str, modality: str = 'A;V;M', task: str = 'AU', **kw) -> nn.Module:
    if 'sformer' not in MODEL_REGISTRY:
        _register_former_models()
    if model_name not in MODEL_REGISTRY:
        raise KeyError(f"model {model_name!r} is not provided by the MI355X hot-path build; available: "
                       f"{sorted(MODEL_REGISTRY)} (the CNN-only models of the reference are out of scope)")
    cls = MODEL_REGISTRY[model_name]
    if cls is SyntheticAVFormer:
        return cls(task=task, **kw)
    return cls(modality=modality, task=task, **kw)
