"""Token producers / consumers either side of the transformer block, host-side mirrors of the
reference classes (same constructor arguments, parameter names and return values) built on the
HIP ``Transformer``:

* ``AU_former``        - reference models/heads.py:258-339
* ``VA_former``        - reference models/heads.py:341-372 (two valence / arousal tokens; SpatialFormer's ``va_head``)
* ``tformer_AU_head``  - reference models/tformer.py:362-403; ``former_AU_head`` is the same structure and
                         stands in for the class models/avformer.py:19,87 imports but the reference never defines
* ``TFormer``          - reference models/vformer.py:270-293 (= tformer.py:271-294)
* ``ResFormerTokens``  - the token section of ``ResFormer.forward``, reference models/sformer.py:313-327
                         (= vformer.py:245-259, tformer.py:246-260); the conv stages around it are out of scope

Everything outside the block runs in the library too (csrc/heads.hip + the parity GEMM), fp32:
BatchNorm1d (batch statistics / running statistics and their update), the 12 projections as ONE GEMM on the concatenated
weights with bias and positional table in its epilogue, the 12 per-token dots, cls / positional assembly, the
feature-axis fusion of avformer.py:100, the feature-map <-> token permutes.  The 24 ``nn.Linear`` holders keep the
reference's parameter names so its checkpoints load; ``_FlatGroup`` keeps their storage adjacent so that no
concatenation runs per step.
"""
from __future__ import annotations

from typing import List, Optional

import torch
from torch import nn

from . import ops
from .transformer import Transformer


def _need_gpu(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(f"{what} (HIP) needs its input on the MI355X; there is no CPU fallback - "
                           "use oracle/ only as a test checker")


class _FlatGroup:
    """Keeps the ``.data`` of same-shaped Parameters as consecutive slices of ONE buffer, so that a group of small
    ``nn.Linear`` layers is one GEMM operand without a per-step concatenation.  The Parameters stay the objects the
    optimizer and ``state_dict`` know; only their storage is re-pointed (``p.data = flat[i]``).  ``Module.to`` /
    ``.cuda()`` give every Parameter fresh storage: the next ``get()`` notices (addresses are no longer adjacent) and
    re-packs once."""

    def __init__(self, owners: List[nn.Module], attr: str):
        # the Parameters are looked up on their modules at every use: ``load_state_dict(assign=True)``, parametrizations or
        # a plain ``lin.weight = nn.Parameter(...)`` REPLACE the objects, and a list captured here would go stale (forward
        # on old tensors, gradients to orphans)
        self._owners = list(owners)
        self._attr = attr
        self.flat: Optional[torch.Tensor] = None

    @property
    def params(self) -> List[nn.Parameter]:
        return [getattr(m, self._attr) for m in self._owners]

    def get(self) -> torch.Tensor:
        ps = self.params
        p0 = ps[0]
        step = p0.numel() * p0.element_size()
        f = self.flat
        if (f is None or f.device != p0.device or f.dtype != p0.dtype
                or any(p.data_ptr() != f.data_ptr() + i * step for i, p in enumerate(ps))):
            f = torch.empty((len(ps),) + tuple(p0.shape), dtype=p0.dtype, device=p0.device)
            with torch.no_grad():
                for i, p in enumerate(ps):
                    f[i].copy_(p.data)
                    p.data = f[i]
            self.flat = f
        return f


class _FrontFn(torch.autograd.Function):
    """the front of AU_former / VA_former (heads.py:291-323, 354-366): BatchNorm1d -> T x Linear(in, E) -> [B, T, E] +
    pos_embedding  (T = 12 AU tokens, 2 VA tokens)"""

    @staticmethod
    def forward(ctx, x, mod, bn_w, bn_b, pos, *wb):
        bn = mod._front_bn()
        T = mod.n_tokens
        training = bool(bn.training or bn.running_mean is None)
        x = x.detach().float().contiguous()
        B = x.shape[0]
        E = pos.shape[-1]
        y, mean, invstd = ops.bn1d_fwd(x, bn_w.detach(), bn_b.detach(), bn.running_mean, bn.running_var,
                                       bn.num_batches_tracked if bn.training else None, bn.eps,
                                       0.1 if bn.momentum is None else bn.momentum, training)
        W = mod._proj_w.get().view(T * E, -1)    # [T E, in]: token i = rows i*E .. (i+1)*E (cat on dim 1, heads.py:318-319)
        bias = mod._proj_b.get().view(T * E)
        tokens = ops.gemm(y, W, epilogue=ops.EPI_BIAS_RES, bias=bias, residual=pos.detach().reshape(T * E), residual_ld=0)
        ctx.save_for_backward(x, y, mean, invstd, bn_w.detach(), W)
        ctx.training, ctx.E, ctx.T = training, E, T
        return tokens.view(B, T, E)

    @staticmethod
    def backward(ctx, dtok):
        x, y, mean, invstd, bn_w, W = ctx.saved_tensors
        E, T = ctx.E, ctx.T
        B = x.shape[0]
        dt = dtok.contiguous().view(B, T * E)
        dbias = ops.colsum(dt)                                   # [T E]: the T bias gradients, concatenated
        dpos = dbias.clone().view(1, T, E) if ctx.needs_input_grad[4] else None  # same sums, own storage (no aliased .grad)
        dW = ops.gemm(dt, y, trans_a=True, trans_b=False)        # [T E, in]
        dy = ops.gemm(dt, W, trans_b=False)                      # [B, in]
        dx, dg, db = ops.bn1d_bwd(x, dy, bn_w, mean, invstd, ctx.training, need_dx=ctx.needs_input_grad[0])
        gw = [dW[i * E:(i + 1) * E] for i in range(T)]
        gb = [dbias[i * E:(i + 1) * E] for i in range(T)]
        return (dx, None, dg, db, dpos, *gw, *gb)


class _DotsFn(torch.autograd.Function):
    """the T per-token bias-free Linear(E, 1) heads (heads.py:325-337, 367-369): logits[b, i] = tokens[b, i, :] . w_i, written
    into a [B, pad_to] row (columns T.. zero: the [B,21] layout)"""

    @staticmethod
    def forward(ctx, tokens, mod, pad_to, *w):
        tokens = tokens.detach().float().contiguous()
        W = mod._last_w.get().view(mod.n_tokens, -1)
        ctx.save_for_backward(tokens, W)
        ctx.T = mod.n_tokens
        return ops.token_dots_fwd(tokens, W, pad_to)

    @staticmethod
    def backward(ctx, dout):
        tokens, W = ctx.saved_tensors
        dtok, dw = ops.token_dots_bwd(dout, tokens, W, need_dtokens=ctx.needs_input_grad[0])
        return (dtok, None, None, *[dw[i:i + 1] for i in range(ctx.T)])


class _AssembleFn(torch.autograd.Function):
    """out[b] = cat(lead, x[b]) + pos  (TFormer: lead = cls_token, vformer.py:279-287; lead None: a bare positional add)"""

    @staticmethod
    def forward(ctx, x, lead, pos):
        ctx.n_lead = 0 if lead is None else lead.numel() // x.shape[-1]
        ctx.shapes = (None if lead is None else lead.shape, pos.shape)
        T = x.shape[1] + ctx.n_lead
        return ops.assemble_tokens(x.detach().float(), None if lead is None else lead.detach(),
                                   pos.detach().reshape(pos.shape[-2], pos.shape[-1])[:T])

    @staticmethod
    def backward(ctx, dout):
        B, T, D = dout.shape
        dout = dout.contiguous()
        lead_shape, pos_shape = ctx.shapes
        dx = dout[:, ctx.n_lead:] if ctx.needs_input_grad[0] else None
        dlead = None
        if ctx.n_lead and ctx.needs_input_grad[1]:
            dlead = ops.colsum(dout.view(B, T * D)[:, :ctx.n_lead * D]).view(lead_shape)
        dpos = None
        if ctx.needs_input_grad[2]:
            dpos = ops.colsum(dout.view(B, T * D)).view(T, D)
            if pos_shape[-2] > T:
                dpos = torch.nn.functional.pad(dpos, (0, 0, 0, pos_shape[-2] - T))
            dpos = dpos.view(pos_shape)
        return dx, dlead, dpos


class _CatFeaturesFn(torch.autograd.Function):
    """features = cat([a, v], dim=2) + pos  (avformer.py:100 then tformer.py:383-386)"""

    @staticmethod
    def forward(ctx, a, v, pos):
        ctx.ea = a.shape[2]
        ctx.pos_shape = pos.shape
        return ops.cat_features(a.detach().float(), v.detach().float(), pos.detach().reshape(pos.shape[-2], pos.shape[-1]))

    @staticmethod
    def backward(ctx, dout):
        B, T, E = dout.shape
        dout = dout.contiguous()
        da = dout[..., :ctx.ea] if ctx.needs_input_grad[0] else None
        dv = dout[..., ctx.ea:] if ctx.needs_input_grad[1] else None
        dpos = ops.colsum(dout.view(B, T * E)).view(ctx.pos_shape) if ctx.needs_input_grad[2] else None
        return da, dv, dpos


class _MapToTokensFn(torch.autograd.Function):
    """[B, C, h*w] -> [B, h*w, C] + pos  (sformer.py:316-318)"""

    @staticmethod
    def forward(ctx, x, pos):
        ctx.pos_shape = pos.shape
        S = x.shape[2]
        return ops.transpose_add(x.detach().float(), pos.detach().reshape(pos.shape[-2], pos.shape[-1])[:S])

    @staticmethod
    def backward(ctx, dt):
        B, S, Cn = dt.shape
        dt = dt.contiguous()
        dx = ops.transpose_add(dt, None) if ctx.needs_input_grad[0] else None
        dpos = None
        if ctx.needs_input_grad[1]:
            dpos = ops.colsum(dt.view(B, S * Cn)).view(S, Cn)
            if ctx.pos_shape[-2] > S:
                dpos = torch.nn.functional.pad(dpos, (0, 0, 0, ctx.pos_shape[-2] - S))
            dpos = dpos.view(ctx.pos_shape)
        return dx, dpos


class _TokensToMapFn(torch.autograd.Function):
    """[B, S, C] -> [B, C, S]  (sformer.py:326-327)"""

    @staticmethod
    def forward(ctx, t):
        return ops.transpose_add(t.detach().float(), None)

    @staticmethod
    def backward(ctx, dx):
        return ops.transpose_add(dx.contiguous(), None)


class _AUHeadBase(nn.Module):
    prefix, n_tokens = "AU", 12   # parameter-name prefix and token count of the head (VA_former: "VA", 2)

    def _front_bn(self):
        return getattr(self, f"{self.prefix}_BN1")

    def _make_front(self, input_dim, emb_dim):
        """BatchNorm1d + the T projections Linear(input_dim, emb_dim) under the reference's names"""
        T, px = self.n_tokens, self.prefix
        setattr(self, f"{px}_BN1", nn.BatchNorm1d(input_dim))
        for i in range(1, T + 1):
            setattr(self, f"{px}_linear_p{i}", nn.Linear(input_dim, emb_dim))
        self._proj_w = _FlatGroup([getattr(self, f"{px}_linear_p{i}") for i in range(1, T + 1)], "weight")
        self._proj_b = _FlatGroup([getattr(self, f"{px}_linear_p{i}") for i in range(1, T + 1)], "bias")

    def _make_last(self, emb_dim):
        T, px = self.n_tokens, self.prefix
        for i in range(1, T + 1):
            setattr(self, f"{px}_linear_last{i}", nn.Linear(emb_dim, 1, bias=False))
        self._last_w = _FlatGroup([getattr(self, f"{px}_linear_last{i}") for i in range(1, T + 1)], "weight")

    def _front_tokens(self, emb):
        bn = self._front_bn()
        return _FrontFn.apply(emb, self, bn.weight, bn.bias, self.pos_embedding, *self._proj_w.params, *self._proj_b.params)

    def _last_logits(self, tokens, pad_to=None):
        # token i -> bias-free Linear(emb,1) number i+1  == batched row-dot [B,T,E] . [T,E]
        return _DotsFn.apply(tokens, self, pad_to, *self._last_w.params)


class AU_former(_AUHeadBase):
    def __init__(self, input_dim=512, emb_dim=128, dropout=0.0, compute_dtype="bf16"):
        super().__init__()
        self.emb_dim = input_dim
        self._make_front(self.emb_dim, emb_dim)
        self.pos_embedding = nn.Parameter(torch.randn(1, 12, emb_dim))
        self.corr_transformer = Transformer(emb_dim, depth=2, heads=8, mlp_dim=256, dim_head=32, dropout=dropout,
                                            compute_dtype=compute_dtype)
        self._make_last(emb_dim)

    def tokens(self, emb):
        """the AU tokens [B, 12, E] after the correlation transformer (what avformer.py:96-99 keeps of this head)"""
        _need_gpu(emb, "AU_former")
        return self.corr_transformer(self._front_tokens(emb))

    def forward(self, emb):
        out = self.tokens(emb)
        return self._last_logits(out), out


class VA_former(_AUHeadBase):
    """reference models/heads.py:341-372: BatchNorm1d -> 2 x Linear(in, E) -> [B, 2, E] + pos_embedding ->
    Transformer(E, depth 2, 8 heads of 32, mlp 128) -> one bias-free Linear(E, 1) per token -> (valence / arousal [B, 2], tokens).
    Parameter names as the reference's (``VA_BN1``, ``VA_linear_p1/2``, ``pos_embedding``, ``corr_transformer.*``,
    ``VA_linear_last1/2``).  Two tokens per clip: the single-launch small-token layer kernels (layer_small.hip)."""
    prefix, n_tokens = "VA", 2

    def __init__(self, input_dim=512, emb_dim=128, dropout=0.0, compute_dtype="bf16"):
        super().__init__()
        self.emb_dim = input_dim
        self._make_front(self.emb_dim, emb_dim)
        self.pos_embedding = nn.Parameter(torch.randn(1, 2, emb_dim))
        self.corr_transformer = Transformer(emb_dim, depth=2, heads=8, mlp_dim=128, dim_head=32, dropout=dropout,
                                            compute_dtype=compute_dtype)
        self._make_last(emb_dim)

    def forward(self, emb):
        _need_gpu(emb, "VA_former")
        out = self.corr_transformer(self._front_tokens(emb))
        return self._last_logits(out), out


class tformer_AU_head(_AUHeadBase):
    def __init__(self, emb_dim=128, dropout=0.0, compute_dtype="bf16"):
        super().__init__()
        self.pos_embedding = nn.Parameter(torch.randn(1, 12, emb_dim))
        self.corr_transformer = Transformer(emb_dim, depth=3, heads=8, mlp_dim=256, dim_head=32, dropout=dropout,
                                            compute_dtype=compute_dtype)
        self._make_last(emb_dim)

    def forward(self, input, pad_to=None):
        _need_gpu(input, "tformer_AU_head")
        bs = input.shape[0]
        tokens = _AssembleFn.apply(input.reshape(bs, 12, -1), None, self.pos_embedding)
        return self._last_logits(self.corr_transformer(tokens), pad_to)

    def forward_fused(self, a_tokens, v_tokens, pad_to=None):
        """``forward(cat([a_tokens, v_tokens], dim=2))`` with the concatenation and the positional add in one pass, and the
        logits written into a zero-padded [B, pad_to] row (avformer.py:100-105)"""
        _need_gpu(a_tokens, "tformer_AU_head")
        tokens = _CatFeaturesFn.apply(a_tokens, v_tokens, self.pos_embedding)
        return self._last_logits(self.corr_transformer(tokens), pad_to)


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
        _need_gpu(x, "TFormer")
        x = x.contiguous().view(-1, self.num_patches, self.dim)
        x = _AssembleFn.apply(x, self.cls_token, self.pos_embedding)   # cat(cls, x) + pos[:, :n+1], vformer.py:282-284
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
        _need_gpu(x, "ResFormerTokens")
        b_l, c, h, w = x.shape
        t = _MapToTokensFn.apply(x.reshape(b_l, c, h * w), self.pos_embedding)   # permute(0,2,1) + pos, sformer.py:316-318
        t = self.spatial_transformer(t)
        return _TokensToMapFn.apply(t).reshape(b_l, c, h, w)                      # sformer.py:326-327
