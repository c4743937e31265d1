"""Functional CPU restatement of the reference's transformer hot path (test oracle).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Every function is a plain, un-fused, op-for-op restatement of the math the reference
executes (fp32 eager ``matmul`` / ``softmax`` / 9-op tanh-GELU), written against a
*state dict* that uses the reference's parameter names, so reference checkpoints and
golden fixtures plug in directly.  Backward passes come from ``torch.autograd`` over
these ops - that is exactly how the reference gets its gradients.

Citations are to ``/root/reference`` (file:line).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch

Tensor = torch.Tensor

# models/loss.py:73 - pos_weight of the BCE-with-logits AU loss
AU_POS_WEIGHT = (1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 3.0, 3.0, 3.0, 1.0, 2.0)

_GELU_C = math.sqrt(2.0 / math.pi)


def gelu_tanh(u: Tensor) -> Tensor:
    """models/heads.py:164-166 - tanh approximation, written with the same op order."""
    cube = torch.pow(u, 3)
    inner = _GELU_C * (u + 0.044715 * cube)
    return 0.5 * u * (1 + torch.tanh(inner))


def layernorm(x: Tensor, weight: Tensor, bias: Tensor, eps: float = 1e-5) -> Tensor:
    """models/heads.py:178-185 (nn.LayerNorm(dim), default eps, biased variance)."""
    mu = x.mean(dim=-1, keepdim=True)
    var = (x - mu).pow(2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * weight + bias


def attention_forward(h: Tensor, w_qkv: Tensor, w_out: Optional[Tensor], b_out: Optional[Tensor],
                      heads: int, mask: Optional[Tensor] = None) -> Tensor:
    """models/heads.py:219-239.  ``h`` is the already-normalised input [B, N, D].

    w_qkv is [3*I, D] (rows q|k|v, each head-major), no bias (heads.py:212).
    w_out/b_out None <=> the reference's nn.Identity case (heads.py:207).
    """
    B, N, _ = h.shape
    inner = w_qkv.shape[0] // 3
    dh = inner // heads
    qkv = h @ w_qkv.t()
    q, k, v = qkv.split(inner, dim=-1)

    def split_heads(t):  # 'b n (h d) -> b h n d'  (heads.py:222)
        return t.reshape(B, N, heads, dh).permute(0, 2, 1, 3)

    q, k, v = split_heads(q), split_heads(k), split_heads(v)
    scores = (q @ k.transpose(-1, -2)) * (dh ** -0.5)  # heads.py:224
    if mask is not None:  # heads.py:227-232 (dead in the reference: no caller passes a mask)
        m = torch.nn.functional.pad(mask.flatten(1), (1, 0), value=True)
        assert m.shape[-1] == scores.shape[-1], 'mask has incorrect dimensions'
        m2 = m[:, None, :, None] & m[:, None, None, :]
        scores = scores.masked_fill(~m2, -torch.finfo(scores.dtype).max)
    probs = scores.softmax(dim=-1)  # heads.py:234 - no dropout on probabilities
    o = probs @ v
    o = o.permute(0, 2, 1, 3).reshape(B, N, inner)  # 'b h n d -> b n (h d)'
    if w_out is None:
        return o
    return o @ w_out.t() + b_out


def feedforward_forward(h: Tensor, w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor) -> Tensor:
    """models/heads.py:188-200 with dropout = identity (eval / p=0)."""
    u = h @ w1.t() + b1
    return gelu_tanh(u) @ w2.t() + b2


def transformer_param_names(layer: int, project_out: bool = True) -> List[str]:
    """State-dict keys of one layer, in the reference's order."""
    p = f"layers.{layer}"
    names = [f"{p}.0.fn.norm.weight", f"{p}.0.fn.norm.bias", f"{p}.0.fn.fn.to_qkv.weight"]
    if project_out:
        names += [f"{p}.0.fn.fn.to_out.0.weight", f"{p}.0.fn.fn.to_out.0.bias"]
    names += [f"{p}.1.fn.norm.weight", f"{p}.1.fn.norm.bias",
              f"{p}.1.fn.fn.net.0.weight", f"{p}.1.fn.fn.net.0.bias",
              f"{p}.1.fn.fn.net.3.weight", f"{p}.1.fn.fn.net.3.bias"]
    return names


def layer_forward(x: Tensor, sd: Dict[str, Tensor], layer: int, heads: int, prefix: str = "",
                  mask: Optional[Tensor] = None, drop=None) -> Tensor:
    """One (Residual(PreNorm(Attention)), Residual(PreNorm(FeedForward))) pair - heads.py:246-255.

    ``drop`` (optional): three tensors of keep/(1-p) factors for the nn.Dropout sites of the layer - after to_out
    (heads.py:216), after GELU (heads.py:194), after net.3 (heads.py:196) - i.e. explicit masks instead of torch's
    generator, so that a kernel's counter-based masks can be replayed exactly."""
    p = f"{prefix}layers.{layer}"
    f0, f1, f2 = drop if drop is not None else (None, None, None)
    h = layernorm(x, sd[f"{p}.0.fn.norm.weight"], sd[f"{p}.0.fn.norm.bias"])
    a = attention_forward(h, sd[f"{p}.0.fn.fn.to_qkv.weight"],
                          sd.get(f"{p}.0.fn.fn.to_out.0.weight"), sd.get(f"{p}.0.fn.fn.to_out.0.bias"),
                          heads, mask)
    x = (a if f0 is None else a * f0.view_as(a)) + x
    h = layernorm(x, sd[f"{p}.1.fn.norm.weight"], sd[f"{p}.1.fn.norm.bias"])
    u = h @ sd[f"{p}.1.fn.fn.net.0.weight"].t() + sd[f"{p}.1.fn.fn.net.0.bias"]
    g = gelu_tanh(u)
    if f1 is not None:
        g = g * f1.view_as(g)
    f = g @ sd[f"{p}.1.fn.fn.net.3.weight"].t() + sd[f"{p}.1.fn.fn.net.3.bias"]
    x = (f if f2 is None else f * f2.view_as(f)) + x
    return x


def transformer_forward(x: Tensor, sd: Dict[str, Tensor], depth: int, heads: int, prefix: str = "",
                        mask: Optional[Tensor] = None, drop=None) -> Tensor:
    """models/heads.py:252-256.  ``drop``: optional list (one entry per layer) of the three factor tensors."""
    for i in range(depth):
        x = layer_forward(x, sd, i, heads, prefix, mask, None if drop is None else drop[i])
    return x


def init_transformer_state(dim: int, depth: int, heads: int, dim_head: int, mlp_dim: int,
                           generator: Optional[torch.Generator] = None,
                           dtype=torch.float32) -> Dict[str, Tensor]:
    """Random state dict with nn.Linear's default distribution (kaiming_uniform(a=sqrt 5) ==
    U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for both weight and bias) and LayerNorm ones/zeros."""
    inner = heads * dim_head
    project_out = not (heads == 1 and dim_head == dim)

    def uni(shape, fan_in):
        bound = 1.0 / math.sqrt(fan_in)
        return (torch.rand(shape, generator=generator, dtype=dtype) * 2 - 1) * bound

    sd: Dict[str, Tensor] = {}
    for i in range(depth):
        p = f"layers.{i}"
        sd[f"{p}.0.fn.norm.weight"] = torch.ones(dim, dtype=dtype)
        sd[f"{p}.0.fn.norm.bias"] = torch.zeros(dim, dtype=dtype)
        sd[f"{p}.0.fn.fn.to_qkv.weight"] = uni((3 * inner, dim), dim)
        if project_out:
            sd[f"{p}.0.fn.fn.to_out.0.weight"] = uni((dim, inner), inner)
            sd[f"{p}.0.fn.fn.to_out.0.bias"] = uni((dim,), inner)
        sd[f"{p}.1.fn.norm.weight"] = torch.ones(dim, dtype=dtype)
        sd[f"{p}.1.fn.norm.bias"] = torch.zeros(dim, dtype=dtype)
        sd[f"{p}.1.fn.fn.net.0.weight"] = uni((mlp_dim, dim), dim)
        sd[f"{p}.1.fn.fn.net.0.bias"] = uni((mlp_dim,), dim)
        sd[f"{p}.1.fn.fn.net.3.weight"] = uni((dim, mlp_dim), mlp_dim)
        sd[f"{p}.1.fn.fn.net.3.bias"] = uni((dim,), mlp_dim)
    return sd


# ----------------------------------------------------------------------------------------------
# heads / glue around the block
# ----------------------------------------------------------------------------------------------

def _last_linears(tokens: Tensor, sd: Dict[str, Tensor], prefix: str, n: int, stem: str) -> Tensor:
    # heads.py:325-337 / tformer.py:389-401: token i -> bias-free Linear(emb, 1) number i+1
    cols = [tokens[:, i, :] @ sd[f"{prefix}{stem}{i + 1}.weight"].t() for i in range(n)]
    return torch.cat(cols, dim=1)


def au_head_forward(features: Tensor, sd: Dict[str, Tensor], prefix: str = "", depth: int = 3,
                    heads: int = 8) -> Tensor:
    """models/tformer.py:381-403 (``tformer_AU_head``; stands in for the reference's missing
    ``former_AU_head``, avformer.py:19/87).  features [B, 12*E] or [B, 12, E] -> logits [B, 12]."""
    bs = features.shape[0]
    tok = features.reshape(bs, 12, -1)
    tok = tok + sd[f"{prefix}pos_embedding"][:, :12]
    out = transformer_forward(tok, sd, depth, heads, prefix=f"{prefix}corr_transformer.")
    return _last_linears(out, sd, prefix, 12, "AU_linear_last")


def au_former_forward(emb: Tensor, sd: Dict[str, Tensor], prefix: str = "", training: bool = False,
                      depth: int = 2, heads: int = 8, eps: float = 1e-5):
    """models/heads.py:291-339 (``AU_former``): BatchNorm1d -> 12 Linear(in,128) -> +pos ->
    Transformer -> 12 per-token dots.  Returns (logits [B,12], tokens [B,12,E])."""
    bs = emb.shape[0]
    w, b = sd[f"{prefix}AU_BN1.weight"], sd[f"{prefix}AU_BN1.bias"]
    if training:
        mu = emb.mean(dim=0)
        var = emb.var(dim=0, unbiased=False)
    else:
        mu, var = sd[f"{prefix}AU_BN1.running_mean"], sd[f"{prefix}AU_BN1.running_var"]
    e = (emb - mu) / torch.sqrt(var + eps) * w + b
    toks = [e @ sd[f"{prefix}AU_linear_p{i + 1}.weight"].t() + sd[f"{prefix}AU_linear_p{i + 1}.bias"]
            for i in range(12)]
    tok = torch.cat(toks, dim=1).reshape(bs, 12, -1)  # heads.py:318-319
    tok = tok + sd[f"{prefix}pos_embedding"][:, :12]
    out = transformer_forward(tok, sd, depth, heads, prefix=f"{prefix}corr_transformer.")
    return _last_linears(out, sd, prefix, 12, "AU_linear_last"), out


def va_former_forward(emb: Tensor, sd: Dict[str, Tensor], prefix: str = "", training: bool = False,
                      depth: int = 2, heads: int = 8, eps: float = 1e-5):
    """models/heads.py:354-372 (``VA_former``): BatchNorm1d -> 2 Linear(in, E) -> cat / view [B, 2, E] -> +pos ->
    Transformer -> one bias-free dot per token.  Returns (valence / arousal [B, 2], tokens [B, 2, E])."""
    bs = emb.shape[0]
    w, b = sd[f"{prefix}VA_BN1.weight"], sd[f"{prefix}VA_BN1.bias"]
    if training:
        mu = emb.mean(dim=0)
        var = emb.var(dim=0, unbiased=False)
    else:
        mu, var = sd[f"{prefix}VA_BN1.running_mean"], sd[f"{prefix}VA_BN1.running_var"]
    e = (emb - mu) / torch.sqrt(var + eps) * w + b
    toks = [e @ sd[f"{prefix}VA_linear_p{i + 1}.weight"].t() + sd[f"{prefix}VA_linear_p{i + 1}.bias"] for i in range(2)]
    tok = torch.cat(toks, dim=1).reshape(bs, 2, -1)  # heads.py:360-361
    tok = tok + sd[f"{prefix}pos_embedding"][:, :2]
    out = transformer_forward(tok, sd, depth, heads, prefix=f"{prefix}corr_transformer.")
    return _last_linears(out, sd, prefix, 2, "VA_linear_last"), out


def tformer_forward(x: Tensor, sd: Dict[str, Tensor], num_patches: int, dim: int, depth: int, heads: int,
                    prefix: str = "") -> Tensor:
    """models/vformer.py:279-293 (``TFormer``): view, prepend CLS, +pos, Transformer, take token 0."""
    x = x.contiguous().view(-1, num_patches, dim)
    b, n, _ = x.shape
    cls = sd[f"{prefix}cls_token"].expand(b, -1, -1)
    x = torch.cat((cls, x), dim=1)
    x = x + sd[f"{prefix}pos_embedding"][:, :(n + 1)]
    x = transformer_forward(x, sd, depth, heads, prefix=f"{prefix}spatial_transformer.")
    return x[:, 0]


def resformer_tokens_forward(x: Tensor, sd: Dict[str, Tensor], depth: int, heads: int, prefix: str = "") -> Tensor:
    """models/sformer.py:313-327 (= vformer.py:245-259, tformer.py:246-260), the token section of ``ResFormer.forward``:
    feature map [B', C, h, w] -> tokens [B', h*w, C] + pos_embedding -> spatial_transformer -> back to [B', C, h, w]."""
    b_l, c, h, w = x.shape
    t = x.reshape((b_l, c, h * w)).permute(0, 2, 1)
    t = t + sd[f"{prefix}pos_embedding"][:, :t.shape[1]]
    t = transformer_forward(t, sd, depth, heads, prefix=f"{prefix}spatial_transformer.")
    return t.permute(0, 2, 1).reshape((b_l, c, h, w))


def au_loss(y_pred: Tensor, y_true: Tensor, ignore: float = -1.0) -> Tensor:
    """models/loss.py:75-103 (``AULoss``): keep rows whose FIRST label != ignore; per-element
    BCE-with-logits with pos_weight; mean over kept rows x 12.  All rows dropped => NaN, as in
    the reference (mean of an empty tensor)."""
    keep = (y_true != ignore)[:, 0]
    z = y_pred[keep]
    y = y_true[keep]
    w = torch.tensor(AU_POS_WEIGHT, dtype=z.dtype, device=z.device)
    # stable form of -[w*y*log(sig z) + (1-y)*log(1-sig z)]
    per = (1 - y) * z + (1 + (w - 1) * y) * torch.nn.functional.softplus(-z)
    return per.mean()
