"""Tensor-level wrappers over the per-operator C entry points (used by the modules and by tests).

All tensors must live on a CUDA(HIP) device; nothing here falls back to PyTorch math.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import BF16, EPI_BIAS_GELU, EPI_BIAS_RES, EPI_DGELU, EPI_NONE, F32  # noqa: F401

_TORCH2AVF = {torch.float32: F32, torch.bfloat16: BF16}
_AVF2TORCH = {F32: torch.float32, BF16: torch.bfloat16}


def avf_dtype(dt) -> int:
    if isinstance(dt, int):
        return dt
    if isinstance(dt, str):
        return {"f32": F32, "fp32": F32, "float32": F32, "bf16": BF16, "bfloat16": BF16}[dt.lower()]
    return _TORCH2AVF[dt]


def torch_dtype(dt) -> torch.dtype:
    return _AVF2TORCH[avf_dtype(dt)]


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("libavformer_hip operators need tensors on the MI355X (got a CPU tensor); "
                               "there is no CPU fallback in the product path")


def _bytes(n: int, device) -> torch.Tensor:
    return torch.empty(max(int(n), 16), dtype=torch.uint8, device=device)


def device_ok() -> bool:
    return bool(_lib.load().avf_device_ok())


# ------------------------------------------------------------------------------------------------
def layernorm_fwd(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float = 1e-5,
                  out_dtype=torch.float32) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """nn.LayerNorm(dim) forward (reference models/heads.py:178-185).  x fp32 [..., D]."""
    _need_cuda(x, weight, bias)
    lib = _lib.load()
    x = x.contiguous()
    D = x.shape[-1]
    rows = x.numel() // D
    y = torch.empty(x.shape, dtype=torch_dtype(out_dtype), device=x.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    _lib.check(lib.avf_layernorm_fwd(_ptr(x), _ptr(weight), _ptr(bias), _ptr(y), avf_dtype(out_dtype), _ptr(mean),
                                     _ptr(rstd), rows, D, float(eps), _stream()), "layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy, x, weight, mean, rstd, dres=None, want_lo=False, want_colsum=False):
    """-> dx fp32, dx_lo (bf16 or None), dgamma, dbeta, colsum(dx) or None."""
    _need_cuda(dy, x, weight, mean, rstd, dres)
    lib = _lib.load()
    dy = dy.contiguous()
    x = x.contiguous()
    D = x.shape[-1]
    rows = x.numel() // D
    dx = torch.empty_like(x)
    dx_lo = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device) if want_lo else None
    dg = torch.empty(D, dtype=torch.float32, device=x.device)
    db = torch.empty(D, dtype=torch.float32, device=x.device)
    cs = torch.empty(D, dtype=torch.float32, device=x.device) if want_colsum else None
    ws = _bytes(lib.avf_layernorm_bwd_workspace_bytes(rows, D), x.device)
    _lib.check(lib.avf_layernorm_bwd(_ptr(dy), avf_dtype(dy.dtype), _ptr(x), _ptr(weight), _ptr(mean), _ptr(rstd),
                                     _ptr(dres.contiguous() if dres is not None else None), _ptr(dx), _ptr(dx_lo),
                                     _ptr(dg), _ptr(db), _ptr(cs), _ptr(ws), rows, D, _stream()), "layernorm_bwd")
    return dx, dx_lo, dg, db, cs


def colsum(t: torch.Tensor) -> torch.Tensor:
    """column sums over all leading axes; a 2-D view whose rows are a constant stride apart (unit column stride) is read
    in place"""
    _need_cuda(t)
    lib = _lib.load()
    if not (t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= t.shape[1]):
        t = t.contiguous()
    cols = t.shape[-1]
    rows = t.numel() // cols
    ld = t.stride(0) if t.dim() == 2 else cols
    out = torch.empty(cols, dtype=torch.float32, device=t.device)
    ws = _bytes(lib.avf_colsum_workspace_bytes(rows, cols), t.device)
    _lib.check(lib.avf_colsum(_ptr(t), avf_dtype(t.dtype), rows, cols, ld, _ptr(out), _ptr(ws), _stream()), "colsum")
    return out


def cast_bf16(t: torch.Tensor) -> torch.Tensor:
    _need_cuda(t)
    t = t.contiguous()
    out = torch.empty(t.shape, dtype=torch.bfloat16, device=t.device)
    _lib.check(_lib.load().avf_cast_f32_to_bf16(_ptr(t), _ptr(out), t.numel(), _stream()), "cast")
    return out


def prep_weight_bf16(w: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    _need_cuda(w)
    w = w.contiguous()
    r, c = w.shape
    lo = torch.empty((r, c), dtype=torch.bfloat16, device=w.device)
    lo_t = torch.empty((c, r), dtype=torch.bfloat16, device=w.device)
    _lib.check(_lib.load().avf_prep_weight_bf16(_ptr(w), _ptr(lo), _ptr(lo_t), r, c, _stream()), "prep_weight")
    return lo, lo_t


def _rows2d(t: torch.Tensor) -> torch.Tensor:
    """a 2-D operand as the library takes it: unit column stride, rows a constant stride (>= width) apart; else a copy"""
    if t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= t.shape[1]:
        return t
    return t.contiguous()


def gemm(a: torch.Tensor, b: torch.Tensor, trans_a: bool = False, trans_b: bool = True, out_dtype=None,
         epilogue: int = EPI_NONE, bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
         aux: Optional[torch.Tensor] = None, residual_ld: Optional[int] = None, out: Optional[torch.Tensor] = None):
    """C = op(A) op(B) with a fused epilogue.  Returns C (and aux for EPI_BIAS_GELU).  A / B may be row-strided 2-D views
    (leading dimension = their row stride); residual_ld = 0 broadcasts one residual row over all rows (a positional
    table); out: a preallocated, possibly row-strided [M, >= N] fp32 / bf16 view to write into."""
    _need_cuda(a, b, bias, residual, aux)
    lib = _lib.load()
    a = _rows2d(a)
    b = _rows2d(b)
    assert a.dtype == b.dtype
    dt = avf_dtype(a.dtype)
    M, K = (a.shape[1], a.shape[0]) if trans_a else (a.shape[0], a.shape[1])
    N = b.shape[0] if trans_b else b.shape[1]
    Kb = b.shape[1] if trans_b else b.shape[0]
    assert K == Kb, (a.shape, b.shape, trans_a, trans_b)
    cdt = torch_dtype(out_dtype) if out_dtype is not None else a.dtype
    if out is not None:
        assert out.dim() == 2 and out.shape[0] == M and out.shape[1] >= N and out.stride(1) == 1 and out.dtype == cdt
        c = out
    else:
        c = torch.empty((M, N), dtype=cdt, device=a.device)
    made_aux = None
    if epilogue == EPI_BIAS_GELU and aux is None:
        made_aux = aux = torch.empty((M, N), dtype=cdt, device=a.device)
    ws = _bytes(lib.avf_gemm_workspace_bytes(dt, int(trans_a), int(trans_b), M, N, K), a.device)
    if residual is not None and residual_ld is None:
        residual = residual.contiguous()
    _lib.check(lib.avf_gemm(dt, int(trans_a), int(trans_b), M, N, K, _ptr(a), a.stride(0), _ptr(b), b.stride(0),
                            _ptr(c), c.stride(0), avf_dtype(cdt), epilogue, _ptr(bias), _ptr(residual),
                            N if residual_ld is None else int(residual_ld), _ptr(aux), N, _ptr(ws), _stream()), "gemm")
    if made_aux is not None:
        return c, made_aux
    return c


def pack_ws(w: torch.Tensor) -> torch.Tensor:
    """Fragment-major image of a bf16 weight [rows % 256 == 0, 512] for gemm_ws (avf_pack_weight_ws)."""
    _need_cuda(w)
    lib = _lib.load()
    w = _rows2d(w)
    assert w.dtype == torch.bfloat16
    rows, cols = w.shape
    n = lib.avf_pack_weight_ws_bytes(rows, cols)
    if n == 0:
        raise ValueError(f"pack_ws: needs rows % 256 == 0 and cols == 512, got {tuple(w.shape)}")
    out = torch.empty(n // 2, dtype=torch.bfloat16, device=w.device)
    _lib.check(lib.avf_pack_weight_ws(_ptr(w), w.stride(0), rows, cols, _ptr(out), _stream()), "pack_weight_ws")
    return out


def gemm_ws_used(M: int, n_out: int, K: int, epilogue: int = EPI_NONE, out_dtype=torch.bfloat16) -> bool:
    """True when ``gemm`` / the layer calls run this bf16 NT shape on the weight-stationary persistent kernel
    (avf_gemm_nt_ws_dispatch); ``gemm_ws`` itself forces that kernel for every shape it can run."""
    return bool(_lib.load().avf_gemm_nt_ws_dispatch(int(M), int(n_out), int(K), int(epilogue), avf_dtype(torch_dtype(out_dtype))))


def gemm_ws(a: torch.Tensor, w_packed: torch.Tensor, n_out: int, out_dtype=None, epilogue: int = EPI_NONE,
            bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None, aux: Optional[torch.Tensor] = None,
            want_colsum: bool = False, want_image: bool = False):
    """C = epilogue(A W^T) on the weight-stationary persistent kernel (avf_gemm_nt_ws); w_packed = pack_ws(W[n_out, 512]).
    Returns C (and the saved pre-activation for EPI_BIAS_GELU; and the column sums of C with want_colsum; and with want_image
    - EPI_DGELU with want_colsum - the MX-FP8 image (q, scales) of the fp32 values behind C)."""
    _need_cuda(a, w_packed, bias, residual, aux)
    lib = _lib.load()
    a = _rows2d(a)
    M, K = a.shape
    cdt = torch_dtype(out_dtype) if out_dtype is not None else a.dtype
    # the C ABI sees raw pointers: everything it cannot check is checked here
    if a.dtype != torch.bfloat16 or w_packed.dtype != torch.bfloat16:
        raise TypeError(f"gemm_ws: bf16 operands only (a {a.dtype}, w_packed {w_packed.dtype})")
    if K != 512 or w_packed.numel() != n_out * 512 or not w_packed.is_contiguous():
        raise ValueError(f"gemm_ws: a is [M, 512] and w_packed = pack_ws(W[{n_out}, 512]); got K = {K}, image of {w_packed.numel()} elements")
    if cdt not in (torch.bfloat16, torch.float32):
        raise TypeError(f"gemm_ws: C is bf16 or fp32, not {cdt}")
    if bias is not None and (bias.dtype != torch.float32 or bias.numel() != n_out):
        raise TypeError("gemm_ws: bias is fp32 [n_out]")
    for name, t in (("residual", residual), ("aux", aux)):
        if t is not None and (t.dtype != cdt or t.numel() != M * n_out):
            raise TypeError(f"gemm_ws: {name} is stored in C's type {cdt} as [M, n_out] (got {t.dtype}, {tuple(t.shape)})")
    if aux is not None and not aux.is_contiguous():
        raise ValueError("gemm_ws: aux must be contiguous")
    c = torch.empty((M, n_out), dtype=cdt, device=a.device)
    made_aux = None
    if epilogue == EPI_BIAS_GELU and aux is None:
        made_aux = aux = torch.empty((M, n_out), dtype=cdt, device=a.device)
    cs = torch.empty(n_out, dtype=torch.float32, device=a.device) if want_colsum else None
    ws = _bytes(lib.avf_gemm_nt_ws_workspace_bytes(M, n_out), a.device) if want_colsum else None
    if residual is not None:
        residual = residual.contiguous()
    cq = torch.empty((M, n_out), dtype=torch.uint8, device=a.device) if want_image else None
    csc = torch.empty((M, n_out // 32), dtype=torch.uint8, device=a.device) if want_image else None
    _lib.check(lib.avf_gemm_nt_ws(M, n_out, K, _ptr(a), a.stride(0), _ptr(w_packed), _ptr(c), c.stride(0), avf_dtype(cdt),
                                  epilogue, _ptr(bias), _ptr(residual), n_out, _ptr(aux), n_out, _ptr(ws), _ptr(cs), _ptr(cq),
                                  _ptr(csc), _stream()),
               "gemm_nt_ws")
    res = (c,)
    if made_aux is not None:
        res += (made_aux,)
    if want_colsum:
        res += (cs,)
    if want_image:
        res += (cq, csc)
    return res[0] if len(res) == 1 else res


def gemm_tn_group(pairs):
    """[(A_i [K, M_i] bf16, B_i [K, N_i] bf16), ...] (up to 4, same K) -> [C_i [M_i, N_i] fp32 = A_i^T B_i]: the grouped
    weight-gradient launch of a layer's backward (avf_gemm_tn_group)."""
    lib = _lib.load()
    n = len(pairs)
    As = [a.contiguous() for a, _ in pairs]
    Bs = [b.contiguous() for _, b in pairs]
    _need_cuda(*As, *Bs)
    K = As[0].shape[0]
    assert all(a.shape[0] == K and b.shape[0] == K and a.dtype == b.dtype == torch.bfloat16 for a, b in zip(As, Bs))
    Ms = (C.c_int64 * n)(*[a.shape[1] for a in As])
    Ns = (C.c_int64 * n)(*[b.shape[1] for b in Bs])
    Cs = [torch.empty((a.shape[1], b.shape[1]), dtype=torch.float32, device=a.device) for a, b in zip(As, Bs)]
    arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
    ws = _bytes(lib.avf_gemm_tn_group_workspace_bytes(n, K, Ms, Ns), As[0].device)
    _lib.check(lib.avf_gemm_tn_group(n, K, arr(As), arr(Bs), arr(Cs), Ms, Ns, _ptr(ws), _stream()), "gemm_tn_group")
    return Cs


def quant_mx8(x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """[rows, cols] f32|bf16 -> (e4m3 bytes [rows, cols] uint8, E8M0 scale bytes [rows, cols/32] uint8)."""
    _need_cuda(x)
    x = x.contiguous()
    rows, cols = x.shape
    q = torch.empty((rows, cols), dtype=torch.uint8, device=x.device)
    s = torch.empty((rows, cols // 32), dtype=torch.uint8, device=x.device)
    _lib.check(_lib.load().avf_quant_mx8(avf_dtype(x.dtype), _ptr(x), rows, cols, _ptr(q), _ptr(s), _stream()), "quant_mx8")
    return q, s


def layernorm_fwd_mx8(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float = 1e-5):
    """LayerNorm rows -> (y bf16, mean, rstd, e4m3 image of y, its scale bytes)."""
    _need_cuda(x, weight, bias)
    x = x.contiguous()
    rows, dim = x.shape
    y = torch.empty((rows, dim), dtype=torch.bfloat16, device=x.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    q = torch.empty((rows, dim), dtype=torch.uint8, device=x.device)
    s = torch.empty((rows, dim // 32), dtype=torch.uint8, device=x.device)
    _lib.check(_lib.load().avf_layernorm_fwd_mx8(_ptr(x), _ptr(weight), _ptr(bias), _ptr(y), _ptr(mean), _ptr(rstd), _ptr(q),
                                                 _ptr(s), rows, dim, eps, _stream()), "layernorm_fwd_mx8")
    return y, mean, rstd, q, s


def layernorm_bwd_mx8(dy: torch.Tensor, x: torch.Tensor, weight: torch.Tensor, mean: torch.Tensor, rstd: torch.Tensor,
                      dres: Optional[torch.Tensor] = None):
    """LayerNorm backward from a bf16 dy -> (dx fp32, dx bf16, e4m3 image of dx, its scale bytes, dgamma, dbeta)."""
    _need_cuda(dy, x, weight, mean, rstd, dres)
    assert dy.dtype == torch.bfloat16 and x.dtype == torch.float32
    dy, x = dy.contiguous(), x.contiguous()
    rows, dim = x.shape
    lib = _lib.load()
    dev = x.device
    dx = torch.empty((rows, dim), dtype=torch.float32, device=dev)
    dx_lo = torch.empty((rows, dim), dtype=torch.bfloat16, device=dev)
    q = torch.empty((rows, dim), dtype=torch.uint8, device=dev)
    s = torch.empty((rows, dim // 32), dtype=torch.uint8, device=dev)
    dg = torch.empty(dim, dtype=torch.float32, device=dev)
    db = torch.empty(dim, dtype=torch.float32, device=dev)
    ws = torch.empty(max(1, lib.avf_layernorm_bwd_workspace_bytes(rows, dim)), dtype=torch.uint8, device=dev)
    _lib.check(lib.avf_layernorm_bwd_mx8(_ptr(dy), _ptr(x), _ptr(weight), _ptr(mean), _ptr(rstd),
                                         _ptr(dres.contiguous() if dres is not None else None), _ptr(dx), _ptr(dx_lo), _ptr(q),
                                         _ptr(s), _ptr(dg), _ptr(db), _ptr(ws), rows, dim, _stream()), "layernorm_bwd_mx8")
    return dx, dx_lo, q, s, dg, db


def attn_fwd_mx8(qkv: torch.Tensor, batch: int, tokens: int, heads: int, dim_head: int):
    """bf16 attention forward -> (o bf16 [B*N, I], lse2, e4m3 image of o, its scale bytes); dim_head 64, tokens <= 576."""
    _need_cuda(qkv)
    assert qkv.dtype == torch.bfloat16
    qkv = qkv.contiguous()
    inner = heads * dim_head
    dev = qkv.device
    o = torch.empty((batch * tokens, inner), dtype=torch.bfloat16, device=dev)
    lse2 = torch.empty((batch, heads, tokens), dtype=torch.float32, device=dev)
    q = torch.empty((batch * tokens, inner), dtype=torch.uint8, device=dev)
    s = torch.empty((batch * tokens, inner // 32), dtype=torch.uint8, device=dev)
    _lib.check(_lib.load().avf_attn_fwd_mx8(_ptr(qkv), _ptr(o), _ptr(lse2), _ptr(q), _ptr(s), batch, tokens, heads, dim_head,
                                            _stream()), "attn_fwd_mx8")
    return o, lse2, q, s


def attn_bwd_mx8(qkv, o, d_o, lse2, batch: int, tokens: int, heads: int, dim_head: int):
    """bf16 attention backward on PRE-SCALED queries -> (dqkv bf16 [B*N, 3I], its e4m3 image, the scale bytes [B*N, 3I/32]);
    only where the merged kernel runs (avf_attn_bwd_emits_mx8)."""
    _need_cuda(qkv, o, d_o, lse2)
    inner = heads * dim_head
    dev = qkv.device
    dqkv = torch.empty((batch * tokens, 3 * inner), dtype=torch.bfloat16, device=dev)
    q = torch.empty((batch * tokens, 3 * inner), dtype=torch.uint8, device=dev)
    s = torch.empty((batch * tokens, 3 * inner // 32), dtype=torch.uint8, device=dev)
    _lib.check(_lib.load().avf_attn_bwd_mx8(_ptr(qkv.contiguous()), _ptr(o.contiguous()), _ptr(d_o.contiguous()), _ptr(lse2), _ptr(dqkv),
                                            _ptr(q), _ptr(s), batch, tokens, heads, dim_head, _stream()), "attn_bwd_mx8")
    return dqkv, q, s


def gemm_mx8(a_q: torch.Tensor, a_s: torch.Tensor, b_q: torch.Tensor, b_s: torch.Tensor, out_dtype=torch.float32,
             epilogue: int = EPI_NONE, bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
             want_image: bool = False, aux: Optional[torch.Tensor] = None):
    """C[M,N] = A[M,K] B[N,K]^T from MX-FP8 images (quant_mx8).  Returns C (and the saved pre-activation for
    EPI_BIAS_GELU; and with want_image the MX-FP8 image (q, scales) of C).  EPI_DGELU reads ``aux`` (the saved
    pre-activation, in C's type)."""
    _need_cuda(a_q, a_s, b_q, b_s, bias, residual, aux)
    M, K = a_q.shape
    N = b_q.shape[0]
    assert b_q.shape[1] == K and a_s.shape == (M, K // 32) and b_s.shape == (N, K // 32)
    c = torch.empty((M, N), dtype=out_dtype, device=a_q.device)
    if epilogue == EPI_DGELU:
        assert aux is not None and aux.shape == (M, N) and aux.dtype == out_dtype
        aux_in, aux = aux.contiguous(), None
    else:
        aux_in = None
        aux = torch.empty((M, N), dtype=out_dtype, device=a_q.device) if epilogue == EPI_BIAS_GELU else None
    cq = torch.empty((M, N), dtype=torch.uint8, device=a_q.device) if want_image else None
    cs = torch.empty((M, N // 32), dtype=torch.uint8, device=a_q.device) if want_image else None
    _lib.check(_lib.load().avf_gemm_mx8_nt(M, N, K, _ptr(a_q.contiguous()), _ptr(a_s.contiguous()), _ptr(b_q.contiguous()),
                                           _ptr(b_s.contiguous()), _ptr(c), N, avf_dtype(out_dtype), epilogue, _ptr(bias),
                                           _ptr(residual.contiguous() if residual is not None else None), N,
                                           _ptr(aux if aux is not None else aux_in), N, _ptr(cq), _ptr(cs), _stream()),
               "gemm_mx8_nt")
    out = (c, aux) if aux is not None else (c,)
    if want_image:
        out = out + (cq, cs)
    return out if len(out) > 1 else out[0]


def attn_fwd(qkv: torch.Tensor, batch: int, tokens: int, heads: int, dim_head: int, q_prescaled: bool = False):
    """Attention core on the packed QKV projection [B*N, 3*H*dh] -> (o [B*N, H*dh], lse2 [B,H,N]).
    q_prescaled (bf16 only): the q columns already carry log2(e)/sqrt(dim_head) (avf_attn_fwd_qs)."""
    _need_cuda(qkv)
    qkv = qkv.contiguous()
    inner = heads * dim_head
    assert qkv.shape == (batch * tokens, 3 * inner)
    o = torch.empty((batch * tokens, inner), dtype=qkv.dtype, device=qkv.device)
    lse2 = torch.empty((batch, heads, tokens), dtype=torch.float32, device=qkv.device)
    if q_prescaled:
        assert qkv.dtype == torch.bfloat16
        _lib.check(_lib.load().avf_attn_fwd_qs(_ptr(qkv), _ptr(o), _ptr(lse2), batch, tokens, heads, dim_head, _stream()),
                   "attn_fwd_qs")
    else:
        _lib.check(_lib.load().avf_attn_fwd(avf_dtype(qkv.dtype), _ptr(qkv), _ptr(o), _ptr(lse2), batch, tokens, heads,
                                            dim_head, _stream()), "attn_fwd")
    return o, lse2


def attn_bwd(qkv, o, d_o, lse2, batch: int, tokens: int, heads: int, dim_head: int, q_prescaled: bool = False) -> torch.Tensor:
    _need_cuda(qkv, o, d_o, lse2)
    lib = _lib.load()
    qkv, o, d_o = qkv.contiguous(), o.contiguous(), d_o.contiguous()
    dqkv = torch.empty_like(qkv)
    ws = _bytes(lib.avf_attn_bwd_workspace_bytes(batch, tokens, heads, dim_head) * (2 if q_prescaled else 1), qkv.device)
    if q_prescaled:
        assert qkv.dtype == torch.bfloat16
        _lib.check(lib.avf_attn_bwd_qs(_ptr(qkv), _ptr(o), _ptr(d_o), _ptr(lse2), _ptr(dqkv), _ptr(ws), batch, tokens, heads,
                                       dim_head, _stream()), "attn_bwd_qs")
    else:
        _lib.check(lib.avf_attn_bwd(avf_dtype(qkv.dtype), _ptr(qkv), _ptr(o), _ptr(d_o), _ptr(lse2), _ptr(dqkv), _ptr(ws),
                                    batch, tokens, heads, dim_head, _stream()), "attn_bwd")
    return dqkv


def attn_fwd_masked(qkv: torch.Tensor, keep: torch.Tensor, batch: int, tokens: int, heads: int, dim_head: int):
    """attn_fwd with the token mask of heads.py:225-232: keep [B, N] uint8 / bool, 1 = token kept (the reference's mask after
    its leading-True pad).  fp32 arithmetic on fp32 or bf16 storage."""
    _need_cuda(qkv, keep)
    qkv = qkv.contiguous()
    keep = keep.to(torch.uint8).contiguous()
    inner = heads * dim_head
    assert qkv.shape == (batch * tokens, 3 * inner) and keep.shape == (batch, tokens)
    o = torch.empty((batch * tokens, inner), dtype=qkv.dtype, device=qkv.device)
    lse2 = torch.empty((batch, heads, tokens), dtype=torch.float32, device=qkv.device)
    _lib.check(_lib.load().avf_attn_fwd_masked(avf_dtype(qkv.dtype), _ptr(qkv), _ptr(o), _ptr(lse2), _ptr(keep), batch, tokens,
                                               heads, dim_head, _stream()), "attn_fwd_masked")
    return o, lse2


def attn_bwd_masked(qkv, o, d_o, lse2, keep, batch: int, tokens: int, heads: int, dim_head: int) -> torch.Tensor:
    _need_cuda(qkv, o, d_o, lse2, keep)
    lib = _lib.load()
    qkv, o, d_o = qkv.contiguous(), o.contiguous(), d_o.contiguous()
    keep = keep.to(torch.uint8).contiguous()
    dqkv = torch.empty_like(qkv)
    ws = _bytes(lib.avf_attn_bwd_workspace_bytes(batch, tokens, heads, dim_head), qkv.device)
    _lib.check(lib.avf_attn_bwd_masked(avf_dtype(qkv.dtype), _ptr(qkv), _ptr(o), _ptr(d_o), _ptr(lse2), _ptr(dqkv), _ptr(ws),
                                       _ptr(keep), batch, tokens, heads, dim_head, _stream()), "attn_bwd_masked")
    return dqkv


# ---- token producers / consumers either side of the stack (csrc/heads.hip) ---------------------------------------
def bn1d_fwd(x, gamma, beta, running_mean, running_var, num_batches_tracked, eps: float, momentum: float, training: bool):
    """nn.BatchNorm1d on [B, C] -> (y, mean, invstd); training updates the running statistics in place"""
    _need_cuda(x, gamma, beta, running_mean, running_var)
    x = x.contiguous()
    B, Cn = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(Cn, dtype=torch.float32, device=x.device)
    invstd = torch.empty(Cn, dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().avf_bn1d_fwd(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var),
                                        _ptr(num_batches_tracked), _ptr(y), _ptr(mean), _ptr(invstd), B, Cn, float(eps),
                                        float(momentum), int(training), _stream()), "bn1d_fwd")
    return y, mean, invstd


def bn1d_bwd(x, dy, gamma, mean, invstd, training: bool, need_dx: bool = True):
    """-> (dx or None, dgamma, dbeta)"""
    _need_cuda(x, dy, gamma, mean, invstd)
    x, dy = x.contiguous(), dy.contiguous()
    B, Cn = x.shape
    dx = torch.empty_like(x) if need_dx else None
    dg = torch.empty(Cn, dtype=torch.float32, device=x.device)
    db = torch.empty(Cn, dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().avf_bn1d_bwd(_ptr(x), _ptr(dy), _ptr(gamma), _ptr(mean), _ptr(invstd), _ptr(dx), _ptr(dg), _ptr(db),
                                        B, Cn, int(training), _stream()), "bn1d_bwd")
    return dx, dg, db


def token_dots_fwd(tokens: torch.Tensor, w: torch.Tensor, pad_to: Optional[int] = None) -> torch.Tensor:
    """tokens [B,T,E] . w [T,E] (row-strided ok) -> [B, pad_to or T], columns T.. zero"""
    _need_cuda(tokens, w)
    tokens = tokens.contiguous()
    B, T, E = tokens.shape
    w = _rows2d(w)
    width = pad_to or T
    out = torch.empty((B, width), dtype=torch.float32, device=tokens.device)
    _lib.check(_lib.load().avf_token_dots_fwd(_ptr(tokens), _ptr(w), w.stride(0), _ptr(out), width, width, B, T, E, _stream()),
               "token_dots_fwd")
    return out


def token_dots_bwd(dout: torch.Tensor, tokens: torch.Tensor, w: torch.Tensor, need_dtokens=True, need_dw=True):
    _need_cuda(dout, tokens, w)
    tokens = tokens.contiguous()
    dout = _rows2d(dout)
    B, T, E = tokens.shape
    w = _rows2d(w)
    dtok = torch.empty_like(tokens) if need_dtokens else None
    dw = torch.empty((T, E), dtype=torch.float32, device=tokens.device) if need_dw else None
    _lib.check(_lib.load().avf_token_dots_bwd(_ptr(dout), dout.stride(0), _ptr(tokens), _ptr(w), w.stride(0), _ptr(dtok), _ptr(dw),
                                              E, B, T, E, _stream()), "token_dots_bwd")
    return dtok, dw


def assemble_tokens(x: torch.Tensor, lead: Optional[torch.Tensor], pos: Optional[torch.Tensor]) -> torch.Tensor:
    """[B,P,D] (+ lead rows [n_lead,D] in front) + pos[n_lead+P, D] -> [B, n_lead+P, D]"""
    _need_cuda(x, lead, pos)
    x = x.contiguous()
    B, P, D = x.shape
    n_lead = 0 if lead is None else lead.numel() // D
    out = torch.empty((B, P + n_lead, D), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().avf_assemble_tokens(_ptr(x), _ptr(lead.contiguous() if lead is not None else None),
                                               _ptr(pos.contiguous() if pos is not None else None), _ptr(out), B, P, n_lead, D,
                                               _stream()), "assemble_tokens")
    return out


def cat_features(a: torch.Tensor, v: torch.Tensor, pos: Optional[torch.Tensor]) -> torch.Tensor:
    """[B,T,Ea] ++ [B,T,Ev] on the feature axis + pos[T, Ea+Ev]"""
    _need_cuda(a, v, pos)
    a, v = a.contiguous(), v.contiguous()
    B, T, Ea = a.shape
    Ev = v.shape[2]
    out = torch.empty((B, T, Ea + Ev), dtype=torch.float32, device=a.device)
    _lib.check(_lib.load().avf_cat_features(_ptr(a), _ptr(v), _ptr(pos.contiguous() if pos is not None else None), _ptr(out), B, T,
                                            Ea, Ev, _stream()), "cat_features")
    return out


def transpose_add(x: torch.Tensor, pos: Optional[torch.Tensor]) -> torch.Tensor:
    """[B, R, C] -> [B, C, R] (+ pos [C, R])"""
    _need_cuda(x, pos)
    x = x.contiguous()
    B, R, Cn = x.shape
    out = torch.empty((B, Cn, R), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().avf_transpose_add(_ptr(x), _ptr(pos.contiguous() if pos is not None else None), _ptr(out), B, R, Cn,
                                             _stream()), "transpose_add")
    return out


def linear_pad_fwd(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], width: int) -> torch.Tensor:
    """[B, K] x [O, K]^T + b into a zero-padded [B, width] row, one launch (K % 4 == 0, K <= 4096)"""
    _need_cuda(x, w, b)
    x, w = x.contiguous(), w.contiguous()
    B, K = x.shape
    out = torch.empty((B, width), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().avf_linear_pad_fwd(_ptr(x), _ptr(w), _ptr(b), _ptr(out), B, K, w.shape[0], width, _stream()),
               "linear_pad_fwd")
    return out


def linear_pad_bwd(dout: torch.Tensor, x: torch.Tensor, w: torch.Tensor, need_dx=True, need_dw=True, need_db=True):
    """-> (dx [B, K], dw [O, K], db [O]) of linear_pad_fwd from the gradient of the padded row (read in place)"""
    _need_cuda(dout, x, w)
    assert dout.dim() == 2 and dout.stride(1) == 1 and dout.dtype == torch.float32
    B, K = x.shape
    O = w.shape[0]
    dev = x.device
    dx = torch.empty((B, K), dtype=torch.float32, device=dev) if need_dx else None
    dw = torch.empty((O, K), dtype=torch.float32, device=dev) if need_dw else None
    db = torch.empty(O, dtype=torch.float32, device=dev) if need_db else None
    _lib.check(_lib.load().avf_linear_pad_bwd(_ptr(dout), dout.stride(0), _ptr(x), _ptr(w), _ptr(dx), _ptr(dw), _ptr(db), B, K, O,
                                              _stream()), "linear_pad_bwd")
    return dx, dw, db


def zero_cols(t: torch.Tensor, c0: int, c1: int):
    _need_cuda(t)
    assert t.dim() == 2 and t.stride(1) == 1 and t.dtype == torch.float32
    _lib.check(_lib.load().avf_zero_cols(_ptr(t), t.stride(0), t.shape[0], c0, c1, _stream()), "zero_cols")


def au_loss(logits: torch.Tensor, labels: torch.Tensor, pos_weight: torch.Tensor, ignore: float = -1.0):
    """-> (loss scalar tensor, dloss/dlogits [rows, ncls]).  Reference models/loss.py:75-103."""
    _need_cuda(logits, labels, pos_weight)
    assert logits.dim() == 2 and labels.dim() == 2 and logits.shape == labels.shape
    assert logits.stride(1) == 1 and labels.stride(1) == 1 and logits.dtype == torch.float32
    labels = labels.to(torch.float32)
    rows, ncls = logits.shape
    loss = torch.empty((), dtype=torch.float32, device=logits.device)
    grad = torch.empty((rows, ncls), dtype=torch.float32, device=logits.device)
    _lib.check(_lib.load().avf_au_loss(_ptr(logits), logits.stride(0), _ptr(labels), labels.stride(0), _ptr(pos_weight),
                                       float(ignore), rows, ncls, _ptr(loss), _ptr(grad), _stream()), "au_loss")
    return loss, grad


def au_loss_sum(logits: torch.Tensor, labels: torch.Tensor, pos_weight: torch.Tensor, ignore: float = -1.0):
    """-> ([sum over kept rows of the row-mean BCE, kept rows] as a 2-vector, d sum / d logits [rows, ncls]): the
    numerator / denominator of loss.py:85-102, for ranks that hold different numbers of ignored rows."""
    _need_cuda(logits, labels, pos_weight)
    assert logits.dim() == 2 and labels.dim() == 2 and logits.shape == labels.shape
    assert logits.stride(1) == 1 and labels.stride(1) == 1 and logits.dtype == torch.float32
    labels = labels.to(torch.float32)
    rows, ncls = logits.shape
    sc = torch.empty(2, dtype=torch.float32, device=logits.device)
    grad = torch.empty((rows, ncls), dtype=torch.float32, device=logits.device)
    _lib.check(_lib.load().avf_au_loss_sum(_ptr(logits), logits.stride(0), _ptr(labels), labels.stride(0), _ptr(pos_weight),
                                           float(ignore), rows, ncls, _ptr(sc), _ptr(grad), _stream()), "au_loss_sum")
    return sc, grad


def au_loss_wide(out: torch.Tensor, labels: torch.Tensor, pos_weight: torch.Tensor, ignore: float = -1.0, sum_mode: bool = False):
    """AULoss on the first ``labels.shape[1]`` slots of the model's output rows ``out`` [rows, width] -> (loss scalar, or the
    (sum, count) 2-vector with sum_mode; d / d out [rows, width], zero beyond the logits) - avf_au_loss_wide."""
    _need_cuda(out, labels, pos_weight)
    assert out.dim() == 2 and labels.dim() == 2 and out.shape[0] == labels.shape[0] and out.shape[1] >= labels.shape[1]
    assert out.stride(1) == 1 and labels.stride(1) == 1 and out.dtype == torch.float32
    labels = labels.to(torch.float32)
    rows, width = out.shape
    ncls = labels.shape[1]
    loss = torch.empty(2 if sum_mode else (), dtype=torch.float32, device=out.device)
    grad = torch.empty((rows, width), dtype=torch.float32, device=out.device)
    _lib.check(_lib.load().avf_au_loss_wide(_ptr(out), out.stride(0), _ptr(labels), labels.stride(0), _ptr(pos_weight), float(ignore),
                                            rows, ncls, width, 1 if sum_mode else 0, _ptr(loss), _ptr(grad), _stream()), "au_loss_wide")
    return loss, grad


def fuse_tokens(clip: torch.Tensor, audio: torch.Tensor, pos: Optional[torch.Tensor], out_bf16: bool = False) -> torch.Tensor:
    """[B,Tv,D] ++ [B,Ta,D] on the token axis, + pos[Tv+Ta, D] (nullable): one pass (avf_fuse_tokens); out_bf16: the
    result is written in bf16 (the storage type of a bf16 residual stream, avf_fuse_tokens_bf16)."""
    _need_cuda(clip, audio)
    clip, audio = clip.contiguous(), audio.contiguous()
    B, Tv, D = clip.shape
    Ta = audio.shape[1]
    assert audio.shape[0] == B and audio.shape[2] == D and clip.dtype == audio.dtype == torch.float32
    if pos is not None:
        pos = pos.contiguous()
        assert pos.numel() == (Tv + Ta) * D and pos.dtype == torch.float32
    out = torch.empty((B, Tv + Ta, D), dtype=torch.bfloat16 if out_bf16 else torch.float32, device=clip.device)
    fn = _lib.load().avf_fuse_tokens_bf16 if out_bf16 else _lib.load().avf_fuse_tokens
    _lib.check(fn(_ptr(clip), _ptr(audio), _ptr(pos), _ptr(out), B, Tv, Ta, D, _stream()), "fuse_tokens")
    return out


def token_mean_fwd(y: torch.Tensor) -> torch.Tensor:
    """[B,T,D] fp32 (or bf16: the bf16 residual stream) -> [B,D] fp32 mean over tokens."""
    _need_cuda(y)
    y = y.contiguous()
    B, T, D = y.shape
    out = torch.empty((B, D), dtype=torch.float32, device=y.device)
    fn = _lib.load().avf_token_mean_fwd_bf16 if y.dtype == torch.bfloat16 else _lib.load().avf_token_mean_fwd
    _lib.check(fn(_ptr(y), _ptr(out), B, T, D, _stream()), "token_mean_fwd")
    return out


def token_mean_bwd(g: torch.Tensor, tokens: int, want_bf16: bool = False, want_colsum: bool = False):
    """g [B,D] -> dy [B,T,D] = g/T broadcast (+ optional bf16 copy and column sums over all B*T rows)."""
    _need_cuda(g)
    g = g.contiguous().to(torch.float32)
    B, D = g.shape
    dy = torch.empty((B, tokens, D), dtype=torch.float32, device=g.device)
    lo = torch.empty((B, tokens, D), dtype=torch.bfloat16, device=g.device) if want_bf16 else None
    cs = torch.empty(D, dtype=torch.float32, device=g.device) if want_colsum else None
    _lib.check(_lib.load().avf_token_mean_bwd(_ptr(g), _ptr(dy), _ptr(lo), _ptr(cs), B, tokens, D, _stream()),
               "token_mean_bwd")
    return dy, lo, cs


def dropout_factors(seed: int, layer: int, site: int, p: float, rows: int, cols: int, device="cuda") -> torch.Tensor:
    """keep/(1-p) factors of dropout site `site` of layer `layer` for a [rows, cols] activation (test aid)."""
    out = torch.empty((rows, cols), dtype=torch.float32, device=device)
    _lib.check(_lib.load().avf_dropout_factors(seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF, layer, site, float(p), rows,
                                               cols, _ptr(out), _stream()), "dropout_factors")
    return out


# hardware self-tests -------------------------------------------------------------------------------
def selftest_mfma_bf16(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    c = torch.empty((16, 16), dtype=torch.float32, device=a.device)
    _lib.check(_lib.load().avf_selftest_mfma_bf16(_ptr(a.contiguous()), _ptr(b.contiguous()), _ptr(c), _stream()), "selftest")
    return c


def selftest_mfma_f32(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    c = torch.empty((16, 16), dtype=torch.float32, device=a.device)
    _lib.check(_lib.load().avf_selftest_mfma_f32(_ptr(a.contiguous()), _ptr(b.contiguous()), _ptr(c), _stream()), "selftest")
    return c


def selftest_tr16(tile: torch.Tensor) -> torch.Tensor:
    out = torch.empty((64, 8), dtype=torch.bfloat16, device=tile.device)
    _lib.check(_lib.load().avf_selftest_tr16(_ptr(tile.contiguous()), _ptr(out), _stream()), "selftest")
    return out
