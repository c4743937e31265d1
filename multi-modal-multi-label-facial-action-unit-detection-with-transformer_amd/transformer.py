"""``Transformer`` - drop-in for the reference's pre-norm transformer stack, computed by HIP kernels.

Mirrors ``Transformer(dim, depth, heads, dim_head, mlp_dim, dropout=0.)`` of the reference
(models/heads.py:242-256 and its byte-identical copies in sformer.py/tformer.py/vformer.py/
dual_sformer.py/vggformer.py): same constructor, same ``forward(x[B,N,dim], mask=None)``, same
``state_dict`` keys and default initialisation (it instantiates the same ``nn.Linear`` /
``nn.LayerNorm`` holders in the same order, so a given ``torch.manual_seed`` yields the weights
the reference would get).  The holders only own parameters: all math runs in
``libavformer_hip.so`` through one forward and one backward C call per layer.

Extra keyword ``compute_dtype``: ``"mx8"`` (the bf16 mode with MX-FP8 operands on the forward GEMMs of to_qkv, to_out, net.0
and net.3 and the backward dX GEMMs of net.3, net.0 and to_out - BASELINE config 5; tolerance against the bf16 mode stated in tests/test_gpu_mx8.py), ``"bf16"`` (throughput mode: bf16 MFMA, fp32 accumulate / LayerNorm /
softmax statistics / residual stream) or ``"f32"`` (parity mode: fp32 MFMA + fp32 attention; matches the
fp32 CPU reference to ~1e-5).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Callable, List, Optional

import torch
from torch import nn

from . import _lib, ops
from .ops import avf_dtype

PARAMS_PER_LAYER = 11

# AVF_DEBUG_CANARY=1: every byte buffer handed to the library (saved activations, workspace, bf16 weights) gets a
# 4 KiB guard band filled with 0xA5 behind it, verified after each forward/backward - catches writes past the end of a
# carved region (tests/test_gpu_transformer.py::test_no_out_of_bounds_writes).
import os as _os

_CANARY = 4096 if _os.environ.get("AVF_DEBUG_CANARY") == "1" else 0
_guarded: list = []  # weak references to the guarded buffers that are still alive


def _alloc_bytes(nbytes: int, dev) -> torch.Tensor:
    t = torch.empty(max(int(nbytes), 16) + _CANARY, dtype=torch.uint8, device=dev)
    if _CANARY:
        import weakref
        t[-_CANARY:].fill_(0xA5)
        _guarded.append(weakref.ref(t))
    return t


def _check_canaries():
    if not _CANARY:
        return
    torch.cuda.synchronize()
    alive = []
    for r in _guarded:
        t = r()
        if t is None:
            continue
        alive.append(r)
        if not bool((t[-_CANARY:] == 0xA5).all()):
            raise RuntimeError(f"libavformer_hip wrote past the end of a {t.numel() - _CANARY}-byte buffer")
    _guarded[:] = alive


class _Holder(nn.Module):
    """Parameter container; mirrors the reference's wrapper nesting so state_dict keys match."""

    def forward(self, *a, **k):  # pragma: no cover - never called
        raise RuntimeError("parameter holder: the computation runs in the enclosing Transformer")


def _make_layer(dim: int, heads: int, dim_head: int, mlp_dim: int, dropout: float) -> nn.ModuleList:
    inner = heads * dim_head
    project_out = not (heads == 1 and dim_head == dim)  # reference heads.py:207
    # construction order == RNG order of the reference: to_qkv, to_out, (LayerNorm), net.0, net.3, (LayerNorm)
    attn = _Holder()
    attn.to_qkv = nn.Linear(dim, inner * 3, bias=False)
    attn.to_out = nn.Sequential(nn.Linear(inner, dim), nn.Dropout(dropout)) if project_out else nn.Identity()
    pre_a = _Holder()
    pre_a.norm = nn.LayerNorm(dim)
    pre_a.fn = attn
    res_a = _Holder()
    res_a.fn = pre_a
    ff = _Holder()
    ff.net = nn.Sequential(nn.Linear(dim, mlp_dim), nn.Identity(), nn.Dropout(dropout), nn.Linear(mlp_dim, dim),
                           nn.Dropout(dropout))
    pre_f = _Holder()
    pre_f.norm = nn.LayerNorm(dim)
    pre_f.fn = ff
    res_f = _Holder()
    res_f.fn = pre_f
    return nn.ModuleList([res_a, res_f])


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class _StackFn(torch.autograd.Function):
    """x -> L layers.  Saved activations live in per-layer byte buffers carved by the library."""

    @staticmethod
    def forward(ctx, x, audio, pos, mod, pool, keep, *params):
        """audio / pos not None: x is the video token tensor and the stack's input is the fused sequence
        cat([x, audio], 1) + pos (avformer.py:95-103 on the sequence axis), built by the library in the residual stream's
        storage type; backward then hands d clip / d audio / d pos back from the fp32 gradient of that sequence."""
        lib = _lib.load()
        ctx.fused = audio is not None
        if ctx.fused:
            T = x.shape[1] + audio.shape[1]
            if pos.shape[-2] < T or pos.shape[-1] != x.shape[-1]:
                raise ValueError(f"pos_embedding {tuple(pos.shape)} does not cover {T} tokens of width {x.shape[-1]}")
            ctx.tv, ctx.shape_pos = x.shape[1], pos.shape
            x = ops.fuse_tokens(x.detach().float(), audio.detach().float(),
                                pos.detach().reshape(pos.shape[-2], pos.shape[-1])[:T], out_bf16=mod.resid_bf16)
        B, N, D = x.shape
        dev = x.device
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        L = mod.depth
        seed_t = mod._advance_seed(dev)  # None unless dropout is live; a device tensor otherwise (graph-replayable)
        if keep is not None and keep.shape != (B, N):
            raise ValueError(f"mask has incorrect dimensions: {tuple(keep.shape)} after the leading True, tokens {(B, N)}")
        ctx.keep = keep  # token mask bytes (heads.py:225-232); the layer configurations carry its device pointer
        # the L layer configurations are rebuilt only when something in them changes (shape, live dropout, seed / mask pointers)
        ckey = (B, N, mod.training, mod.dropout, None if seed_t is None else seed_t.data_ptr(),
                None if keep is None else keep.data_ptr(), mod.compute_dtype, mod.mx8, mod.mx8_bwd, mod.resid_bf16)
        chit = mod.__dict__.get("_cfg_cache")
        if chit is not None and chit[0] == ckey:
            cfgs = chit[1]
        else:
            cfgs = [mod._cfg(B, N, l, seed_t, keep) for l in range(L)]
            mod.__dict__["_cfg_cache"] = (ckey, cfgs)
        cfg = cfgs[0]
        params = list(params)  # (grad mode is off inside Function.forward: the Parameters are used for their storage only)
        need_grad = any(ctx.needs_input_grad)  # (grad mode is off inside Function.forward)
        ctx.grad_in = ctx.needs_input_grad[:3]
        saved_bytes = lib.avf_layer_saved_bytes(C.byref(cfg))
        if saved_bytes == 0:
            _lib.check(1, "layer configuration")
        ws = mod._workspace(lib, cfg, dev)
        lowps = mod._lowp(lib, cfg, params, dev, stream)
        rs16 = bool(cfg.resid_bf16)
        x0 = x.detach().contiguous().view(B * N, D)
        if rs16 and x0.dtype != torch.bfloat16:  # the stream enters in bf16 (a caller may hand it over in bf16 already)
            xb = torch.empty((B * N, D), dtype=torch.bfloat16, device=dev)
            _lib.check(lib.avf_cast_f32_to_bf16(_ptr(x0), _ptr(xb), B * N * D, stream), "cast")
            x0 = xb
        xs = [x0]
        saved = []
        shared = None
        for l in range(L):
            pp = mod._param_struct(params, l)
            if need_grad:
                sv = _alloc_bytes(saved_bytes, dev)
            else:
                shared = shared if shared is not None else _alloc_bytes(saved_bytes, dev)
                sv = shared
            x_out = torch.empty((B * N, D), dtype=torch.bfloat16 if rs16 else torch.float32, device=dev)
            _lib.check(lib.avf_layer_fwd(C.byref(cfgs[l]), C.byref(pp), _ptr(lowps[l]), _ptr(xs[-1]), _ptr(x_out), _ptr(sv),
                                         _ptr(ws), stream), f"layer_fwd[{l}]")
            saved.append(sv)
            xs.append(x_out)
        ctx.mod = mod
        ctx.cfgs = cfgs
        ctx.seed_t = seed_t  # the kernels of backward read the same device word
        ctx.shape = (B, N, D)
        ctx.xs = xs[:-1] if need_grad else None
        ctx.saved_bufs = saved if need_grad else None
        ctx.params = params
        ctx.lowps = lowps
        ctx.pool = pool
        _check_canaries()
        if pool:  # token-mean pooling of the last layer's output, fused on the library side (heads that pool)
            pooled = torch.empty((B, D), dtype=torch.float32, device=dev)
            fn = lib.avf_token_mean_fwd_bf16 if rs16 else lib.avf_token_mean_fwd
            _lib.check(fn(_ptr(xs[-1]), _ptr(pooled), B, N, D, stream), "token_mean_fwd")
            return pooled
        out = xs[-1].view(B, N, D)
        return out.float() if rs16 else out  # the caller's interface stays fp32

    @staticmethod
    def backward(ctx, dy):
        with torch.cuda.device(dy.device):
            return _StackFn._backward(ctx, dy)

    @staticmethod
    def _backward(ctx, dy):
        if ctx.saved_bufs is None:
            raise RuntimeError("Transformer (HIP): backward through the stack a second time - its saved activations are released "
                               "after the first backward (retain_graph=True is not supported; run the forward again)")
        lib = _lib.load()
        mod = ctx.mod
        cfgs = ctx.cfgs
        cfg = cfgs[0]
        B, N, D = ctx.shape
        dev = dy.device
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        L = mod.depth
        ws = mod._workspace(lib, cfg, dev)
        bf16 = cfg.dtype == _lib.BF16
        # bf16 images of the gradient stream handed from layer to layer (mx8 backward: with the MX-FP8 image behind)
        lo_bytes = lib.avf_layer_grad_stream_bytes(C.byref(cfg)) if bf16 else 0
        lo_a = torch.empty(lo_bytes, dtype=torch.uint8, device=dev) if bf16 else None
        lo_b = torch.empty(lo_bytes, dtype=torch.uint8, device=dev) if bf16 else None
        have_lo = False
        top_colsum = False
        if ctx.pool:
            # dy is the gradient of the pooled [B, D] output: one kernel writes the top layer's incoming gradient in fp32
            # and (no dropout mask to apply on it) in bf16, and its column sums - the top layer's b2 gradient - directly
            g = dy.contiguous().to(torch.float32)
            dx = torch.empty((B * N, D), dtype=torch.float32, device=dev)
            have_lo = top_colsum = bf16 and cfg.dropout_p == 0.0
        else:
            dx = dy.contiguous().view(B * N, D).to(torch.float32)
            if dx.data_ptr() == dy.data_ptr():
                dx = dx.clone()  # layers write their input gradient in place
        live = mod.flat_parameters()
        sizes = [p.numel() for p in ctx.params[:PARAMS_PER_LAYER]]
        hook = mod._grad_hook
        # one flat fp32 gradient bucket per layer; the tensors' .grad become views of it.  The buckets of all layers are
        # consecutive slices of ONE allocation, so a data-parallel wrapper can reduce several adjacent layers with one
        # collective (fewer, larger all-reduces: less host time per step, better xGMI efficiency)
        per_layer = sum(sizes)
        flat_all = torch.empty(L * per_layer, dtype=torch.float32, device=dev)
        # (one split call for all L * 11 views, a reshape only for the four matrices of a layer, gradient pointers by
        # arithmetic on the bucket's base address: this loop runs on the autograd thread once per step per stack)
        shapes = [p.shape for p in ctx.params[:PARAMS_PER_LAYER]]
        parts = flat_all.split_with_sizes(sizes * L)
        views = [[parts[l * PARAMS_PER_LAYER + i] if len(shapes[i]) == 1 else parts[l * PARAMS_PER_LAYER + i].view(shapes[i])
                  for i in range(PARAMS_PER_LAYER)] for l in range(L)]
        flats = list(flat_all.split(per_layer))
        offs, acc = [], 0
        for n in sizes:
            offs.append(4 * acc)
            acc += n
        base = flat_all.data_ptr()
        B2 = PARAMS_PER_LAYER - 1  # index of net.3.bias: its gradient = column sums of the layer's dx_out
        gs16 = bool(cfg.grad_stream_bf16)
        if ctx.pool:
            _lib.check(lib.avf_token_mean_bwd(_ptr(g), None if (gs16 and have_lo) else _ptr(dx),
                                              _ptr(lo_a) if have_lo else None,
                                              _ptr(views[L - 1][B2]) if top_colsum else None, B, N, D, stream),
                       "token_mean_bwd")
        def hand_over(l):
            waiter = hook(l, flats[l]) if hook is not None else None  # e.g. launch this layer's all-reduce now
            # Parameter gradients are handed over directly (views of the layer's flat buffer, no copy):
            # ``.grad = view`` when empty, ``.grad += view`` when accumulating.
            for p, v in zip(live[l * PARAMS_PER_LAYER:(l + 1) * PARAMS_PER_LAYER], views[l]):
                if not p.requires_grad:
                    continue
                if p.grad is None:
                    p.grad = v
                else:
                    if waiter is not None:
                        waiter()  # accumulation needs the reduced values
                    p.grad.add_(v)

        for l in reversed(range(L)):
            gp = _lib.LayerPtrs(*[base + 4 * l * per_layer + o for o in offs])
            pp = mod._param_struct(ctx.params, l)
            # LN1' of this layer writes the column sums of dx_in directly into the previous layer's b2 gradient
            # bf16 gradient stream: the fp32 buffer carries the gradient only into the top layer (when no bf16 image came
            # with it) and out of the bottom one; in between the layers hand each other the bf16 image alone
            dx_out_p = None if (gs16 and have_lo) else _ptr(dx)
            dx_in_p = None if (gs16 and l > 0) else _ptr(dx)
            _lib.check(lib.avf_layer_bwd(C.byref(cfgs[l]), C.byref(pp), _ptr(ctx.lowps[l]), _ptr(ctx.xs[l]),
                                         _ptr(ctx.saved_bufs[l]), dx_out_p, _ptr(lo_a) if have_lo else None,
                                         _ptr(views[l][B2]) if (l < L - 1 or top_colsum) else None, dx_in_p, _ptr(lo_b),
                                         _ptr(views[l - 1][B2]) if l > 0 else None, C.byref(gp), _ptr(ws), stream),
                       f"layer_bwd[{l}]")
            lo_a, lo_b = lo_b, lo_a
            have_lo = bf16
            hand_over(l)
        _check_canaries()
        ctx.saved_bufs = None
        ctx.xs = None
        tail = (None, None, None, *([None] * (L * PARAMS_PER_LAYER)))
        dx = dx.view(B, N, D)
        if not ctx.fused:
            return (dx, None, None) + tail
        d_clip = dx[:, :ctx.tv] if ctx.grad_in[0] else None
        d_audio = dx[:, ctx.tv:] if ctx.grad_in[1] else None
        d_pos = None
        if ctx.grad_in[2]:  # d pos = sum over the clips (fp32 column sums of the [B, T*D] view)
            d_pos = ops.colsum(dx.view(B, N * D)).view(N, D)
            if ctx.shape_pos[-2] > N:  # embedding table longer than the sequence: the unused rows get zero gradient
                d_pos = torch.nn.functional.pad(d_pos, (0, 0, 0, ctx.shape_pos[-2] - N))
            d_pos = d_pos.view(ctx.shape_pos)
        return (d_clip, d_audio, d_pos) + tail


class Transformer(nn.Module):
    """MI355X-native ``Transformer`` (reference models/heads.py:242-256)."""
    _instances = 0

    def __init__(self, dim, depth, heads, dim_head, mlp_dim, dropout=0., compute_dtype="bf16", residual_dtype="f32"):
        super().__init__()
        self.dim, self.depth, self.heads, self.dim_head, self.mlp_dim = dim, depth, heads, dim_head, mlp_dim
        self.dropout = float(dropout)
        # "mx8": the bf16 path with MX-FP8 operands for the forward GEMMs of to_qkv, net.0 and net.3 (BASELINE config 5)
        # ("mx8-fwd": forward GEMMs only, the round-1 form of the mode - kept addressable for A/B runs and its tests)
        cd = compute_dtype.lower().replace("_", "-") if isinstance(compute_dtype, str) else ""
        self.mx8 = cd in ("mx8", "fp8", "mxfp8", "mx8-fwd")
        self.mx8_bwd = self.mx8 and cd != "mx8-fwd"
        if self.mx8 and (dim % 128 or mlp_dim % 128 or dim > 1536):
            raise ValueError(f"compute_dtype='mx8' needs dim and mlp_dim to be multiples of 128 and dim <= 1536 "
                             f"(dim={dim}, mlp_dim={mlp_dim})")
        self.compute_dtype = _lib.BF16 if self.mx8 else avf_dtype(compute_dtype)
        # residual_dtype="bf16" (throughput modes only): the forward residual stream x -> x + attn(..) -> x + mlp(..) is
        # STORED in bf16 between the kernels (LayerNorm statistics, GEMM accumulation and the add itself stay fp32): a third
        # fewer HBM bytes in the two LayerNorms and the two residual GEMM epilogues of a layer, for one bf16 rounding per
        # residual add (tolerance in tests/test_gpu_resid16.py).  "f32" keeps the fp32 stream.
        rd = str(residual_dtype).lower()
        if rd not in ("f32", "fp32", "float32", "bf16", "bfloat16"):
            raise ValueError(f"residual_dtype must be 'f32' or 'bf16', got {residual_dtype!r}")
        self.resid_bf16 = rd in ("bf16", "bfloat16")
        if self.resid_bf16 and (self.compute_dtype != _lib.BF16 or dim % 8 or dim > 1536):
            raise ValueError("residual_dtype='bf16' needs compute_dtype 'bf16' / 'mx8', dim % 8 == 0 and dim <= 1536")
        self.project_out = not (heads == 1 and dim_head == dim)
        self.layers = nn.ModuleList([_make_layer(dim, heads, dim_head, mlp_dim, dropout) for _ in range(depth)])
        if not self.project_out:
            self.register_buffer("_identity_w", torch.eye(dim), persistent=False)
            self.register_buffer("_identity_b", torch.zeros(dim), persistent=False)
        self._ws = None
        self._lowp_bufs = None
        self._lowp_ptrs = None
        self.cache_weights = False
        self._lowp_ready = False  # set by optim.FusedAdam: the bf16 copies already reflect the current masters
        # weights written behind the optimizer's back (checkpoint restore into a live model) invalidate the copies
        self.register_load_state_dict_post_hook(lambda module, incompatible_keys: module.refresh_weights())
        self._grad_hook: Optional[Callable] = None
        self._seed_dev = None
        self._last_seed_t = None
        # per-module salt of the dropout seed: construction order (deterministic from run to run, unlike id(self))
        Transformer._instances += 1
        self._seed_salt = Transformer._instances

    # per-process caches (ctypes structs, device scratch, bf16 weight images): never copied or pickled with the module
    _CACHES = {"_ws": None, "_lowp_bufs": None, "_lowp_ptrs": None, "_lowp_versions": None, "_lowp_ready": False,
               "_seed_dev": None, "_last_seed_t": None}
    _LAZY_CACHES = ("_pstruct_cache", "_mx_ptr_array", "_flat_cache", "_cfg_cache")  # created on first use

    def __getstate__(self):
        d = self.__dict__.copy()
        for k, v in self._CACHES.items():
            if k in d:
                d[k] = v
        for k in self._LAZY_CACHES:
            d.pop(k, None)
        return d

    # ---- parameter plumbing --------------------------------------------------------------------
    def layer_parameters(self, l: int) -> List[torch.Tensor]:
        """The 11 tensors of layer ``l`` in state_dict order."""
        attn_w, ff_w = self.layers[l]
        a, f = attn_w.fn, ff_w.fn
        if self.project_out:
            w_out, b_out = a.fn.to_out[0].weight, a.fn.to_out[0].bias
        else:  # nn.Identity to_out (heads.py:207): the library's projection GEMM runs on the identity matrix and a zero bias -
            # exact in every mode - held as non-persistent buffers (state_dict keeps the reference's keys), frozen, shared by
            # the layers; the library drops the dropout site nn.Identity does not have (cfg.project_out = 0)
            w_out, b_out = self._identity_w, self._identity_b
        return [a.norm.weight, a.norm.bias, a.fn.to_qkv.weight, w_out, b_out,
                f.norm.weight, f.norm.bias, f.fn.net[0].weight, f.fn.net[0].bias, f.fn.net[3].weight, f.fn.net[3].bias]

    def flat_parameters(self) -> List[torch.Tensor]:
        """all layers' tensors in order; the list is cached (walking the module tree costs ~0.2 ms per call) and rebuilt
        when a holder's Parameter object has been replaced"""
        cache = self.__dict__.get("_flat_cache")
        if cache is not None:
            first = self.layers[0][0].fn.norm.weight
            last = self.layers[self.depth - 1][1].fn.fn.net[3].bias
            if cache[0] is first and cache[-1] is last and (self.project_out or cache[3] is self._identity_w):
                return cache
        out = []
        for l in range(self.depth):
            out += self.layer_parameters(l)
        self.__dict__["_flat_cache"] = out
        return out

    def set_grad_hook(self, hook: Optional[Callable]):
        """``hook(layer_index, flat_fp32_grad_of_that_layer)`` is called right after the layer's backward has
        been enqueued (reverse layer order) - the data-parallel wrapper launches its all-reduce there.  It may
        return a callable that makes the current stream wait for that reduction."""
        self._grad_hook = hook

    def _cfg(self, B: int, N: int, layer: int = 0, seed_t: Optional[torch.Tensor] = None,
             keep: Optional[torch.Tensor] = None) -> _lib.LayerCfg:
        p = self.dropout if self.training else 0.0  # nn.Dropout semantics: identity in eval()
        return _lib.LayerCfg(B, N, self.dim, self.heads, self.dim_head, self.mlp_dim, self.compute_dtype,
                             int(self.project_out), 1e-5, float(p), 0, 0, layer,
                             seed_t.data_ptr() if (seed_t is not None and p != 0.0) else None,
                             int(self._grad_stream_bf16(p)), int(self.mx8), int(self.resid_bf16), int(self.mx8_bwd),
                             int(self.mx8_bwd and layer < self.depth - 1), None if keep is None else keep.data_ptr())

    def _grad_stream_bf16(self, p: float) -> bool:
        """backward keeps the residual gradient between the LayerNorm backward kernels in bf16 (the GEMMs read that image
        anyway): no fp32 dx write / read per sublayer.  Throughput mode without live dropout only; AVF_GRAD_STREAM=f32
        restores the fp32 stream."""
        # round 5: also WITH live dropout when every stream is bf16 (residual_dtype="bf16", not the fp8 mode, not the
        # single-launch short-sequence layers): each hand-off then carries two bf16 images, the stream and its masked copy
        drop_ok = p == 0.0 or (self.resid_bf16 and not self.mx8 and self.dim % 8 == 0)
        return (self.compute_dtype == _lib.BF16 and drop_ok and self.dim <= 1536 and self.dim % 4 == 0
                and os.environ.get("AVF_GRAD_STREAM", "bf16") != "f32")

    def _advance_seed(self, dev) -> Optional[torch.Tensor]:
        """Dropout seed of this forward, as a DEVICE tensor: the module's counter (initialised from torch.initial_seed()
        and the module's construction index, so runs are reproducible under torch.manual_seed) is advanced by an in-place add and snapshotted; the kernels read
        the snapshot at run time.  Both ops are capturable, so a hipGraph of a training step draws fresh masks on every
        replay.  (The masks come from a counter-based hash in the kernels, not from torch's generator.)"""
        if not (self.training and self.dropout > 0.0):
            return None
        if self._seed_dev is None or self._seed_dev.device != dev:
            host = (torch.initial_seed() * 0x9E3779B97F4A7C15 + self._seed_salt * 0xD1B54A32D192ED03
                    + self.__dict__.get("_seed_rank", 0) * 0xA0761D6478BD642F) & 0x7FFFFFFFFFFFFFFF
            self._seed_dev = torch.tensor([host], dtype=torch.int64, device=dev)
        if dev.type != "cuda":  # (the counter's bookkeeping alone, tests/test_cabi_cpu.py: no kernel reads it there)
            self._seed_dev.add_(1)
            self._last_seed_t = self._seed_dev.clone()
            return self._last_seed_t
        # counter += 1 and the snapshot in ONE launch (an in-place add + a clone were two per stack and step)
        self._last_seed_t = torch.empty_like(self._seed_dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.load().avf_seed_advance(self._seed_dev.data_ptr(), self._last_seed_t.data_ptr(),
                                                    C.c_void_p(torch.cuda.current_stream().cuda_stream)), "seed_advance")
        return self._last_seed_t

    @property
    def last_seed(self) -> int:
        """the 64-bit seed the latest training forward used (host sync; for tests / mask replay)"""
        return 0 if self._last_seed_t is None else int(self._last_seed_t.item()) & 0xFFFFFFFFFFFFFFFF

    def _param_struct(self, params, l) -> _lib.LayerPtrs:
        """LayerPtrs of layer l; the ctypes struct is reused while the eleven data pointers are unchanged"""
        ptrs = tuple(p.data_ptr() for p in params[l * PARAMS_PER_LAYER:(l + 1) * PARAMS_PER_LAYER])
        cache = self.__dict__.setdefault("_pstruct_cache", {})
        hit = cache.get(l)
        if hit is not None and hit[0] == ptrs:
            return hit[1]
        st = _lib.LayerPtrs(*ptrs)
        cache[l] = (ptrs, st)
        return st

    def _workspace(self, lib, cfg, dev):
        need = lib.avf_layer_workspace_bytes(C.byref(cfg))
        if self._ws is None or self._ws.numel() - _CANARY < need or self._ws.device != dev:
            self._ws = _alloc_bytes(need, dev)
        return self._ws

    def _lowp(self, lib, cfg, params, dev, stream):
        """bf16 weight copies (+ transposes) of the fp32 masters.  Refreshed on EVERY forward (an optimizer
        step changes the masters; in-place fused optimizers are not reliably visible through tensor version
        counters).  Inference loops may set ``self.cache_weights = True`` after the weights are final; call
        ``refresh_weights()`` if they change afterwards."""
        if cfg.dtype != _lib.BF16:
            return [None] * self.depth
        need = lib.avf_layer_lowp_bytes(C.byref(cfg))
        fresh = False
        if (self._lowp_bufs is None or self._lowp_bufs[0].numel() - _CANARY < need or self._lowp_bufs[0].device != dev):
            self._lowp_bufs = [_alloc_bytes(need, dev) for _ in range(self.depth)]
            fresh = True
        ptrs = [p.data_ptr() for p in params]
        vers = [p._version for p in params]  # detached views share the masters' version counters
        same = not fresh and self._lowp_ptrs == ptrs and self.__dict__.get("_lowp_versions") == vers
        # (1) the optimizer (optim.FusedAdam) rewrote the images in its own pass - valid for ONE forward, and only while
        #     nothing has modified a master in place since (EMA swap, clipping, a second optimizer: version counters);
        # (2) cache_weights (inference), or a stack without a single trainable tensor (the reference's frozen pretrained
        #     branches, avformer.py:76-85): reuse the images while pointers and version counters are unchanged.
        #     Writes through ``p.data`` bypass the counters: call refresh_weights() after those.
        ready = self._lowp_ready and same
        self._lowp_ready = False  # one forward per optimizer step; anything else re-prepares (weights may have changed)
        frozen = not any(q.requires_grad for q in self.flat_parameters())
        changed = ready  # the optimizer rewrote the bf16 images
        if not ready and not ((self.cache_weights or frozen) and same):
            for l in range(self.depth):
                pp = self._param_struct(params, l)
                _lib.check(lib.avf_layer_prepare_weights(C.byref(cfg), C.byref(pp), _ptr(self._lowp_bufs[l]), stream),
                           f"prepare_weights[{l}]")
            self._lowp_ptrs = ptrs
            self._lowp_versions = vers
            changed = True
        if self.mx8 and changed:  # the e4m3 images follow the bf16 ones: one launch for the stack
            if fresh or self.__dict__.get("_mx_ptr_array") is None:
                self._mx_ptr_array = (C.c_void_p * self.depth)(*[b.data_ptr() for b in self._lowp_bufs])
            _lib.check(lib.avf_stack_quant_weights_mx8(C.byref(cfg), self.depth, self._mx_ptr_array, stream),
                       "stack_quant_weights_mx8")
        return self._lowp_bufs

    def set_seed_rank(self, rank: int):
        """mix the data-parallel rank into the dropout seed (dp.DataParallel calls this): ranks that call
        torch.manual_seed with the same value still draw different masks"""
        self._seed_rank = int(rank)
        self._seed_dev = None

    def refresh_weights(self):
        self._lowp_ptrs = None
        self._lowp_versions = None
        self._lowp_ready = False

    # ---- forward -------------------------------------------------------------------------------
    def forward(self, x, mask=None, pool=None, fuse=None):
        """``pool='mean'`` (an addition to the reference signature) returns the token mean [B, dim] of the stack's output
        instead of [B, N, dim]: the pooling and its backward run inside the library (heads that pool, e.g.
        SyntheticAVFormer).  ``fuse=(audio_tokens, pos_embedding)``: ``x`` holds the video tokens and the stack runs on
        ``cat([x, audio], 1) + pos`` built by one library pass (dim % 4 == 0)."""
        if pool not in (None, 'mean'):
            raise ValueError(f"pool must be None or 'mean', got {pool!r}")
        if pool == 'mean' and self.dim % 4 != 0:
            raise ValueError("pool='mean' needs dim % 4 == 0 (use .mean(dim=1) on the unpooled output otherwise)")
        if not x.is_cuda:
            raise RuntimeError("Transformer (HIP) needs its input on the MI355X; there is no CPU fallback - "
                               "use oracle/ only as a test checker")
        if x.dim() != 3 or x.shape[-1] != self.dim:
            raise ValueError(f"expected [B, N, {self.dim}], got {tuple(x.shape)}")
        keep = None
        if mask is not None:
            # heads.py:225-232 (no reference caller passes a mask, but it is part of the signature): [B, ...] bool over the
            # tokens AFTER the first one; the reference pads a leading True.  The library takes the padded mask as bytes.
            if fuse is not None:
                raise ValueError("mask and fuse= cannot be combined")
            keep = torch.nn.functional.pad(mask.flatten(1).to(torch.bool), (1, 0), value=True)
            if keep.shape[0] != x.shape[0] or keep.shape[-1] != x.shape[1]:
                raise AssertionError("mask has incorrect dimensions")  # the reference's assert (heads.py:229)
            keep = keep.to(device=x.device, dtype=torch.uint8).contiguous()
        audio = pos = None
        if fuse is not None:
            audio, pos = fuse
            if self.dim % 4 != 0 or audio.dim() != 3 or audio.shape[0] != x.shape[0] or audio.shape[-1] != self.dim:
                raise ValueError("fuse=(audio, pos) needs dim % 4 == 0 and audio tokens [B, T_a, dim]")
            if x.shape[0] == 0 or x.shape[1] + audio.shape[1] == 0:
                x, audio, pos = torch.cat([x, audio], 1) + pos[..., :x.shape[1] + audio.shape[1], :], None, None
        if x.shape[0] == 0 or x.shape[1] == 0:
            # empty batch / empty sequence: nothing to launch.  As in the reference (every op is per token), the result is
            # the empty tensor of the right shape; parameters receive zero gradients through the zero-weight sum
            out = x.to(torch.float32) if pool is None else x.to(torch.float32).mean(dim=1)
            return out + sum(p.sum() for p in self.parameters()) * 0.0
        params = self.flat_parameters()
        for p in params:
            if p.dtype != torch.float32 or not p.is_cuda:
                raise RuntimeError("Transformer (HIP): parameters must be fp32 tensors on the GPU (model.to('cuda'))")
        with torch.cuda.device(x.device):  # launches go to the input's device and its current stream
            return _StackFn.apply(x.to(torch.float32), audio, pos, self, pool == 'mean', keep, *params)
