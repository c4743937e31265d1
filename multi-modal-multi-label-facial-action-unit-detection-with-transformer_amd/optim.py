"""Adam for models built on the HIP ``Transformer``: the parameters of every stack are stepped by the library
(``avf_layer_adam_step``: one launch per layer that also rewrites the layer's bf16 weight copies, so the next forward
skips its weight-preparation pass); the parameters around the stacks (embeddings, heads) take the same kernel through
``avf_adam_step_tensors``.

Semantics are ``torch.optim.Adam(params, lr, betas, eps, weight_decay)`` - the optimizer of the reference's training
loop (train.py:318-322): L2 weight decay added to the gradient, bias correction, ``amsgrad=False``.  State keys
(``step``, ``exp_avg``, ``exp_avg_sq``) are the same, so ``state_dict()`` looks like torch's.
"""
from __future__ import annotations

import ctypes as C
from typing import Iterable, List

import torch

from . import _lib
from .transformer import PARAMS_PER_LAYER, Transformer


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, model: torch.nn.Module, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if not isinstance(model, torch.nn.Module):
            raise TypeError("FusedAdam takes the model (it needs to find the Transformer stacks), not a parameter list")
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self._stacks: List[Transformer] = [m for m in model.modules() if isinstance(m, Transformer)]
        stack_params, owned = [], set()
        for st in self._stacks:
            for p in st.flat_parameters():
                if id(p) not in owned and p.requires_grad:
                    stack_params.append(p)
                    owned.add(id(p))
        rest = [p for p in model.parameters() if id(p) not in owned and p.requires_grad]
        groups = [dict(params=stack_params, hip=True)]
        if rest:
            groups.append(dict(params=rest, hip=False))
        super().__init__(groups, defaults)
        self._owned = owned
        self._step_dev = None
        self._flat = {}  # (stack index, layer) -> (exp_avg flat, exp_avg_sq flat, [views], [views])

    # ---- state --------------------------------------------------------------------------------------
    def _layer_state(self, si: int, l: int, params):
        key = (si, l)
        if key not in self._flat:
            n = sum(p.numel() for p in params)
            dev = params[0].device
            m = torch.zeros(n, dtype=torch.float32, device=dev)
            v = torch.zeros(n, dtype=torch.float32, device=dev)
            mv, vv, off = [], [], 0
            for p in params:
                k = p.numel()
                mv.append(m[off:off + k].view_as(p))
                vv.append(v[off:off + k].view_as(p))
                off += k
                if id(p) not in self._owned:
                    continue  # frozen tensor of a partly trainable layer: its slice is private scratch (the kernel skips
                              # tensors without a gradient); it is in no param_group, so it must not appear in self.state
                st = self.state[p]
                if "exp_avg" in st:  # state loaded before the first step: adopt it
                    mv[-1].copy_(st["exp_avg"])
                    vv[-1].copy_(st["exp_avg_sq"])
                st["exp_avg"], st["exp_avg_sq"] = mv[-1], vv[-1]
            self._flat[key] = (m, v, mv, vv)
        return self._flat[key]

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._flat.clear()  # re-adopt the loaded tensors into flat buffers at the next step
        steps = [float(s["step"]) for s in self.state.values() if "step" in s]
        self._step_dev = None
        self._loaded_step = max(steps) if steps else 0.0

    # ---- step ---------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        hip_group = self.param_groups[0]
        dev = next((p.device for grp in self.param_groups for p in grp["params"]), None)
        if dev is None:
            return loss
        if dev.type != "cuda":
            raise RuntimeError("FusedAdam: the model must be on the GPU (no CPU fallback)")
        if self._step_dev is None or self._step_dev.device != dev:
            self._step_dev = torch.full((1,), getattr(self, "_loaded_step", 0.0), dtype=torch.float32, device=dev)
        # every stack and every loose tensor of this step in ONE descriptor table (avf_adam_batch_begin / _end): the per-stack
        # launches of the reference's real model (five small stacks + the head's tensors) were 80 us of a 650 us step.
        # The whole session runs with the model's device current (the table is launched by _end, which must see the device its
        # pointers live on), and a failure while the table is being collected ABORTS the session: a half-built table is never
        # launched, the step counter is not advanced, and the original exception propagates.
        with torch.cuda.device(dev):
            _lib.check(lib.avf_adam_batch_begin(), "adam_batch_begin")
            self._step_dev.add_(1.0)  # device-side counter: the kernels read it at run time (graph-capturable)
            try:
                keep = self._step_body(lib, hip_group, dev)  # (converted gradients: alive until the table has been launched)
            except BaseException:
                lib.avf_adam_batch_abort()
                self._step_dev.sub_(1.0)
                raise
            _lib.check(lib.avf_adam_batch_end(), "adam_batch_end")
            del keep
        return loss

    def _step_body(self, lib, hip_group, dev):
        keep = []  # (temporaries whose pointers the pending table holds until the batch is launched)
        if self._stacks:
            b1, b2 = hip_group["betas"]
            with torch.cuda.device(dev):
                stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
                for si, st in enumerate(self._stacks):
                    params = st.flat_parameters()
                    if not any(id(p) in self._owned for p in params):
                        continue  # fully frozen stack (pretrained branch): no state, no launch; its forward caches the images
                    cfg = st._cfg(1, 1)
                    bf16 = cfg.dtype == _lib.BF16
                    if bf16 and st._lowp_bufs is None:
                        st._lowp(lib, cfg, [p.detach() for p in params], dev, stream)  # allocate (and fill) the copies
                    lr, eps, wd = float(hip_group["lr"]), float(hip_group["eps"]), float(hip_group["weight_decay"])
                    L = st.depth
                    # ctypes arrays of per-layer structs; the ones of the stable pointers (parameters, moments, bf16
                    # images) are built once per stack, the gradient one every step
                    key = (si, "arrays")
                    pptr = tuple(p.data_ptr() for p in params)
                    hit = self._flat.get(key)
                    if hit is None or hit[0] != pptr:
                        P_, M_, V_ = (_lib.LayerPtrs * L)(), (_lib.LayerPtrs * L)(), (_lib.LayerPtrs * L)()
                        for l in range(L):
                            lp = params[l * PARAMS_PER_LAYER:(l + 1) * PARAMS_PER_LAYER]
                            m, v, mv, vv = self._layer_state(si, l, lp)
                            P_[l] = _lib.LayerPtrs(*[t.data_ptr() for t in lp])
                            M_[l] = _lib.LayerPtrs(*[t.data_ptr() for t in mv])
                            V_[l] = _lib.LayerPtrs(*[t.data_ptr() for t in vv])
                            for p in lp:
                                if id(p) in self._owned:
                                    self.state[p]["step"] = self._step_dev  # shared device counter (as capturable Adam)
                        lows = (C.c_void_p * L)(*[(b.data_ptr() if bf16 else None) for b in (st._lowp_bufs or [None] * L)]) \
                            if bf16 else None
                        hit = (pptr, P_, M_, V_, lows)
                        self._flat[key] = hit
                    G_ = (_lib.LayerPtrs * L)()
                    for l in range(L):
                        gptr = []
                        for p in params[l * PARAMS_PER_LAYER:(l + 1) * PARAMS_PER_LAYER]:
                            g = p.grad if (p.requires_grad and id(p) in self._owned) else None
                            if g is not None and (g.dtype != torch.float32 or not g.is_contiguous()):
                                g = g.to(torch.float32).contiguous()
                                keep.append(g)
                            gptr.append(None if g is None else g.data_ptr())
                        G_[l] = _lib.LayerPtrs(*gptr)
                    _lib.check(lib.avf_stack_adam_step(C.byref(cfg), L, hit[1], G_, hit[2], hit[3], hit[4], lr, float(b1),
                                                       float(b2), eps, wd, _ptr(self._step_dev), stream), "stack_adam_step")
                    if bf16:
                        st._lowp_ptrs = [p.data_ptr() for p in params]
                        st._lowp_versions = [p._version for p in params]  # an in-place edit after this step voids the skip
                        st._lowp_ready = True  # the next forward may skip its weight-preparation pass
        if len(self.param_groups) > 1:
            grp = self.param_groups[1]
            todo = [p for p in grp["params"] if p.grad is not None]
            if todo:
                dev = todo[0].device
                if dev.type != "cuda":
                    raise RuntimeError("FusedAdam: the model must be on the GPU (no CPU fallback)")
                ps, gs, ms, vs = [], [], [], []
                for p in todo:
                    st = self.state[p]
                    if "exp_avg" not in st:
                        st["exp_avg"] = torch.zeros_like(p, dtype=torch.float32, memory_format=torch.contiguous_format)
                        st["exp_avg_sq"] = torch.zeros_like(p, dtype=torch.float32, memory_format=torch.contiguous_format)
                    st["step"] = self._step_dev
                    g = p.grad
                    if g.dtype != torch.float32 or not g.is_contiguous():
                        g = g.to(torch.float32).contiguous()
                    if not p.is_contiguous() or p.dtype != torch.float32:
                        raise RuntimeError("FusedAdam: parameters must be contiguous fp32 tensors")
                    ps.append(p); gs.append(g); ms.append(st["exp_avg"]); vs.append(st["exp_avg_sq"])
                n = len(ps)
                arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
                numel = (C.c_int64 * n)(*[t.numel() for t in ps])
                b1, b2 = grp["betas"]
                with torch.cuda.device(dev):
                    _lib.check(lib.avf_adam_step_tensors(n, arr(ps), arr(gs), arr(ms), arr(vs), numel, float(grp["lr"]),
                                                         float(b1), float(b2), float(grp["eps"]), float(grp["weight_decay"]),
                                                         _ptr(self._step_dev),
                                                         C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                               "adam_step_tensors")
                keep.extend(gs)
        return keep
