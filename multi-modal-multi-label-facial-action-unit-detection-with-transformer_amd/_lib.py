"""ctypes binding of libavformer_hip.so (the C ABI declared in include/avformer_hip.h).

The product path has NO fallback: if the shared library is missing or cannot be loaded, every
entry point raises.  (The CPU oracle under /oracle is test infrastructure and is never imported
from here.)
"""
from __future__ import annotations

import ctypes as C
import os
import threading

from . import _build

F32, BF16 = 0, 1
EPI_NONE, EPI_BIAS_RES, EPI_BIAS_GELU, EPI_DGELU = 0, 1, 2, 3

_vp = C.c_void_p
_i64 = C.c_int64
_int = C.c_int
_f = C.c_float
_sz = C.c_size_t


class LayerCfg(C.Structure):
    _fields_ = [("batch", C.c_int32), ("tokens", C.c_int32), ("dim", C.c_int32), ("heads", C.c_int32),
                ("dim_head", C.c_int32), ("mlp_dim", C.c_int32), ("dtype", C.c_int32), ("project_out", C.c_int32),
                ("ln_eps", C.c_float), ("dropout_p", C.c_float), ("seed_lo", C.c_uint32), ("seed_hi", C.c_uint32),
                ("layer_index", C.c_int32), ("seed_dev", C.c_void_p), ("grad_stream_bf16", C.c_int32), ("mx8_fwd", C.c_int32),
                ("resid_bf16", C.c_int32), ("mx8_bwd", C.c_int32), ("dx_out_mx8", C.c_int32), ("key_mask", C.c_void_p)]


PARAM_FIELDS = ("ln1_w", "ln1_b", "w_qkv", "w_out", "b_out", "ln2_w", "ln2_b", "w1", "b1", "w2", "b2")


class LayerPtrs(C.Structure):
    """avf_layer_params / avf_layer_grads: 11 device pointers in state_dict order."""
    _fields_ = [(n, _vp) for n in PARAM_FIELDS]


# name -> (restype, argtypes); every symbol declared in include/avformer_hip.h
SIGNATURES = {
    "avf_version": (_int, []),
    "avf_sizeof_layer_cfg": (_sz, []),
    "avf_sizeof_layer_params": (_sz, []),
    "avf_last_error": (C.c_char_p, []),
    "avf_device_ok": (_int, []),
    "avf_layernorm_fwd": (_int, [_vp, _vp, _vp, _vp, _int, _vp, _vp, _i64, _int, _f, _vp]),
    "avf_layernorm_bwd_workspace_bytes": (_sz, [_i64, _int]),
    "avf_layernorm_bwd": (_int, [_vp, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _int, _vp]),
    "avf_colsum_workspace_bytes": (_sz, [_i64, _int]),
    "avf_colsum": (_int, [_vp, _int, _i64, _int, _i64, _vp, _vp, _vp]),
    "avf_cast_f32_to_bf16": (_int, [_vp, _vp, _i64, _vp]),
    "avf_prep_weight_bf16": (_int, [_vp, _vp, _vp, _int, _int, _vp]),
    "avf_gemm_workspace_bytes": (_sz, [_int, _int, _int, _i64, _i64, _i64]),
    "avf_gemm": (_int, [_int, _int, _int, _i64, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _int, _int, _vp, _vp,
                        _i64, _vp, _i64, _vp, _vp]),
    "avf_pack_weight_ws_ok": (_int, [_i64, _i64]),
    "avf_pack_weight_ws_bytes": (_sz, [_i64, _i64]),
    "avf_gemm_nt_ws_workspace_bytes": (_sz, [_i64, _i64]),
    "avf_gemm_nt_ws_dispatch": (_int, [_i64, _i64, _i64, _int, _int]),
    "avf_pack_weight_ws": (_int, [_vp, _i64, _i64, _i64, _vp, _vp]),
    "avf_gemm_nt_ws": (_int, [_i64, _i64, _i64, _vp, _i64, _vp, _vp, _i64, _int, _int, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp,
                              _vp]),
    "avf_gemm_tn_group_workspace_bytes": (_sz, [_int, _i64, _vp, _vp]),
    "avf_gemm_tn_group": (_int, [_int, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "avf_stack_quant_weights_mx8": (_int, [_vp, _int, _vp, _vp]),
    "avf_quant_mx8": (_int, [_int, _vp, _i64, _i64, _vp, _vp, _vp]),
    "avf_gemm_mx8_nt": (_int, [_i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _int, _int, _vp, _vp, _i64, _vp, _i64,
                               _vp, _vp, _vp]),
    "avf_layernorm_fwd_mx8": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _int, _f, _vp]),
    "avf_attn_fwd": (_int, [_int, _vp, _vp, _vp, _int, _int, _int, _int, _vp]),
    "avf_attn_bwd_workspace_bytes": (_sz, [_int, _int, _int, _int]),
    "avf_attn_bwd": (_int, [_int, _vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _vp]),
    "avf_attn_fwd_qs": (_int, [_vp, _vp, _vp, _int, _int, _int, _int, _vp]),
    "avf_attn_bwd_qs": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _vp]),
    "avf_fuse_tokens": (_int, [_vp, _vp, _vp, _vp, _int, _int, _int, _int, _vp]),
    "avf_token_mean_fwd": (_int, [_vp, _vp, _int, _int, _int, _vp]),
    "avf_token_mean_fwd_bf16": (_int, [_vp, _vp, _int, _int, _int, _vp]),
    "avf_fuse_tokens_bf16": (_int, [_vp, _vp, _vp, _vp, _int, _int, _int, _int, _vp]),
    "avf_token_mean_bwd": (_int, [_vp, _vp, _vp, _vp, _int, _int, _int, _vp]),
    "avf_bn1d_fwd": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _f, _f, _int, _vp]),
    "avf_bn1d_bwd": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _int, _vp]),
    "avf_token_dots_fwd": (_int, [_vp, _vp, _i64, _vp, _i64, _int, _int, _int, _int, _vp]),
    "avf_token_dots_bwd": (_int, [_vp, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _int, _int, _int, _vp]),
    "avf_assemble_tokens": (_int, [_vp, _vp, _vp, _vp, _int, _int, _int, _int, _vp]),
    "avf_cat_features": (_int, [_vp, _vp, _vp, _vp, _int, _int, _int, _int, _vp]),
    "avf_transpose_add": (_int, [_vp, _vp, _vp, _int, _int, _int, _vp]),
    "avf_zero_cols": (_int, [_vp, _i64, _int, _int, _int, _vp]),
    "avf_seed_advance": (_int, [_vp, _vp, _vp]),
    "avf_linear_pad_fwd": (_int, [_vp, _vp, _vp, _vp, _int, _int, _int, _int, _vp]),
    "avf_linear_pad_bwd": (_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _int, _int, _int, _vp]),
    "avf_au_loss": (_int, [_vp, _i64, _vp, _i64, _vp, _f, _int, _int, _vp, _vp, _vp]),
    "avf_au_loss_sum": (_int, [_vp, _i64, _vp, _i64, _vp, _f, _int, _int, _vp, _vp, _vp]),
    "avf_au_loss_wide": (_int, [_vp, _i64, _vp, _i64, _vp, _f, _int, _int, _int, _int, _vp, _vp, _vp]),
    "avf_layer_saved_bytes": (_sz, [C.POINTER(LayerCfg)]),
    "avf_layer_lowp_bytes": (_sz, [C.POINTER(LayerCfg)]),
    "avf_layernorm_bwd_mx8": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _int, _vp]),
    "avf_attn_fwd_mx8": (_int, [_vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _vp]),
    "avf_attn_bwd_emits_mx8": (_int, [_int, _int]),
    "avf_attn_bwd_mx8": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _vp]),
    "avf_attn_fwd_masked": (_int, [_int, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _vp]),
    "avf_attn_bwd_masked": (_int, [_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _vp]),
    "avf_layer_workspace_bytes": (_sz, [C.POINTER(LayerCfg)]),
    "avf_layer_grad_stream_bytes": (_sz, [C.POINTER(LayerCfg)]),
    "avf_layer_prepare_weights": (_int, [C.POINTER(LayerCfg), C.POINTER(LayerPtrs), _vp, _vp]),
    "avf_layer_fwd": (_int, [C.POINTER(LayerCfg), C.POINTER(LayerPtrs), _vp, _vp, _vp, _vp, _vp, _vp]),
    "avf_layer_bwd": (_int, [C.POINTER(LayerCfg), C.POINTER(LayerPtrs), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                             C.POINTER(LayerPtrs), _vp, _vp]),
    "avf_layer_adam_step": (_int, [C.POINTER(LayerCfg), C.POINTER(LayerPtrs), C.POINTER(LayerPtrs), C.POINTER(LayerPtrs),
                                   C.POINTER(LayerPtrs), _vp, _f, _f, _f, _f, _f, _vp, _vp]),
    "avf_stack_adam_step": (_int, [C.POINTER(LayerCfg), _int, C.POINTER(LayerPtrs), C.POINTER(LayerPtrs), C.POINTER(LayerPtrs),
                                   C.POINTER(LayerPtrs), _vp, _f, _f, _f, _f, _f, _vp, _vp]),
    "avf_adam_step_tensors": (_int, [_int, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _f, _vp, _vp]),
    "avf_adam_batch_begin": (_int, []),
    "avf_adam_batch_end": (_int, []),
    "avf_adam_batch_abort": (_int, []),
    "avf_selftest_adam_table": (_int, [_vp]),
    "avf_dropout_factors": (_int, [C.c_uint32, C.c_uint32, _int, _int, _f, _i64, _int, _vp, _vp]),
    "avf_timing_enable": (_int, [_int]),
    "avf_timing_read": (_int, [_int, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double),
                               C.POINTER(C.c_double)]),
    "avf_hip_error_reset": (_int, [_vp]),
    "avf_crash_line_arm": (_int, [C.c_char_p, _int]),
    "avf_crash_line_disarm": (_int, []),
    "avf_set_f32_arith": (_int, [_int]),
    "avf_get_f32_arith": (_int, []),
    "avf_selftest_mfma_bf16": (_int, [_vp, _vp, _vp, _vp]),
    "avf_selftest_mfma_f32": (_int, [_vp, _vp, _vp, _vp]),
    "avf_selftest_tr16": (_int, [_vp, _vp, _vp]),
}

_lock = threading.Lock()
_lib = None


class HipLibraryError(RuntimeError):
    pass


def load(build_if_missing: bool = True):
    """Load (once) and return the ctypes handle.  If the shared library has not been built yet and hipcc is present,
    it is compiled in-tree first (``python __graft_entry__.py`` does the same explicitly).  Raises HipLibraryError
    when neither a built library nor a toolchain is available - there is no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = _build.lib_path()
        if not os.path.exists(path):
            err = None
            if build_if_missing:
                try:
                    _build.build()
                except Exception as e:  # toolchain missing or compile error: report both facts
                    err = e
            if not os.path.exists(path):
                raise HipLibraryError(
                    f"{path} not found and could not be built ({err}); run python __graft_entry__.py on a machine "
                    f"with hipcc - there is no CPU fallback")
        try:
            lib = C.CDLL(path)
        except OSError as e:  # pragma: no cover - environment dependent
            raise HipLibraryError(f"cannot load {path}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError as e:
                raise HipLibraryError(f"{path} does not export {name}") from e
            fn.restype = res
            fn.argtypes = args
        # a stale .so (or an edited struct) must not be called with structs of another layout
        if lib.avf_sizeof_layer_cfg() != C.sizeof(LayerCfg) or lib.avf_sizeof_layer_params() != C.sizeof(LayerPtrs):
            raise HipLibraryError(
                f"{path}: avf_layer_cfg / avf_layer_params are {lib.avf_sizeof_layer_cfg()} / "
                f"{lib.avf_sizeof_layer_params()} bytes in the library but {C.sizeof(LayerCfg)} / {C.sizeof(LayerPtrs)} "
                f"in this binding - rebuild with python __graft_entry__.py")
        _lib = lib
    return _lib


KERNEL_CLASSES = ("gemm_bf16_nt", "gemm_bf16_tn", "gemm_f32", "attn_fwd", "attn_bwd", "layernorm", "other", "gemm_mx8_nt")


def timing_enable(on: bool):
    check(load().avf_timing_enable(int(on)), "timing_enable")


def timing_read():
    """-> {class name: dict(ms, launches, flops, bytes)} for the launches recorded since timing_enable(True)."""
    lib = load()
    out = {}
    for i, name in enumerate(KERNEL_CLASSES):
        ms, n, fl, by = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
        check(lib.avf_timing_read(i, C.byref(ms), C.byref(n), C.byref(fl), C.byref(by)), "timing_read")
        out[name] = dict(ms=ms.value, launches=n.value, flops=fl.value, bytes=by.value)
    return out


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().avf_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"libavformer_hip: {what} failed (rc={rc}): {msg}")


F32_ARITH = {"f32": 0, "bf16x3": 1}


def set_f32_arithmetic(mode: str) -> str:
    """Arithmetic of compute_dtype="f32" GEMMs and attention, process-wide: "bf16x3" (default: fp32 operands split in
    three bf16 products on the bf16 matrix pipe: <= 3 * 2^-16 = 4.6e-5 relative error per product in the worst case, 4e-6 typical) or "f32" (the f32-input MFMA).
    Returns the previous mode's name."""
    prev = load().avf_set_f32_arith(F32_ARITH[mode])
    return "bf16x3" if prev else "f32"


def get_f32_arithmetic() -> str:
    return "bf16x3" if load().avf_get_f32_arith() else "f32"
