"""Helpers shared by the -m gpu tests (HIP path vs the CPU oracle)."""
import torch

import avformer_amd as A  # noqa: F401  (alias of the hyphenated package directory)
import oracle

DEV = "cuda"


import contextlib


@contextlib.contextmanager
def f32_arithmetic(mode):
    """run a block with the parity mode's arithmetic set to "bf16x3" (default: three bf16 products per fp32 product on the
    bf16 matrix pipe) or "f32" (the f32-input MFMA); avf_set_f32_arith is process-wide, so the previous mode is restored"""
    from avformer_amd import _lib
    prev = _lib.set_f32_arithmetic(mode)
    try:
        yield mode
    finally:
        _lib.set_f32_arithmetic(prev)


F32_ARITHS = ("bf16x3", "f32")


def rel_fro(a, b):
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def max_abs(a, b):
    return (a.detach().float().cpu() - b.detach().float().cpu()).abs().max().item()


def make_hip_transformer(sd, dim, depth, heads, dim_head, mlp_dim, compute_dtype):
    t = A.Transformer(dim, depth, heads, dim_head, mlp_dim, 0.0, compute_dtype=compute_dtype)
    missing = t.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return t.to(DEV)


def oracle_transformer_run(x, sd, depth, heads, loss_fn):
    """CPU oracle forward + autograd backward -> y, dx, {param: grad}."""
    x = x.detach().cpu().clone().requires_grad_(True)
    ps = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in sd.items()}
    y = oracle.transformer_forward(x, ps, depth, heads)
    loss_fn(y).backward()
    return y.detach(), x.grad, {k: v.grad for k, v in ps.items()}


def hip_transformer_run(t, x, loss_fn):
    x = x.detach().to(DEV).clone().requires_grad_(True)
    for p in t.parameters():
        p.grad = None
    y = t(x)
    loss_fn(y).backward()
    torch.cuda.synchronize()
    return y.detach(), x.grad, {k: p.grad for k, p in t.named_parameters()}


# ------------------------------------------------------------------------------------------------------------------
# Calibrated low-precision bounds.  A bf16 / mx8 assertion states a CAP (what the format allows in the worst case)
# and is additionally held to 3x the error MEASURED for that exact case on an MI355X, recorded in
# tests/golden/lowp_measured.json (written by a calibration run: AVF_RECORD_ERRORS=<path> pytest -m gpu, then
# tools/calibrate_bounds.py).  A kernel regression that triples an error therefore fails even where the cap is loose.
# The message of a failing assertion prints measured value, calibrated value and bound.
# ------------------------------------------------------------------------------------------------------------------
import atexit
import json
import os

_MEAS_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lowp_measured.json")
try:
    with open(_MEAS_PATH) as _f:
        CALIBRATED = json.load(_f)["measured"]
except Exception:
    CALIBRATED = {}
_RECORD = os.environ.get("AVF_RECORD_ERRORS")
_seen = {}
BOUND_FACTOR = 3.0
BOUND_FLOOR = 2e-6  # errors this small are fp32 rounding noise; 3x of (nearly) nothing is not a meaningful bound


def bound_for(tag, cap, floor=BOUND_FLOOR):
    m = CALIBRATED.get(tag)
    if m is None:
        return cap
    return min(cap, max(BOUND_FACTOR * m, floor))


def check(tag, err, cap, floor=BOUND_FLOOR):
    """assert err <= min(cap, max(3 x the calibrated measurement of `tag`, floor)).  `floor`: for quantities that are small
    differences of large sums (a loss value), where 3 x a tiny measured deviation would be noise, not a bound"""
    err = float(err)
    _seen[tag] = max(err, _seen.get(tag, 0.0))
    b = bound_for(tag, cap, floor)
    cal = CALIBRATED.get(tag)
    assert err <= b, (f"{tag}: measured {err:.3e} > bound {b:.3e} (cap {cap:.1e}, calibrated "
                      f"{'-' if cal is None else format(cal, '.3e')} x{BOUND_FACTOR:g})")
    return err


def check_rel(tag, a, b, cap):
    return check(tag, rel_fro(a, b), cap)


def check_abs(tag, a, b, cap, floor=BOUND_FLOOR):
    return check(tag, max_abs(a, b), cap, floor)


def _dump_seen():
    if _RECORD and _seen:
        old = {}
        try:
            with open(_RECORD) as f:
                old = json.load(f)
        except Exception:
            pass
        old.update(_seen)
        os.makedirs(os.path.dirname(os.path.abspath(_RECORD)), exist_ok=True)
        with open(_RECORD, "w") as f:
            json.dump(old, f, indent=0, sort_keys=True)


atexit.register(_dump_seen)
