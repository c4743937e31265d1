#!/usr/bin/env python3
"""Where do the hot-path kernels stand against the vendor's own kernels for the same operators on the same box?

A yard-stick, not a parity test and nothing the product uses: the roofline fractions of DESIGN.md section 4 say how far a
kernel is from the MFMA peak, this says how far the libraries shipped with ROCm get at the same (small) problem sizes.

  NT GEMMs   torch.matmul(a, w.t())            -> hipBLASLt, plain (no epilogue on either side)
  TN GEMMs   torch.matmul(a.t(), b)            -> hipBLASLt, bf16 output (ours writes fp32: the weight gradients)
  attention  F.scaled_dot_product_attention     -> the flash backend torch ships for ROCm, forward and forward + backward

Every timing is `iters` back-to-back launches between two events on the current stream (torch ops and ours alike).
Usage: python tools/vendor_yardstick.py [--configs c2,c3] > profiles/rNN_vendor_yardstick.txt
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import avformer_amd as A  # noqa: E402

ops = A.ops

CONFIGS = {"c2": (32, 324), "c3": (32, 512)}  # (clips, tokens) at d = 512, 8 heads of 64, mlp 1024


def t_events(fn, iters=40, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def line(tag, flops, tv, to):
    print(f"  {tag:44s} vendor {tv * 1e6:7.1f} us {flops / tv / 1e12:7.1f} TF/s | ours {to * 1e6:7.1f} us {flops / to / 1e12:7.1f} TF/s"
          f" | ours/vendor time {to / tv:5.2f}")


def gemms(R):
    print(f" NT GEMMs, {R} rows (hipBLASLt through torch.matmul; ours: the dispatch of avf_gemm, persistent kernel where it applies)")
    for name, n, k in (("to_qkv", 1536, 512), ("to_out / d_o", 512, 512), ("net.0 / dGELU", 1024, 512),
                       ("net.3 / dh2", 512, 1024), ("dX of to_qkv", 512, 1536)):
        a = torch.randn(R, k, device="cuda").bfloat16()
        w = (torch.randn(n, k, device="cuda") / k ** 0.5).bfloat16()
        out = torch.empty(R, n, device="cuda", dtype=torch.bfloat16)
        tv = t_events(lambda: torch.matmul(a, w.t(), out=out))
        if k == 512 and ops.gemm_ws_used(R, n, k):
            wp = ops.pack_ws(w)
            to = t_events(lambda: ops.gemm_ws(a, wp, n, out_dtype=torch.bfloat16))
        else:
            to = t_events(lambda: ops.gemm(a, w, out_dtype=torch.bfloat16))
        line(f"{name} [{R} x {k}] x [{n} x {k}]^T", 2.0 * R * n * k, tv, to)
    print(f" TN GEMMs (weight gradients), reduction over {R} rows; vendor: four launches, bf16 out; ours: ONE grouped launch + fold, fp32 out")
    shapes = ((512, 1536), (512, 512), (512, 1024), (1024, 512))
    As = [torch.randn(R, m, device="cuda").bfloat16() for m, _ in shapes]
    Bs = [torch.randn(R, n, device="cuda").bfloat16() for _, n in shapes]
    outs = [torch.empty(m, n, device="cuda", dtype=torch.bfloat16) for m, n in shapes]

    def vendor():
        for a, b, o in zip(As, Bs, outs):
            torch.matmul(a.t(), b, out=o)

    tv = t_events(vendor)
    to = t_events(lambda: ops.gemm_tn_group(list(zip(As, Bs))))
    line("dWqkv + dWo + dW1 + dW2", 2.0 * R * sum(m * n for m, n in shapes), tv, to)


def attention(B, N, H=8, dh=64):
    print(f" attention, {B} clips x {N} tokens, {H} heads of {dh} (vendor: F.scaled_dot_product_attention, flash backend)")
    from torch.nn.attention import SDPBackend, sdpa_kernel
    I = H * dh
    qkv = (torch.randn(B * N, 3 * I, device="cuda") * 0.5).bfloat16()
    q, k, v = (qkv[:, i * I:(i + 1) * I].reshape(B, N, H, dh).transpose(1, 2).contiguous().requires_grad_(True) for i in range(3))
    go = torch.randn(B, H, N, dh, device="cuda").bfloat16()
    with sdpa_kernel(SDPBackend.FLASH_ATTENTION):
        with torch.no_grad():
            tvf = t_events(lambda: F.scaled_dot_product_attention(q, k, v))

        def fb():
            out = F.scaled_dot_product_attention(q, k, v)
            out.backward(go)
            q.grad = k.grad = v.grad = None

        tvfb = t_events(fb)
    qs = qkv.clone()
    qs[:, :I] = (qs[:, :I].float() * (1.4426950408889634 / dh ** 0.5)).bfloat16()  # the layer path's pre-scaled q columns
    o, lse2 = ops.attn_fwd(qs, B, N, H, dh, q_prescaled=True)
    d_o = torch.randn_like(o)
    tof = t_events(lambda: ops.attn_fwd(qs, B, N, H, dh, q_prescaled=True))
    tob = t_events(lambda: ops.attn_bwd(qs, o, d_o, lse2, B, N, H, dh, q_prescaled=True))
    ff, fbw = 4.0 * B * H * N * N * dh, 10.0 * B * H * N * N * dh
    line("forward", ff, tvf, tof)
    line("backward (vendor: fwd+bwd minus fwd)", fbw, max(tvfb - tvf, 1e-9), tob)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="c2,c3")
    args = ap.parse_args()
    print(f"vendor yard-stick on {torch.cuda.get_device_name(0)}, torch {torch.__version__}; standalone launches back to back (the step's "
          f"launches run between other kernels: compare ratios, not absolute us, with profiles/*_shapes.csv)")
    for c in args.configs.split(","):
        B, N = CONFIGS[c]
        print(f"== {c.upper()}: {B} clips x {N} tokens = {B * N} rows")
        gemms(B * N)
        attention(B, N)


if __name__ == "__main__":
    main()
