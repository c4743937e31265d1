#!/usr/bin/env python3
"""Per-operator micro-benchmarks on the MI355X (tuning aid; not part of the product path).
usage: python tools/bench_ops.py [gemm_nt|gemm_tn|attn|ln|all] [--rows R]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import avformer_amd as A  # noqa: E402

ops = A.ops


def timeit(fn, iters=30, warm=5):
    """seconds per call, from the library's HIP-event brackets around the kernel launches (host overhead excluded)"""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    A._lib.timing_enable(True)
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    A._lib.timing_enable(False)
    tm = A._lib.timing_read()
    ms = sum(v["ms"] for v in tm.values())
    return ms / iters * 1e-3


def gemm_nt(R):
    D, I, M = 512, 512, 1024
    shapes = [("qkv", R, 3 * I, D, ops.EPI_NONE, torch.bfloat16), ("out+res", R, D, I, ops.EPI_BIAS_RES, torch.float32),
              ("mlp1+gelu", R, M, D, ops.EPI_BIAS_GELU, torch.bfloat16), ("mlp2+res", R, D, M, ops.EPI_BIAS_RES, torch.float32),
              ("dX_w2+dgelu", R, M, D, ops.EPI_DGELU, torch.bfloat16), ("dX_w1", R, D, M, ops.EPI_NONE, torch.bfloat16),
              ("dX_qkv", R, D, 3 * I, ops.EPI_NONE, torch.bfloat16)]
    for name, m, n, k, epi, cdt in shapes:
        a = torch.randn(m, k, device="cuda").bfloat16()
        b = (torch.randn(n, k, device="cuda") / k ** 0.5).bfloat16()
        bias = torch.randn(n, device="cuda") if epi in (ops.EPI_BIAS_RES, ops.EPI_BIAS_GELU) else None
        res = torch.randn(m, n, device="cuda") if epi == ops.EPI_BIAS_RES else None
        aux = torch.randn(m, n, device="cuda").bfloat16() if epi in (ops.EPI_DGELU, ops.EPI_BIAS_GELU) else None
        fn = lambda: ops.gemm(a, b, out_dtype=cdt, epilogue=epi, bias=bias, residual=res, aux=aux)
        t = timeit(fn)
        print(f"gemm_nt {name:12s} M={m} N={n} K={k}: {t * 1e6:8.1f} us  {2.0 * m * n * k / t / 1e12:7.1f} TF/s")


def gemm_tn(R):
    D, I, M = 512, 512, 1024
    for name, m, n in [("dWqkv", 3 * I, D), ("dWo", D, I), ("dW1", M, D), ("dW2", D, M)]:
        a = torch.randn(R, m, device="cuda").bfloat16()
        b = torch.randn(R, n, device="cuda").bfloat16()
        fn = lambda: ops.gemm(a, b, trans_a=True, trans_b=False, out_dtype=torch.float32)
        t = timeit(fn)
        print(f"gemm_tn {name:12s} M={m} N={n} K={R}: {t * 1e6:8.1f} us  {2.0 * m * n * R / t / 1e12:7.1f} TF/s")


def gemm_tn_group(R):
    """the layer's four weight gradients as the grouped launch the backward uses"""
    D, I, M = 512, 512, 1024
    shapes = [(3 * I, D), (M, D), (D, M), (D, I)]
    pairs = [(torch.randn(R, m, device="cuda").bfloat16(), torch.randn(R, n, device="cuda").bfloat16()) for m, n in shapes]
    t = timeit(lambda: ops.gemm_tn_group(pairs))
    fl = sum(2.0 * m * n * R for m, n in shapes)
    print(f"gemm_tn_group (4 dW of a d=512 layer) K={R}: {t * 1e6:8.1f} us (GEMM + fold)  {fl / t / 1e12:7.1f} TF/s")


def attn(B, N):
    H, dh = 8, 64
    qkv = torch.randn(B * N, 3 * H * dh, device="cuda").bfloat16()
    d_o = torch.randn(B * N, H * dh, device="cuda").bfloat16()
    o, lse = ops.attn_fwd(qkv, B, N, H, dh)
    t = timeit(lambda: ops.attn_fwd(qkv, B, N, H, dh))
    fl = 4.0 * B * H * N * N * dh
    print(f"attn_fwd B={B} N={N}: {t * 1e6:8.1f} us  {fl / t / 1e12:7.1f} TF/s")
    if N <= 576:
        t = timeit(lambda: ops.attn_fwd_mx8(qkv, B, N, H, dh))
        print(f"attn_fwd+mx8 image B={B} N={N}: {t * 1e6:8.1f} us")
    t = timeit(lambda: ops.attn_bwd(qkv, o, d_o, lse, B, N, H, dh))
    print(f"attn_bwd B={B} N={N}: {t * 1e6:8.1f} us  {2.5 * fl / t / 1e12:7.1f} TF/s (nominal 2.5x fwd flops)")


def ln(R):
    D = 512
    x = torch.randn(R, D, device="cuda")
    w = torch.randn(D, device="cuda")
    b = torch.randn(D, device="cuda")
    y, mean, rstd = ops.layernorm_fwd(x, w, b, 1e-5, torch.bfloat16)
    t = timeit(lambda: ops.layernorm_fwd(x, w, b, 1e-5, torch.bfloat16))
    print(f"ln_fwd R={R}: {t * 1e6:8.1f} us  {R * D * 6 / t / 1e9:7.1f} GB/s")
    dy = torch.randn(R, D, device="cuda").bfloat16()
    t = timeit(lambda: ops.layernorm_bwd(dy, x, w, mean, rstd, dres=x, want_lo=True, want_colsum=True))
    print(f"ln_bwd R={R}: {t * 1e6:8.1f} us  {R * D * 16 / t / 1e9:7.1f} GB/s")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="?", default="all")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--tokens", type=int, default=324)
    a = ap.parse_args()
    R = a.batch * a.tokens
    if a.what in ("gemm_nt", "all"):
        gemm_nt(R)
    if a.what in ("gemm_tn", "all"):
        gemm_tn(R)
    if a.what in ("gemm_tn_group", "all"):
        gemm_tn_group(R)
    if a.what in ("attn", "all"):
        attn(a.batch, a.tokens)
    if a.what in ("ln", "all"):
        ln(R)
