"""MX-FP8 NT GEMM against the bf16 NT GEMM on the forward nn.Linear shapes of a layer (C2: M = 32*324, C3: M = 32*512).

    python tools/bench_mx8.py [--rows 10368]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import avformer_amd as A  # noqa: E402


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10368)
    args = ap.parse_args()
    M = args.rows
    ops = A.ops
    shapes = [("qkv", 1536, 512, ops.EPI_NONE), ("mlp1+gelu", 1024, 512, ops.EPI_BIAS_GELU), ("mlp2+res", 512, 1024, ops.EPI_BIAS_RES),
              ("out+res", 512, 512, ops.EPI_BIAS_RES)]
    print(f"M = {M}")
    for name, N, K, epi in shapes:
        a = torch.randn(M, K, device="cuda")
        b = torch.randn(N, K, device="cuda") / K ** 0.5
        bias = torch.randn(N, device="cuda") if epi != ops.EPI_NONE else None
        res = torch.randn(M, N, device="cuda") if epi == ops.EPI_BIAS_RES else None
        od = torch.float32 if epi == ops.EPI_BIAS_RES else torch.bfloat16
        a16, b16 = a.bfloat16(), b.bfloat16()
        aq, as_ = ops.quant_mx8(a16)
        bq, bs = ops.quant_mx8(b)
        t16 = timeit(lambda: ops.gemm(a16, b16, out_dtype=od, epilogue=epi, bias=bias, residual=res))
        t8 = timeit(lambda: ops.gemm_mx8(aq, as_, bq, bs, out_dtype=od, epilogue=epi, bias=bias, residual=res))
        timg = timeit(lambda: ops.gemm_mx8(aq, as_, bq, bs, out_dtype=od, epilogue=epi, bias=bias, want_image=True)) \
            if epi == ops.EPI_BIAS_GELU else None
        tq = timeit(lambda: ops.quant_mx8(a16))
        fl = 2.0 * M * N * K
        print(f"{name:10s} N={N:5d} K={K:5d}  bf16 {t16:7.1f} us ({fl / t16 / 1e6:6.0f} TF)   mx8 {t8:7.1f} us ({fl / t8 / 1e6:6.0f} TF)"
              f"   x{t16 / t8:4.2f}   standalone quant of A {tq:6.1f} us"
              + (f"   mx8 + image of C {timg:6.1f} us" if timg is not None else ""))


if __name__ == "__main__":
    main()
