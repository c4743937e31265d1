"""Weight-stationary persistent NT GEMM (csrc/gemm_ws.hip) against the tiled kernel: bit-equality and back-to-back time.

    python tools/diag/ws_ab.py --rows 16384 [--rounds 5]

Per shape (the layer's K = 512 GEMMs): the two kernels run the same inputs, outputs must be bit-identical (same epilogue
code, same k order of the accumulation); timing alternates tiled / persistent over several rounds in ONE process and
prints the median and the minimum of each (cdna_hip_programming.md rule 24).
"""
import argparse
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import avformer_amd as A  # noqa: E402


def timeit(fn, iters=40):
    for _ in range(3):
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
    ap.add_argument("--rows", type=int, default=16384)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--resid", default="bf16", choices=["bf16", "f32"])
    args = ap.parse_args()
    M = args.rows
    ops = A.ops
    bf = torch.bfloat16
    rdt = bf if args.resid == "bf16" else torch.float32
    shapes = [("qkv", 1536, ops.EPI_NONE, bf), ("out+res", 512, ops.EPI_BIAS_RES, rdt), ("mlp1+gelu", 1024, ops.EPI_BIAS_GELU, bf),
              ("dx_w2+dgelu", 1024, ops.EPI_DGELU, bf), ("dx_out", 512, ops.EPI_NONE, bf)]
    torch.manual_seed(0)
    print(f"M = {M}, K = 512, residual {args.resid}")
    for name, N, epi, od in shapes:
        K = 512
        a = torch.randn(M, K, device="cuda").to(bf)
        w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(bf)
        wp = ops.pack_ws(w)
        bias = torch.randn(N, device="cuda") if epi in (ops.EPI_BIAS_RES, ops.EPI_BIAS_GELU) else None
        res = torch.randn(M, N, device="cuda").to(od) if epi == ops.EPI_BIAS_RES else None
        aux_in = torch.randn(M, N, device="cuda").to(bf) if epi == ops.EPI_DGELU else None

        def tiled():
            return ops.gemm(a, w, out_dtype=od, epilogue=epi, bias=bias, residual=res, aux=aux_in)

        def ws():
            return ops.gemm_ws(a, wp, N, out_dtype=od, epilogue=epi, bias=bias, residual=res, aux=aux_in)

        r0, r1 = tiled(), ws()
        torch.cuda.synchronize()
        r0 = r0 if isinstance(r0, tuple) else (r0,)
        r1 = r1 if isinstance(r1, tuple) else (r1,)
        same = all(torch.equal(x, y) for x, y in zip(r0, r1))
        maxd = max(float((x.float() - y.float()).abs().max()) for x, y in zip(r0, r1))
        t0s, t1s = [], []
        for _ in range(args.rounds):
            t0s.append(timeit(tiled))
            t1s.append(timeit(ws))
        fl = 2.0 * M * N * K
        m0, m1 = statistics.median(t0s), statistics.median(t1s)
        print(f"{name:12s} N={N:5d}  bit-equal {same} (max |d| {maxd:.3g})  tiled {m0:6.1f} us (min {min(t0s):6.1f}, {fl / m0 / 1e6:5.0f} TF)"
              f"   ws {m1:6.1f} us (min {min(t1s):6.1f}, {fl / m1 / 1e6:5.0f} TF = {fl / m1 / 1e6 / 2500:.3f})   x{m0 / m1:.2f}", flush=True)


if __name__ == "__main__":
    main()
