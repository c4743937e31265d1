"""Random-shape check of the NT GEMM kernels (bf16 and MX-FP8 operands, every fused epilogue, both output types)
against an fp32 product of the same (rounded / dequantised) operands.

    python tests/fuzz/fuzz_gemm.py [--cases 300] [--seed 0]
"""
import argparse
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import avformer_amd as A  # noqa: E402
import oracle  # noqa: E402


def gelu(x):
    return oracle.gelu_tanh(x)


def dgelu(u):
    u = u.double().requires_grad_(True)
    oracle.gelu_tanh(u).sum().backward()
    return u.grad.float()


def run(cases=300, seed=0):
    """`cases` random NT GEMM cases from `seed`; returns worst error / tolerance, raises AssertionError on the first failing case
    (tests/test_gpu_fuzz.py runs a fixed budget of these under pytest -m gpu)"""
    args = argparse.Namespace(cases=cases, seed=seed)
    rnd = random.Random(args.seed)
    ops = A.ops
    worst = 0.0
    for case in range(args.cases):
        mx = rnd.random() < 0.4
        # (up to 2048 rows the bf16 kernel takes its small-M tile; 2100 / 4224: the 96- and 128-row tiles)
        M = rnd.choice([1, 3, 16, 31, 64, 65, 96, 127, 128, 129, 200, 324, 500, 777, 1296, 2100, 4224])
        N = rnd.choice([4, 8, 12, 16, 20, 24, 32, 36, 64, 96, 100, 128, 136, 160, 256, 260, 384, 512])
        K = rnd.choice([128, 256, 384, 512, 1024]) if mx else rnd.choice([8, 16, 24, 64, 72, 128, 192, 256, 520, 1024])
        epi = rnd.choice([ops.EPI_NONE, ops.EPI_BIAS_RES, ops.EPI_BIAS_GELU] + ([] if mx else [ops.EPI_DGELU]))
        od = torch.float32 if epi == ops.EPI_BIAS_RES else rnd.choice([torch.float32, torch.bfloat16])
        g = torch.Generator().manual_seed(case)
        a = torch.randn(M, K, generator=g)
        b = torch.randn(N, K, generator=g) / K ** 0.5
        bias = torch.randn(N, generator=g) if epi in (ops.EPI_BIAS_RES, ops.EPI_BIAS_GELU) or rnd.random() < 0.3 else None
        if epi == ops.EPI_DGELU:
            bias = None
        res = torch.randn(M, N, generator=g) if epi == ops.EPI_BIAS_RES else None
        aux_in = torch.randn(M, N, generator=g).to(od) if epi == ops.EPI_DGELU else None
        want_image = mx and epi == ops.EPI_BIAS_GELU and N % 32 == 0 and rnd.random() < 0.7
        if mx:
            aq, as_ = ops.quant_mx8(a.cuda())
            bq, bs = ops.quant_mx8(b.cuda())
            ar, br = oracle.mx8_dequant(aq, as_), oracle.mx8_dequant(bq, bs)
            out = ops.gemm_mx8(aq, as_, bq, bs, out_dtype=od, epilogue=epi, bias=None if bias is None else bias.cuda(),
                               residual=None if res is None else res.cuda(), want_image=want_image)
        else:
            ar, br = a.bfloat16().float(), b.bfloat16().float()
            out = ops.gemm(a.bfloat16().cuda(), b.bfloat16().cuda(), out_dtype=od, epilogue=epi,
                           bias=None if bias is None else bias.cuda(), residual=None if res is None else res.cuda(),
                           aux=None if aux_in is None else aux_in.cuda())
        ref = (ar.double() @ br.double().t()).float()
        if bias is not None:
            ref = ref + bias
        u_ref = ref
        if epi == ops.EPI_BIAS_RES:
            ref = ref + res
        elif epi == ops.EPI_BIAS_GELU:
            ref = gelu(u_ref)
        elif epi == ops.EPI_DGELU:
            ref = ref * dgelu(aux_in.float())
        outs = out if isinstance(out, tuple) else (out,)
        c = outs[0].float().cpu()
        tol = (2e-2 if od == torch.bfloat16 else 6e-3) if not mx else (2.5e-2 if od == torch.bfloat16 else 1.2e-2)
        scale = ref.abs().max().item() + 1e-6
        err = (c - ref).abs().max().item() / scale
        ok = err < tol
        if epi == ops.EPI_BIAS_GELU:
            u = outs[1].float().cpu()
            eu = (u - u_ref).abs().max().item() / (u_ref.abs().max().item() + 1e-6)
            ok = ok and eu < tol
            err = max(err, eu)
        if want_image:  # the image is taken from the fp32 values before C's own rounding
            if od == torch.float32:
                q_ref, s_ref = oracle.mx8_quant(outs[0])
                ok = ok and torch.equal(outs[2].cpu(), q_ref) and torch.equal(outs[3].cpu(), s_ref)
            else:
                deq = oracle.mx8_dequant(outs[2], outs[3])
                ok = ok and (deq - c).abs().max().item() / scale < 0.13  # 2^-4 rounding, 12 % in the saturating corner (449..511 -> 448)
        worst = max(worst, err / tol)
        assert ok, f"FAIL case {case} (seed {args.seed}): mx={mx} M={M} N={N} K={K} epi={epi} out={od} err={err:.3g} tol={tol}"
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    try:
        worst = run(args.cases, args.seed)
    except AssertionError as e:
        print(e)
        sys.exit(1)
    print(f"{args.cases} cases ok; worst error / tolerance = {worst:.2f}")


if __name__ == "__main__":
    main()
