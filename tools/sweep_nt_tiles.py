"""Times the bf16 NT GEMM on the layer's eight shapes (forward + dX) under the tile configuration AVF_NT_TILE forces.

    for t in 0 1 2 3; do AVF_NT_TILE=$t python tools/sweep_nt_tiles.py --rows 10368; done
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import avformer_amd as A  # noqa: E402
from tools.bench_mx8 import timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10368)
    ap.add_argument("--dim", type=int, default=512)
    ap.add_argument("--mlp", type=int, default=1024)
    args = ap.parse_args()
    M, D, H = args.rows, args.dim, args.mlp
    ops = A.ops
    shapes = [("qkv", 3 * D, D, ops.EPI_NONE, torch.bfloat16), ("out+res", D, D, ops.EPI_BIAS_RES, torch.float32),
              ("mlp1+gelu", H, D, ops.EPI_BIAS_GELU, torch.bfloat16), ("mlp2+res", D, H, ops.EPI_BIAS_RES, torch.float32),
              ("dx_w2+dgelu", H, D, ops.EPI_DGELU, torch.bfloat16), ("dx_w1", D, H, ops.EPI_NONE, torch.bfloat16),
              ("dx_out", D, D, ops.EPI_NONE, torch.bfloat16), ("dx_qkv", D, 3 * D, ops.EPI_NONE, torch.bfloat16)]
    out = []
    for name, N, K, epi, od in shapes:
        a = torch.randn(M, K, device="cuda").bfloat16()
        b = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
        bias = torch.randn(N, device="cuda") if epi in (ops.EPI_BIAS_RES, ops.EPI_BIAS_GELU) else None
        res = torch.randn(M, N, device="cuda") if epi == ops.EPI_BIAS_RES else None
        aux = torch.randn(M, N, device="cuda").bfloat16() if epi in (ops.EPI_DGELU, ops.EPI_BIAS_GELU) else None
        t = timeit(lambda: ops.gemm(a, b, out_dtype=od, epilogue=epi, bias=bias, residual=res, aux=aux))
        out.append(f"{name} {t:5.1f}")
    print(f"tile {os.environ.get('AVF_NT_TILE', 'auto'):>4s} M={M}: " + "  ".join(out))


if __name__ == "__main__":
    main()
