"""How does the tiled NT kernel's time depend on the number of workgroups per CU?  dX of to_qkv (N = 512, K = 1536, plain)
under one forced tile (AVF_TUNING=1 AVF_NT_TILE=5: 96 x 128 tiles, 2 workgroups per CU fit), row counts chosen so that the grid is
256 (one per CU), 384, 432 (C2), 512 (two per CU) and 768 workgroups.  Back to back, hot operands: the isolated kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import avformer_amd as A  # noqa: E402
from tools.bench_mx8 import timeit  # noqa: E402

ops = A.ops
N, K = 512, int(os.environ.get("K", "1536"))
bm = int(os.environ.get("BM", "96"))
for wgs in (128, 256, 320, 384, 432, 448, 512, 576, 640, 768, 1024):
    M = wgs // 4 * bm
    a = torch.randn(M, K, device="cuda").bfloat16()
    b = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    t = timeit(lambda: ops.gemm(a, b, out_dtype=torch.bfloat16))
    print(f"tile {os.environ.get('AVF_NT_TILE', 'auto')} K={K} workgroups {wgs:5d} M={M:6d}: {t:6.2f} us  {2.0 * M * N * K / t / 1e6:7.1f} TFLOP/s", flush=True)
