#!/usr/bin/env python3
"""How fast does the vendor library (torch.matmul -> hipBLASLt) run the path's plain NT GEMM shapes?  A reference point for
gemm_bf16_nt_glds_kernel (no epilogues on either side); timed with torch events over 50 back-to-back launches."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import avformer_amd as A
ops = A.ops


def t_events(fn, iters=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


for M in (10368, 16384):
    for (n, k) in ((1536, 512), (512, 512), (1024, 512), (512, 1024), (512, 1536)):
        a = torch.randn(M, k, device="cuda").bfloat16()
        w = (torch.randn(n, k, device="cuda") / k ** 0.5).bfloat16()
        out = torch.empty(M, n, device="cuda", dtype=torch.bfloat16)
        tl = t_events(lambda: torch.matmul(a, w.t(), out=out))
        to = t_events(lambda: ops.gemm(a, w, out_dtype=torch.bfloat16))
        f = 2.0 * M * n * k
        print(f"M={M} N={n} K={k}: hipBLASLt {tl * 1e6:6.1f} us {f / tl / 1e12:6.1f} TF/s | ours {to * 1e6:6.1f} us {f / to / 1e12:6.1f} TF/s")
