#!/usr/bin/env python3
"""Diagnostic: what does the simplest streaming kernel take for the bytes a LayerNorm launch moves?  (rocprofv3 kernel trace;
bash tools/diag/qp_any.sh copyfloor tools/diag/copy_floor.py)  a.copy_(b) of R x 512 bf16: R*512*2 read + the same written =
the traffic of ln_fwd_row8 at that R; add3 = a + b + c -> d: three reads + one write = ln_bwd_row8's."""
import sys
import torch

for R in (10368, 16384):
    a = torch.randn(R, 512, device="cuda").bfloat16()
    b = torch.empty_like(a)
    c = torch.randn_like(a)
    d = torch.randn_like(a)
    out = torch.empty_like(a)
    for _ in range(40):
        b.copy_(a)                      # elementwise copy kernel: 1 read + 1 write
    for _ in range(40):
        torch.add(a, c, out=out)        # 2 reads + 1 write
    for _ in range(40):
        torch.addcmul(a, c, d, out=out)  # 3 reads + 1 write
    torch.cuda.synchronize()
