#!/usr/bin/env python3
"""Diagnostic: what would running a layer's fold on a side branch of the captured graph, beside the NEXT layer's first GEMM, buy?

Proxy: main branch = the persistent dGELU GEMM of C2 / C3 (ops.gemm_ws, EPI_DGELU + column sums), side branch = the sum of four fp32
slabs of 2.1 M floats (the traffic of fold_group_kernel: 33.5 MB read, 8.4 MB written).  Both captured into one hipGraph, serial
and forked; us per replay of each.   python tools/diag/fold_overlap_probe.py [rows]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import avformer_amd as A  # noqa: E402

ops = A.ops
R = int(sys.argv[1]) if len(sys.argv) > 1 else 10368
a = torch.randn(R, 512, device="cuda").bfloat16()
w = (torch.randn(1024, 512, device="cuda") / 512 ** 0.5).bfloat16()
wp = ops.pack_ws(w)
u = torch.randn(R, 1024, device="cuda").bfloat16()
slabs = torch.randn(4, 2100000, device="cuda")
out = torch.empty(2100000, device="cuda")


def gemm():
    return ops.gemm_ws(a, wp, 1024, out_dtype=torch.bfloat16, epilogue=A._lib.EPI_DGELU, aux=u, want_colsum=True)


def fold():
    torch.sum(slabs, dim=0, out=out)


def capture(forked, reps=6):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    side = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        gemm(); fold()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        cur = torch.cuda.current_stream()
        for _ in range(reps):  # fold of "layer l" beside the GEMM of "layer l-1"; a serial kernel in between (the rest of the layer)
            if forked:
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    fold()
                gemm()
                cur.wait_stream(side)
            else:
                fold()
                gemm()
            gemm()  # (stands for the other launches of the layer: nothing overlaps it)
    return g, reps


def t(g, reps, iters=50):
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters / reps


gs, r = capture(False)
gf, _ = capture(True)
for _ in range(2):
    a_, b_ = t(gs, r), t(gf, r)
    print(f"{R} rows: serial {a_:7.2f} us per (fold + 2 GEMMs), forked {b_:7.2f} us  -> {a_ - b_:5.2f} us hidden per layer")
