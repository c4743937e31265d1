#!/usr/bin/env python3
"""Attention operator check + timing on the MI355X (tuning aid).  Pre-scaled q (the layer path's form).
usage: python tools/diag/attn_ab.py [--shapes B,N,H ...] [--check]     (AVF_ATTN_MERGED=0 and AVF_ATTN_MERGED_MIN_N select the backward)"""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import avformer_amd as A  # noqa: E402

ops = A.ops


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    A._lib.timing_enable(True)
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    A._lib.timing_enable(False)
    tm = A._lib.timing_read()
    return sum(v["ms"] for v in tm.values()) / iters * 1e-3


def ref64(qkv, B, N, H, dh, d_o):
    I = H * dh
    x = qkv.double().view(B, N, 3, H, dh).permute(2, 0, 3, 1, 4).contiguous().requires_grad_(True)
    q, k, v = x[0], x[1], x[2]
    s = q @ k.transpose(-1, -2) / math.sqrt(dh)
    p = torch.softmax(s, -1)
    o = (p @ v).permute(0, 2, 1, 3).reshape(B * N, I)
    o.backward(d_o.double())
    g = x.grad.permute(1, 3, 0, 2, 4).reshape(B * N, 3 * I)
    return o.detach(), g


def rel(a, b):
    return ((a.double().cpu() - b).norm() / b.norm()).item()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", nargs="*", default=["32,512,8", "32,324,8", "64,512,8"])
    ap.add_argument("--check", action="store_true")
    a = ap.parse_args()
    dh = 64
    c = math.log2(math.e) / math.sqrt(dh)
    for sh in a.shapes:
        B, N, H = (int(v) for v in sh.split(","))
        I = H * dh
        g = torch.Generator().manual_seed(N + H)
        qkv = torch.randn(B * N, 3 * I, generator=g).to(torch.bfloat16)
        d_o = torch.randn(B * N, I, generator=g).to(torch.bfloat16)
        dev = qkv.float().clone()
        dev[:, :I] = (dev[:, :I] * c).to(torch.bfloat16).float()
        ref_in = dev.clone()
        ref_in[:, :I] = ref_in[:, :I] / c
        qd = dev.to(torch.bfloat16).cuda()
        gd = d_o.cuda()
        o, lse = ops.attn_fwd(qd, B, N, H, dh, q_prescaled=True)
        dqkv = ops.attn_bwd(qd, o, gd, lse, B, N, H, dh, q_prescaled=True)
        torch.cuda.synchronize()
        msg = f"B={B} N={N} H={H}:"
        if a.check:
            Bc = min(B, 2)
            o_ref, g_ref = ref64(ref_in[: Bc * N], Bc, N, H, dh, d_o[: Bc * N].float())
            got = dqkv[: Bc * N]
            msg += f" err o {rel(o[:Bc * N], o_ref):.2e} dq {rel(got[:, :I], g_ref[:, :I]):.2e} dk {rel(got[:, I:2*I], g_ref[:, I:2*I]):.2e} dv {rel(got[:, 2*I:], g_ref[:, 2*I:]):.2e}"
            msg += f" finite {bool(torch.isfinite(dqkv.float()).all())}"
        fl = 4.0 * B * H * N * N * dh
        tf = timeit(lambda: ops.attn_fwd(qd, B, N, H, dh, q_prescaled=True))
        tb = timeit(lambda: ops.attn_bwd(qd, o, gd, lse, B, N, H, dh, q_prescaled=True))
        msg += f"  fwd {tf * 1e6:7.1f} us ({fl / tf / 2.5e15:.3f})  bwd {tb * 1e6:7.1f} us ({2.5 * fl / tb / 2.5e15:.3f})"
        print(msg, flush=True)


if __name__ == "__main__":
    main()
