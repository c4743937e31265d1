#!/usr/bin/env python3
"""Diagnostic: shader clock and package power while ONE hot-path kernel runs back to back (C2 or C3 shapes).

The replayed step runs against the package power limit (amd-smi: PPT violation active; shader clock 2.15 - 2.37 GHz of 2.4,
DESIGN.md section 10).  The firmware does not move the clock per kernel, so the kernels that draw the most set the clock of
the whole step; this prints which ones those are.  rocm-smi is sampled from a thread while the launches run.

  python tools/diag/kernel_power.py [c2|c3]
"""
import os
import re
import statistics
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import avformer_amd as A  # noqa: E402

ops = A.ops


def sample(stop, out):
    while not stop.is_set():
        try:
            txt = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
            m = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", txt)
            w = re.search(r"Package Power \(W\): ([0-9.]+)", txt)
            if m and w:
                out.append((int(m.group(1)), float(w.group(1))))
        except Exception:
            pass
        time.sleep(0.3)


def measure(name, fn, flops, seconds=4.0):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    stop, out = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, out))
    n = 0
    t0 = time.perf_counter()
    th.start()
    while time.perf_counter() - t0 < seconds:
        for _ in range(200):
            fn()
        n += 200
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stop.set()
    th.join()
    out = out[2:] or out  # the first samples see the ramp
    clk = statistics.median(c for c, _ in out) if out else 0
    pw = statistics.median(p for _, p in out) if out else 0
    us = dt / n * 1e6
    tf = f"{flops / (dt / n) / 1e12:7.1f} TF/s" if flops else "            "
    print(f"  {name:34s} {us:8.2f} us/launch {tf}   sclk {clk:5.0f} MHz   package {pw:6.0f} W   ({len(out)} samples)", flush=True)


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
    B, N = {"c2": (32, 324), "c3": (32, 512)}[cfg]
    R, H, dh, D, Mlp = B * N, 8, 64, 512, 1024
    print(f"{cfg}: {R} rows; every kernel alone, back to back (host-launched: short kernels include launch gaps)")
    bf = lambda *s: torch.randn(*s, device="cuda").bfloat16()
    w = lambda n, k: (torch.randn(n, k, device="cuda") / k ** 0.5).bfloat16()
    # NT GEMMs
    a512, a1024, a1536 = bf(R, 512), bf(R, 1024), bf(R, 1536)
    wq = ops.pack_ws(w(1536, 512))
    measure("ws NT  N=1536 K=512 (to_qkv)", lambda: ops.gemm_ws(a512, wq, 1536, out_dtype=torch.bfloat16), 2.0 * R * 1536 * 512)
    w2 = w(512, 1024)
    measure("tiled NT N=512 K=1024 (dh2)", lambda: ops.gemm(a1024, w2, out_dtype=torch.bfloat16), 2.0 * R * 512 * 1024)
    w3 = w(512, 1536)
    measure("tiled NT N=512 K=1536 (dX of to_qkv)", lambda: ops.gemm(a1536, w3, out_dtype=torch.bfloat16), 2.0 * R * 512 * 1536)
    # weight gradients
    shapes = ((512, 1536), (512, 512), (512, 1024), (1024, 512))
    pairs = [(bf(R, m), bf(R, n)) for m, n in shapes]
    measure("grouped TN (four weight gradients)", lambda: ops.gemm_tn_group(pairs), 2.0 * R * sum(m * n for m, n in shapes))
    # attention
    I = H * dh
    qkv = (torch.randn(R, 3 * I, device="cuda") * 0.5).bfloat16()
    o, lse2 = ops.attn_fwd(qkv, B, N, H, dh, q_prescaled=True)
    d_o = torch.randn_like(o)
    measure("attention forward", lambda: ops.attn_fwd(qkv, B, N, H, dh, q_prescaled=True), 4.0 * B * H * N * N * dh)
    measure("attention backward (merged)", lambda: ops.attn_bwd(qkv, o, d_o, lse2, B, N, H, dh, q_prescaled=True), 10.0 * B * H * N * N * dh)
    # LayerNorm (fp32-stream entry points: the bandwidth-bound class)
    x = torch.randn(R, D, device="cuda")
    g, b = torch.ones(D, device="cuda"), torch.zeros(D, device="cuda")
    y, mean, rstd = ops.layernorm_fwd(x, g, b)
    dy = torch.randn(R, D, device="cuda")
    measure("LayerNorm backward (fp32 stream)", lambda: ops.layernorm_bwd(dy, x, g, mean, rstd), 0)


if __name__ == "__main__":
    main()
