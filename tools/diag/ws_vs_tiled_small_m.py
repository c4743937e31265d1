#!/usr/bin/env python3
"""Diagnostic: persistent (weight-stationary) against tiled NT kernel for K = 512 at row counts between 2048 and C2's, with the
weight image COLD as in the step (1 GiB written before every launch, A rewritten after it).  Two processes under rocprofv3
(tools/diag/ws_vs_tiled_small_m.sh): `ws` calls ops.gemm_ws, `tiled` runs ops.gemm with AVF_TUNING=1 AVF_NT_WS=0; --summarise
prints the median kernel time per (rows, N) of one trace."""
import os
import statistics
import sys

MS = (2048, 2592, 4096, 5184, 7776, 10368)
NS = (512, 1024, 1536)
IT = 12


def run(mode):
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import avformer_amd as A
    ops = A.ops
    flush = torch.empty(1 << 28, device="cuda", dtype=torch.float32)
    for M in MS:
        for N in NS:
            src = torch.randn(M, 512, device="cuda").bfloat16()
            a = src.clone()
            w = (torch.randn(N, 512, device="cuda") / 512 ** 0.5).bfloat16()
            wp = ops.pack_ws(w)
            for _ in range(IT):
                flush.fill_(1.0); a.copy_(src)
                if mode == "ws":
                    ops.gemm_ws(a, wp, N, out_dtype=torch.bfloat16)
                else:
                    ops.gemm(a, w, out_dtype=torch.bfloat16)
            torch.cuda.synchronize()


def summarise(path):
    import csv
    rows = [r for r in csv.DictReader(open(path)) if "gemm_bf16_nt" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    assert len(rows) == len(MS) * len(NS) * IT, len(rows)
    i = 0
    for M in MS:
        for N in NS:
            d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3 for r in rows[i:i + IT]]
            kinds = {("ws" if "nt_ws" in r["Kernel_Name"] else "tiled") for r in rows[i:i + IT]}
            i += IT
            T = (M + 31) // 32
            G = min(256 // (N // 256), T)
            print(f"rows {M:6d} N {N:5d}: {'/'.join(sorted(kinds)):6s} {statistics.median(d[2:]):6.2f} us   (row tiles per persistent workgroup {T / G:4.1f})")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
        summarise(sys.argv[2])
    else:
        run(sys.argv[1] if len(sys.argv) > 1 else "ws")
