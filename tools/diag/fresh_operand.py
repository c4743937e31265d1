#!/usr/bin/env python3
"""Diagnostic: does an NT GEMM run slower when its A operand was written by the kernel just before it (as in the step) than
when the same buffer is re-read launch after launch (as in a standalone timing loop)?

The per-shape tables of the step (profiles/r05_c2_shapes.csv) show the tiled kernel 20 - 25 % slower than the standalone loop
of tools/vendor_yardstick.py on the same shapes, the persistent kernel not.  Three states of A for every shape:
  resident  the loop re-reads one buffer (whatever cache level holds it keeps it)
  fresh     a copy kernel rewrites A right before every GEMM launch (dirty lines of all eight L2s written back at its end)
  cold      as fresh, then 1 GiB of other memory is written before the GEMM (nothing of A or W left in any cache)
  coldW     the 1 GiB write first, then A rewritten: A as in fresh, the weight out of every cache (in the step a weight image
            was last read a whole pass earlier)
  coldA     A rewritten, the 1 GiB write, then the weight image read by two small kernels
  coldW+touch  as coldW, then the weight image read by two small kernels before the GEMM: does a touch by an EARLIER kernel (the
            lines then sit in the memory-side cache, not in the GEMM's L2s) remove the penalty?
Run under rocprofv3 (tools/diag/fresh_operand.sh): kernel durations from the trace, median of 30 launches per shape and state.
"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import avformer_amd as A  # noqa: E402

ops = A.ops


ITERS, WARM = 30, 5
SHAPES = (("to_qkv (ws)", 1536, 512), ("d_o (ws)", 512, 512), ("net.0 (ws)", 1024, 512), ("dh2 (tiled)", 512, 1024),
          ("dX of to_qkv (tiled)", 512, 1536))
STATES = ("resident", "fresh", "cold", "coldW", "coldA", "coldW+touch")


def run(R):
    """the launches, in a fixed order: for every shape, for every state, WARM + ITERS x [state's writer kernels; GEMM]"""
    flush = torch.empty(1 << 28, device="cuda", dtype=torch.float32)  # 1 GiB
    for name, n, k in SHAPES:
        src = torch.randn(R, k, device="cuda").bfloat16()
        a = src.clone()
        w = (torch.randn(n, k, device="cuda") / k ** 0.5).bfloat16()
        if k == 512 and ops.gemm_ws_used(R, n, k):
            wp = ops.pack_ws(w)
            wt = wp
            fn = lambda: ops.gemm_ws(a, wp, n, out_dtype=torch.bfloat16)
        else:
            wt = w
            fn = lambda: ops.gemm(a, w, out_dtype=torch.bfloat16)
        torch.cuda.synchronize()
        for state in STATES:
            for _ in range(WARM + ITERS):
                if state in ("coldW", "coldW+touch"):
                    flush.fill_(1.0)
                if state != "resident":
                    a.copy_(src)
                if state in ("cold", "coldA"):
                    flush.fill_(1.0)
                if state in ("coldA", "coldW+touch"):
                    wt.float().sum()  # two small kernels read the weight image again
                fn()
            torch.cuda.synchronize()


def summarise(trace_csv, R):
    """rocprofv3 --kernel-trace CSV of run(): the GEMM dispatches in start order are SHAPES x STATES x (WARM + ITERS)"""
    import csv
    rows = [r for r in csv.DictReader(open(trace_csv)) if "gemm_bf16_nt" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    per = WARM + ITERS
    assert len(rows) == len(SHAPES) * len(STATES) * per, (len(rows), per)
    print(f"{R} rows; kernel duration in us (rocprofv3 kernel trace, median of {ITERS} launches), by the state of the A operand")
    i = 0
    for name, n, k in SHAPES:
        out = []
        for state in STATES:
            d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3 for r in rows[i + WARM:i + per]]
            i += per
            out.append(f"{state} {statistics.median(d):6.2f}")
        print(f"  {name:24s} N={n:5d} K={k:5d}: " + "   ".join(out))


EPI_STATES = ("resident", "fresh", "coldX", "coldX+coldW")


def run_epi(R):
    """the epilogue forms whose EXTRA operand is cold in the step: dGELU reads the pre-activation u saved by the forward pass
    (persistent kernel, aux staged by LDS-DMA); bias + residual reads a residual stream written a few launches earlier (tiled
    kernel, K = 1024).  coldX: A rewritten, then the 1 GiB write with the extra operand out of cache, then A rewritten AGAIN and the
    weight image touched - only the extra operand is cold."""
    flush = torch.empty(1 << 28, device="cuda", dtype=torch.float32)
    src = torch.randn(R, 512, device="cuda").bfloat16()
    a = src.clone()
    w = (torch.randn(1024, 512, device="cuda") / 512 ** 0.5).bfloat16()
    wp = ops.pack_ws(w)
    u = torch.randn(R, 1024, device="cuda").bfloat16()
    src2 = torch.randn(R, 1024, device="cuda").bfloat16()
    a2 = src2.clone()
    w2 = (torch.randn(512, 1024, device="cuda") / 1024 ** 0.5).bfloat16()
    res = torch.randn(R, 512, device="cuda").bfloat16()
    bias = torch.zeros(512, device="cuda")
    cases = (
        (lambda: ops.gemm_ws(a, wp, 1024, out_dtype=torch.bfloat16, epilogue=A._lib.EPI_DGELU, aux=u, want_colsum=True), a, src, wp),
        (lambda: ops.gemm(a2, w2, out_dtype=torch.bfloat16, epilogue=A._lib.EPI_BIAS_RES, bias=bias, residual=res), a2, src2, w2),
    )
    for fn, av, sv, wt in cases:
        torch.cuda.synchronize()
        for state in EPI_STATES:
            for _ in range(WARM + ITERS):
                if state != "resident":
                    av.copy_(sv)
                if state.startswith("coldX"):
                    flush.fill_(1.0)
                    av.copy_(sv)
                    if state == "coldX":
                        wt.float().sum()
                fn()
            torch.cuda.synchronize()


def summarise_epi(trace_csv, R):
    import csv
    rows = [r for r in csv.DictReader(open(trace_csv)) if "gemm_bf16_nt" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    per = WARM + ITERS
    names = ("dGELU (ws, N=1024 K=512, aux u)", "bias+res (tiled, N=512 K=1024, residual)")
    assert len(rows) == len(names) * len(EPI_STATES) * per, (len(rows), per)
    print(f"{R} rows; kernel duration in us (median of {ITERS} launches), by the state of the epilogue's extra operand (coldX: only it is cold)")
    i = 0
    for name in names:
        out = []
        for state in EPI_STATES:
            d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3 for r in rows[i + WARM:i + per]]
            i += per
            out.append(f"{state} {statistics.median(d):6.2f}")
        print(f"  {name:44s}: " + "   ".join(out))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--epi":
        run_epi(int(sys.argv[2]))
        sys.exit(0)
    if len(sys.argv) > 3 and sys.argv[1] == "--summarise-epi":
        summarise_epi(sys.argv[2], int(sys.argv[3]))
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
        summarise(sys.argv[2], int(sys.argv[3]))
    else:
        run(int(sys.argv[1]) if len(sys.argv) > 1 else 10368)
