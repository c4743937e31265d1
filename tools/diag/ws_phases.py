"""Phase stamps of the weight-stationary persistent NT GEMM (csrc/gemm_ws.hip, diagnostic build lib/ws_dbg.so).

    tools/diag/build_ws_dbg.sh && AVF_LIB_PATH=<pkg>/lib/ws_dbg.so python tools/diag/ws_phases.py --rows 16384 --n 1024 --epi gelu

Prints, for waves 0 (early half) and 4 (late half), the median over the workgroups of every phase in shader cycles
(s_memtime): prologue (DMA issue + weight loads issued; their arrival), then per tile: barrier wait, DMA issue, the 64 MFMAs,
the vmcnt wait, the late half's barrier, the epilogue.
"""
import argparse
import ctypes as C
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import avformer_amd as A  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=16384)
    ap.add_argument("--n", type=int, default=1024)
    ap.add_argument("--epi", default="gelu", choices=["none", "res", "gelu", "dgelu"])
    args = ap.parse_args()
    ops = A.ops
    lib = A._lib.load()
    bf = torch.bfloat16
    M, N, K = args.rows, args.n, 512
    epi = {"none": ops.EPI_NONE, "res": ops.EPI_BIAS_RES, "gelu": ops.EPI_BIAS_GELU, "dgelu": ops.EPI_DGELU}[args.epi]
    a = torch.randn(M, K, device="cuda").to(bf)
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(bf)
    wp = ops.pack_ws(w)
    bias = torch.randn(N, device="cuda") if epi in (ops.EPI_BIAS_RES, ops.EPI_BIAS_GELU) else None
    res = torch.randn(M, N, device="cuda").to(bf) if epi == ops.EPI_BIAS_RES else None
    aux = torch.randn(M, N, device="cuda").to(bf) if epi == ops.EPI_DGELU else None
    for _ in range(20):
        ops.gemm_ws(a, wp, N, out_dtype=bf, epilogue=epi, bias=bias, residual=res, aux=aux)
    torch.cuda.synchronize()
    NS = 64
    nwg = 256
    NWV = 8
    buf = (C.c_uint64 * (nwg * NWV * NS))()
    fn = lib.avf_ws_stamps_read
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_size_t]
    assert fn(buf, C.sizeof(buf)) == 0
    st = [[[buf[(g * NWV + h) * NS + i] for i in range(NS)] for h in range(NWV)] for g in range(nwg)]
    live = [g for g in range(nwg) if st[g][0][0] != 0]
    ntile = max(sum(1 for t in range(10) if st[g][0][4 + 6 * t] != 0) for g in live)
    full = [g for g in live if st[g][0][4 + 6 * (ntile - 1)] != 0]
    print(f"M={M} N={N} epi={args.epi}: {len(live)} live workgroups, up to {ntile} tiles ({len(full)} workgroups with that many)")
    med = lambda xs: int(statistics.median(xs))
    # stamps 62 / 63: s_memrealtime (100 MHz) beside stamps 0 / 3 (s_memtime): the rate of the cycle counter during this kernel
    rt = [(st[g][0][3] - st[g][0][0]) / max(1, st[g][0][63] - st[g][0][62]) * 100.0 for g in live if st[g][0][63] > st[g][0][62]]
    if rt:
        print(f"s_memtime ticks at {statistics.median(rt):.0f} MHz during this kernel (against the 100 MHz s_memrealtime)")
    for h, nm in [(w, f"wave {w} ({'early' if w < 4 else 'late'})") for w in range(NWV)]:
        d = lambda i, j: med([st[g][h][j] - st[g][h][i] for g in full])
        print(f"{nm}: total {d(0, 3)}  issue(DMA+W) {d(0, 1)}  W/tile0 arrival {d(1, 2)}  to first tile top {d(2, 4)}")
        for t in range(ntile):
            b = 4 + 6 * t
            nxt = (4 + 6 * (t + 1)) if t + 1 < ntile else 3
            print(f"   tile {t}: barrier {d(b, b + 1):6d}  dma-issue {d(b + 1, b + 2):5d}  mfma {d(b + 2, b + 3):6d}  vmcnt {d(b + 3, b + 4):6d}"
                  f"  late-barrier {d(b + 4, b + 5):6d}  epilogue {d(b + 5, nxt):6d}   | period {d(b, nxt):6d}")
    # arrival order at the tile barriers: per tile, when each wave reaches its barrier relative to the first arriver (median)
    for t in range(1, ntile):
        arr = []
        for w in range(NWV):
            idx = (4 + 6 * t) if w < 4 else (8 + 6 * (t - 1))  # early waves: top of tile t; late waves: behind the MFMAs of t - 1
            arr.append(med([st[g][w][idx] - min(st[g][v][(4 + 6 * t) if v < 4 else (8 + 6 * (t - 1))] for v in range(NWV)) for g in full]))
        print(f"   barrier {t}: arrival after the first wave:", arr)
    # skew of the workgroups: start and end relative to the first start
    s0 = min(st[g][0][0] for g in live)
    print("workgroup start (min/med/max):", min(st[g][0][0] - s0 for g in live), med([st[g][0][0] - s0 for g in live]),
          max(st[g][0][0] - s0 for g in live), " end:", min(st[g][0][3] - s0 for g in live), med([st[g][0][3] - s0 for g in live]),
          max(st[g][0][3] - s0 for g in live))


if __name__ == "__main__":
    main()
