"""bf16 vs MX-FP8 NT GEMM over K at fixed M, N (back to back, us per launch): separates the per-K-step cost of the two kernels
from their fixed cost (launch, prologue, epilogue)."""
import sys
import torch
sys.path.insert(0, ".")
import avformer_amd as A  # noqa: E402

ops = A.ops
DEV = "cuda:0"


def timed(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


for M, N in ((16384, 1024), (32768, 512)):
    for K in (512, 1024, 2048, 4096):
        a = torch.randn(M, K, device=DEV).bfloat16()
        w = torch.randn(N, K, device=DEV).bfloat16()
        aq, asc = ops.quant_mx8(a)
        wq, wsc = ops.quant_mx8(w)
        tb = timed(lambda: ops.gemm(a, w))
        tm = timed(lambda: ops.gemm_mx8(aq, asc, wq, wsc, out_dtype=torch.bfloat16))
        fl = 2.0 * M * N * K
        print(f"M={M} N={N} K={K}: bf16 {tb:.1f} us ({fl / tb / 1e6:.0f} TFLOP/s), mx8 {tm:.1f} us ({fl / tm / 1e6:.0f} TFLOP/s)", flush=True)
