import sys, torch
sys.path.insert(0, ".")
import avformer_amd as A
ops = A.ops
M, N, K = 10368, 1024, 512
a = torch.randn(M, K, device="cuda").bfloat16(); w = torch.randn(N, K, device="cuda").bfloat16() * 0.05
b = torch.randn(N, device="cuda"); u = torch.randn(M, N, device="cuda").bfloat16()
def timed(fn, n=60):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n
print("gelu", round(timed(lambda: ops.gemm(a, w, epilogue=ops.EPI_BIAS_GELU, bias=b)), 1), "dgelu", round(timed(lambda: ops.gemm(a, w, epilogue=ops.EPI_DGELU, aux=u)), 1),
      "plain", round(timed(lambda: ops.gemm(a, w)), 1))
