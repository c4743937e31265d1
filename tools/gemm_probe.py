import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import avformer_amd as A
ops = A.ops
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    A._lib.timing_enable(True)
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    A._lib.timing_enable(False)
    tm = A._lib.timing_read()
    return sum(v["ms"] for v in tm.values()) / iters * 1e-3
shapes = [(10368, 1536, k) for k in (64, 128, 256, 512, 1024)] + [(10368, 512, k) for k in (64, 256, 512, 1024, 1536)]
for (m, n, k) in shapes:
    a = torch.randn(m, k, device="cuda").bfloat16()
    b = (torch.randn(n, k, device="cuda") / k ** 0.5).bfloat16()
    t = timeit(lambda: ops.gemm(a, b, out_dtype=torch.bfloat16))
    res = torch.randn(m, n, device="cuda"); bias = torch.randn(n, device="cuda")
    t2 = timeit(lambda: ops.gemm(a, b, out_dtype=torch.float32, epilogue=ops.EPI_BIAS_RES, bias=bias, residual=res))
    print(f"M={m} N={n} K={k}: plain bf16 out {t*1e6:7.1f} us ({2.0*m*n*k/t/1e12:6.1f} TF/s) | bias+res f32 out {t2*1e6:7.1f} us")
