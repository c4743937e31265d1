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
for n, k in ((1536, 512), (512, 512), (512, 1536)):
    for m in (2592, 5184, 10368, 20736, 41472, 82944):
        a = torch.randn(m, k, device="cuda").bfloat16()
        b = (torch.randn(n, k, device="cuda") / k ** 0.5).bfloat16()
        t = timeit(lambda: ops.gemm(a, b, out_dtype=torch.bfloat16))
        print(f"M={m:6d} N={n} K={k}: {t*1e6:8.1f} us {2.0*m*n*k/t/1e12:7.1f} TF/s   tiles128={((m+127)//128)*((n+127)//128)}")
