import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
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
for (m, n, k) in [(10368, 1536, 512), (10368, 1536, 2048), (10368, 1536, 8192), (10368, 512, 8192), (16384, 1024, 8192), (16384, 2048, 4096), (4096, 4096, 4096)]:
    a = torch.randn(m, k, device="cuda").bfloat16()
    b = (torch.randn(n, k, device="cuda") / k ** 0.5).bfloat16()
    t = timeit(lambda: ops.gemm(a, b, out_dtype=torch.bfloat16))
    print(f"M={m} N={n} K={k}: {t*1e6:8.1f} us {2.0*m*n*k/t/1e12:7.1f} TF/s")
