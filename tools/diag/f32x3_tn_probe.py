import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import avformer_amd as A
from avformer_amd import _lib
from tools.bench_mx8 import timeit
ops = A.ops
K = int(sys.argv[1]) if len(sys.argv) > 1 else 10368
for (M, N) in [(1024, 512), (512, 1024), (1024, 1024), (512, 512), (1536, 512), (512, 1536), (1024, 640), (640, 1024), (1024, 516), (2048, 512)]:
    a = torch.randn(K, M, device="cuda"); b = torch.randn(K, N, device="cuda")
    t = timeit(lambda: ops.gemm(a, b, trans_a=True, trans_b=False), iters=20)
    print(f"TN {M}x{N}x{K}: {t:8.1f} us {2.0*M*N*K/t/1e6:6.1f} TF/s", flush=True)
