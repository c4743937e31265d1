"""fp32 NT GEMM (the parity mode's kernels) back to back on a few shapes; round-5 record: the general 64 x 64 kernel reaches
89.6 TFLOP/s at 4096^3 and 83 on the QKV shape of C2, the 128 x 128 fast form (forced, experiment build) 76.1 / 66.6 - hence the
fast form serves the split-K weight gradients only (gemm_f32.hip).  A v_mfma_f32_32x32x2_f32 build of the fast form (experiment, not
kept) measured 78.2 / 69.8 on the same two shapes and 55 - 57 TFLOP/s on the split weight gradients where the 16x16x4 form does 57 - 69."""
import sys, torch
sys.path.insert(0, ".")
import avformer_amd as A
from tools.bench_mx8 import timeit
ops = A.ops
for (M, N, K) in [(4096, 4096, 4096), (10368, 1536, 512), (10368, 512, 1536), (10368, 512, 512), (16384, 1024, 512)]:
    a = torch.randn(M, K, device="cuda")
    b = torch.randn(N, K, device="cuda") / K ** 0.5
    t = timeit(lambda: ops.gemm(a, b))
    print(f"NT {M}x{N}x{K}: {t:8.1f} us  {2.0 * M * N * K / t / 1e6:6.1f} TFLOP/s", flush=True)
for (M, N, K) in [(1536, 512, 10368), (1024, 512, 10368), (512, 1024, 10368), (512, 512, 10368)]:  # weight gradients (TN, split-K)
    a = torch.randn(K, M, device="cuda")
    b = torch.randn(K, N, device="cuda")
    t = timeit(lambda: ops.gemm(a, b, trans_a=True, trans_b=False))
    print(f"TN {M}x{N}x{K}: {t:8.1f} us  {2.0 * M * N * K / t / 1e6:6.1f} TFLOP/s", flush=True)
