"""A/B of the fused FeedForward kernels against the two-GEMM path, back to back under one event pair (us per call)."""
import sys
import torch
sys.path.insert(0, ".")
import avformer_amd as A  # noqa: E402

ops = A.ops
DEV = "cuda:0"


def timed(fn, n=50):
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


for R, D, M in ((10368, 512, 1024), (16384, 512, 1024), (8192, 768, 3072)):
    h = torch.randn(R, D, device=DEV).bfloat16()
    w1 = (torch.randn(M, D, device=DEV) * D ** -0.5).bfloat16()
    w2 = (torch.randn(D, M, device=DEV) * M ** -0.5).bfloat16()
    b1 = torch.randn(M, device=DEV) * 0.1
    b2 = torch.randn(D, device=DEV) * 0.1
    x_mid = torch.randn(R, D, device=DEV).bfloat16()
    u = torch.randn(R, M, device=DEV).bfloat16()
    w2_t, w1_t = w2.T.contiguous(), w1.T.contiguous()

    def two_fwd():
        g, _ = ops.gemm(h, w1, epilogue=ops.EPI_BIAS_GELU, bias=b1)
        return ops.gemm(g, w2, epilogue=ops.EPI_BIAS_RES, bias=b2, residual=x_mid)

    def two_bwd():
        du = ops.gemm(h, w2_t, epilogue=ops.EPI_DGELU, aux=u)
        return ops.gemm(du, w1_t)

    t2f, t1f = timed(two_fwd), timed(lambda: ops.mlp_fused_fwd(h, w1, b1, w2, b2, x_mid))
    t2b, t1b = timed(two_bwd), timed(lambda: ops.mlp_fused_bwd(h, w2_t, w1_t, u))
    fl = 4.0 * R * D * M
    print(f"R={R} D={D} M={M}: fwd two GEMMs {t2f:.1f} us, fused {t1f:.1f} us ({fl / t1f / 1e6:.0f} TFLOP/s); "
          f"bwd two GEMMs {t2b:.1f} us, fused {t1b:.1f} us ({fl / t1b / 1e6:.0f} TFLOP/s)", flush=True)
