"""merged attention backward with and without the MX-FP8 image of dqkv (us per launch, back to back)."""
import math
import sys
import torch
sys.path.insert(0, ".")
import avformer_amd as A  # noqa: E402

ops = A.ops
for B, N, H in ((64, 512, 8), (32, 512, 8), (32, 324, 8)):
    qkv = torch.randn(B * N, 3 * H * 64, device="cuda")
    qkv[:, :H * 64] *= math.log2(math.e) / 8.0
    qkv = qkv.bfloat16()
    d_o = torch.randn(B * N, H * 64, device="cuda").bfloat16()
    o, lse2 = ops.attn_fwd(qkv, B, N, H, 64, q_prescaled=True)[:2]

    def timed(fn, n=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) * 1e3 / n

    t0 = timed(lambda: ops.attn_bwd(qkv, o, d_o, lse2, B, N, H, 64, q_prescaled=True))
    t1 = timed(lambda: ops.attn_bwd_mx8(qkv, o, d_o, lse2, B, N, H, 64))
    tq = timed(lambda: ops.quant_mx8(qkv))
    print(f"B={B} N={N} H={H}: plain {t0:.1f} us, with image {t1:.1f} us; separate quantiser pass over a [B N, 3 I] tensor {tq:.1f} us", flush=True)
