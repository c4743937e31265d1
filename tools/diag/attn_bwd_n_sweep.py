"""attention backward (pre-scaled q, dim_head 64) over the token count, back to back (us per launch); run once with
AVF_ATTN_MERGED_MIN_N=1 (merged kernel everywhere) and once with =100000 (two head-resident kernels)."""
import math
import sys
import torch
sys.path.insert(0, ".")
import avformer_amd as A  # noqa: E402

ops = A.ops
out = []
for N in (17, 33, 49, 64, 100, 128, 196, 256):
    B, H = (64 if N <= 64 else 32), 8
    qkv = torch.randn(B * N, 3 * H * 64, device="cuda")
    qkv[:, :H * 64] *= math.log2(math.e) / 8.0
    qkv = qkv.bfloat16()
    d_o = torch.randn(B * N, H * 64, device="cuda").bfloat16()
    o, lse2 = ops.attn_fwd(qkv, B, N, H, 64, q_prescaled=True)[:2]
    f = lambda: ops.attn_bwd(qkv, o, d_o, lse2, B, N, H, 64, q_prescaled=True)
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        f()
    b.record()
    torch.cuda.synchronize()
    out.append(f"N={N}(B={B}): {a.elapsed_time(b) * 1e3 / 50:.1f}")
print("  ".join(out), flush=True)
