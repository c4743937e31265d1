"""head-resident attention forward, back to back (us per launch)."""
import math
import sys
import torch
sys.path.insert(0, ".")
import avformer_amd as A  # noqa: E402

ops = A.ops
for B, N, H in ((32, 324, 8), (32, 512, 8), (64, 512, 8)):
    qkv = torch.randn(B * N, 3 * H * 64, device="cuda")
    qkv[:, :H * 64] *= math.log2(math.e) / 8.0
    qkv = qkv.bfloat16()
    for _ in range(5):
        ops.attn_fwd(qkv, B, N, H, 64, q_prescaled=True)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(40):
        ops.attn_fwd(qkv, B, N, H, 64, q_prescaled=True)
    b.record()
    torch.cuda.synchronize()
    print(f"B={B} N={N} H={H}: forward {a.elapsed_time(b) * 1e3 / 40:.1f} us", flush=True)
