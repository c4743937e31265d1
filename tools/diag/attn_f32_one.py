"""one parity-mode attention shape (forward + backward) a few times, for rocprofv3 counter passes: attn_f32_one.py B N H"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import avformer_amd as A
ops = A.ops
B, N, H = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dh = 64
qkv = torch.randn(B * N, 3 * H * dh, device="cuda")
d_o = torch.randn(B * N, H * dh, device="cuda")
for _ in range(4):
    o, lse2 = ops.attn_fwd(qkv, B, N, H, dh)
    ops.attn_bwd(qkv, o, d_o, lse2, B, N, H, dh)
torch.cuda.synchronize()
