"""parity-mode attention in both arithmetics (bf16x3 / f32 MFMA): time per launch at the C2 / C3 head shapes and error against fp64"""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import avformer_amd as A
from avformer_amd import _lib
from tools.bench_mx8 import timeit
ops = A.ops
for (B, N, H, dh) in [(32, 324, 8, 64), (32, 512, 8, 64), (2, 324, 2, 64)]:
    g = torch.Generator().manual_seed(N)
    qkv = torch.randn(B * N, 3 * H * dh, generator=g)
    d_o = torch.randn(B * N, H * dh, generator=g)
    ref = None
    if B <= 2:
        x = qkv.double().requires_grad_(True)
        q, k, v = [t.view(B, N, H, dh).transpose(1, 2) for t in x.chunk(3, -1)]
        p = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(dh), -1)
        oo = (p @ v).transpose(1, 2).reshape(B * N, H * dh)
        oo.backward(d_o.double())
        ref = (oo.detach(), x.grad)
    qc, dc = qkv.cuda(), d_o.cuda()
    for mode in ("f32", "bf16x3"):
        _lib.set_f32_arithmetic(mode)
        o, lse2 = ops.attn_fwd(qc, B, N, H, dh)
        dqkv = ops.attn_bwd(qc, o, dc, lse2, B, N, H, dh)
        tf = timeit(lambda: ops.attn_fwd(qc, B, N, H, dh), iters=20)
        tb = timeit(lambda: ops.attn_bwd(qc, o, dc, lse2, B, N, H, dh), iters=20)
        fl = 4.0 * B * H * N * N * dh
        line = f"B{B} N{N} H{H} {mode:7s}: fwd {tf:7.1f} us ({fl / tf / 1e6:6.1f} TF/s)  bwd {tb:7.1f} us ({2.5 * fl / tb / 1e6:6.1f} TF/s)"
        if ref is not None:
            eo = float((o.double().cpu() - ref[0]).norm() / ref[0].norm())
            eg = float((dqkv.double().cpu() - ref[1]).norm() / ref[1].norm())
            line += f"  relfro o {eo:.2e} dqkv {eg:.2e}"
        print(line, flush=True)
_lib.set_f32_arithmetic("bf16x3")
