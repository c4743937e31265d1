#!/usr/bin/env python3
"""Random eligible shapes of the single-launch layer forward (bf16, <= 16 tokens, dim_head 32): outputs and gradients of
the bf16 Transformer against the fp32 parity mode on the same weights.  usage: python tools/fuzz_small_layer.py [count] [seed]"""
import os, random, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import avformer_amd as A


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


count = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
for it in range(count):
    D, I, M = rng.choice([128, 256]), rng.choice([128, 256]), rng.choice([128, 256])
    H, N, B, L = I // 32, rng.randint(1, 16), rng.choice([1, 2, 7, 64, 300]), rng.randint(1, 3)
    torch.manual_seed(rng.randint(0, 1 << 30))
    t32 = A.Transformer(D, L, H, 32, M, compute_dtype="f32").cuda()
    t16 = A.Transformer(D, L, H, 32, M, compute_dtype="bf16").cuda()
    t16.load_state_dict(t32.state_dict())
    x = torch.randn(B, N, D, device="cuda")
    outs = []
    for t in (t32, t16):
        xi = x.clone().requires_grad_(True)
        y = t(xi)
        y.pow(2).mean().backward()
        outs.append((y.detach(), xi.grad, {n: p.grad for n, p in t.named_parameters()}))
    (y0, dx0, g0), (y1, dx1, g1) = outs
    ey, ed = rel(y1, y0), rel(dx1, dx0)
    eg = max(rel(g1[n], g0[n]) for n in g0)
    ok = ey < 1.5e-2 and ed < 3e-2 and eg < 5e-2
    print(f"{'ok ' if ok else 'BAD'} B={B} N={N} D={D} I={I} M={M} L={L}: y {ey:.2e} dx {ed:.2e} grads {eg:.2e}")
    if not ok:
        sys.exit(1)
print("all ok")
