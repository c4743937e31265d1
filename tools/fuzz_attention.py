#!/usr/bin/env python3
"""Random-shape check of the bf16 attention kernels (all families, raw and pre-scaled q) against an fp64 restatement.
usage: python tools/fuzz_attention.py [count] [seed]"""
import math, os, random, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import avformer_amd as A
ops = A.ops


def ref(qkv, B, N, H, dh, d_o):
    I = H * dh
    qkv = qkv.double().clone().requires_grad_(True)
    q, k, v = qkv.view(B, N, 3 * I).split(I, dim=-1)
    sh = lambda t: t.reshape(B, N, H, dh).permute(0, 2, 1, 3)
    q, k, v = sh(q), sh(k), sh(v)
    s = (q @ k.transpose(-1, -2)) * dh ** -0.5
    o = (s.softmax(-1) @ v).permute(0, 2, 1, 3).reshape(B * N, I)
    lse2 = torch.logsumexp(s, dim=-1) * math.log2(math.e)
    o.backward(d_o.double())
    return o.detach(), lse2.detach(), qkv.grad


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


count = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = 0.0
for it in range(count):
    dh = rng.choice([32, 64, 64, 64])
    N = rng.choice([rng.randint(1, 40), rng.randint(41, 400), rng.randint(380, 600), rng.randint(570, 700)])
    B, H = rng.randint(1, 3), rng.randint(1, 4)
    qs = rng.random() < 0.5
    scale = rng.choice([0.3, 1.0, 3.0])
    g = torch.Generator().manual_seed(rng.randint(0, 1 << 30))
    qkv = (torch.randn(B * N, 3 * H * dh, generator=g) * scale).to(torch.bfloat16)
    d_o = torch.randn(B * N, H * dh, generator=g).to(torch.bfloat16)
    ref_in = qkv.float()
    if qs:
        c = math.log2(math.e) / math.sqrt(dh)
        dev = qkv.float().clone()
        dev[:, :H * dh] = (dev[:, :H * dh] * c).to(torch.bfloat16).float()
        ref_in = dev.clone()
        ref_in[:, :H * dh] /= c
        qkv = dev.to(torch.bfloat16)
    o_ref, lse_ref, dq_ref = ref(ref_in, B, N, H, dh, d_o.float())
    o, lse2 = ops.attn_fwd(qkv.cuda(), B, N, H, dh, q_prescaled=qs)
    dqkv = ops.attn_bwd(qkv.cuda(), o, d_o.cuda(), lse2, B, N, H, dh, q_prescaled=qs)
    eo, el, ed = rel(o, o_ref), float((lse2.cpu().double() - lse_ref).abs().max()), rel(dqkv, dq_ref) if N > 1 else 0.0
    ok = eo < 1.2e-2 and el < 6e-2 and ed < 3e-2 and bool(torch.isfinite(dqkv.float()).all())
    worst = max(worst, eo, ed)
    print(f"{'ok ' if ok else 'BAD'} B={B} N={N} H={H} dh={dh} qs={int(qs)} scale={scale}: o {eo:.2e} lse {el:.2e} dqkv {ed:.2e}")
    if not ok:
        sys.exit(1)
print("all ok; worst relative error", worst)
