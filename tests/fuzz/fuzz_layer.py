#!/usr/bin/env python3
"""Random stack configurations (any dim / inner / mlp widths the modes accept, 1..700 tokens, with and without pooling):
fp32 parity mode against the CPU oracle's autograd (tight), bf16 and - where the widths allow - mx8 against the parity mode.
usage: python tests/fuzz/fuzz_layer.py [count] [seed]"""
import os, random, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import avformer_amd as A
import oracle


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def run(count=30, seed=0, verbose=True):
    """`count` random stack configurations from `seed`; raises AssertionError on the first configuration out of bounds.  The
    parity mode runs in both of its arithmetics (bf16x3 - the default - and the f32-input MFMA), at the same limits.
    (tests/test_gpu_fuzz.py runs a fixed budget of these under pytest -m gpu)"""
    from avformer_amd import _lib
    rng = random.Random(seed)
    for it in range(count):
        dh = rng.choice([32, 64])
        H = rng.choice([1, 2, 4, 8, 12])
        D = rng.choice([32, 64, 96, 128, 256, 384, 512])
        M = rng.choice([64, 128, 200, 256, 512, 1024])
        N = rng.choice([1, 2, 7, 12, 17, 31, 32, 33, 49, 64, 100, 196, 324, 400, 577, 640])
        B = rng.choice([1, 2, 3, 5])
        if H == 1 and dh == D:
            continue  # nn.Identity to_out: its own fixture (G14) and tests (test_gpu_identity.py)
        L = rng.randint(1, 2)
        pool = rng.random() < 0.3 and D % 4 == 0
        torch.manual_seed(rng.randint(0, 1 << 30))
        t32 = A.Transformer(D, L, H, dh, M, compute_dtype="f32").cuda()
        sd = t32.state_dict()
        x = torch.randn(B, N, D, device="cuda")
        # oracle (CPU autograd)
        xc = x.cpu().clone().requires_grad_(True)
        ps = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in sd.items()}
        yc = oracle.transformer_forward(xc, ps, L, H)
        if pool:
            yc = yc.mean(dim=1)
        yc.pow(2).mean().backward()
        modes = ["f32", "f32m", "bf16"] + (["bf16r"] if D % 8 == 0 else []) + (["mx8"] if D % 128 == 0 and M % 128 == 0 else [])
        line = []
        ok = True
        for mode in modes:
            # f32: the parity mode in its default arithmetic (bf16x3); f32m: the same on the f32-input MFMA
            # bf16r: bf16 compute on the bf16 residual stream (the benchmarked default since round 3: row8 LayerNorm kernels)
            prev = _lib.set_f32_arithmetic("f32" if mode == "f32m" else "bf16x3")
            try:
                t = t32 if mode in ("f32", "f32m") else A.Transformer(
                    D, L, H, dh, M, compute_dtype="bf16" if mode == "bf16r" else mode,
                    residual_dtype="bf16" if mode == "bf16r" else "f32").cuda()
                if mode not in ("f32", "f32m"):
                    t.load_state_dict(sd)
                t.zero_grad(set_to_none=True)
                xi = x.clone().requires_grad_(True)
                y = t(xi, pool="mean" if pool else None)
                y.pow(2).mean().backward()
            finally:
                _lib.set_f32_arithmetic(prev)
            ey, ed = rel(y, yc), rel(xi.grad, xc.grad)
            eg = max(rel(p.grad, ps[n].grad) for n, p in t.named_parameters())
            # (mx8: pooled outputs cancel signal, not e4m3 noise; backward operands are e4m3 too since round 2)
            lim = {"f32": (2e-5, 1e-4, 2e-4), "f32m": (2e-5, 1e-4, 2e-4), "bf16": (1.5e-2, 3e-2, 6e-2),
                   "bf16r": (2.5e-2, 4e-2, 8e-2), "mx8": (8e-2, 1e-1, 1.6e-1)}[mode]
            good = ey < lim[0] and ed < lim[1] and eg < lim[2]
            ok = ok and good
            line.append(f"{mode}{'' if good else '!'} y {ey:.1e} dx {ed:.1e} g {eg:.1e}")
        msg = f"{'ok ' if ok else 'BAD'} B={B} N={N} D={D} H={H} dh={dh} M={M} L={L} pool={int(pool)}: " + " | ".join(line)
        if verbose:
            print(msg)
        assert ok, msg + f"  (seed {seed}, configuration {it})"


if __name__ == "__main__":
    try:
        run(int(sys.argv[1]) if len(sys.argv) > 1 else 30, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    except AssertionError as e:
        print(e)
        sys.exit(1)
    print("all ok")
