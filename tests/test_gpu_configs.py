"""-m gpu: BASELINE.json configs[2..4] (C3, C4, C5) through the HIP path, and the operator-level fixtures G1 / G2 / G10.

Three layers of evidence per configuration ("configs not exercised"):
  (i)   FULL size (the per-GPU batch of the config): size-independent properties - clips are independent (a batch equals
        its two halves stacked, forward bitwise; input gradients bitwise; parameter gradients add up over the halves),
        repeat calls are bitwise identical, everything is finite; and the first clips of the full batch equal the
        small-batch run bitwise, which (ii) holds to the oracle.
  (ii)  the SAME tokens / width / heads / depth with the batch reduced to 2 against the CPU oracle (forward and every
        gradient): parity mode f32 at rtol 1e-3 (north_star), throughput mode bf16 (mx8 for C5) at calibrated bounds
        (gpu_util.check: min(cap, 3 x measured)).
  (iii) the attention operator at the config's (tokens, heads) against an fp64 restatement.
"""
import math

import pytest
import torch

import oracle
from conftest import load_golden, split_golden
from gpu_util import (DEV, F32_ARITHS, check_abs, check_rel, f32_arithmetic, hip_transformer_run, make_hip_transformer,
                      oracle_transformer_run, rel_fro)

pytestmark = pytest.mark.gpu

SUMSQ = lambda y: y.pow(2).sum()  # a per-clip additive loss: gradients of a batch are sums over its clips
SQ = lambda y: y.pow(2).mean()

#        name: (B_full, N, D, L, H, dh, M, throughput mode)
FULL = {
    "c3": (32, 512, 512, 6, 8, 64, 1024, "bf16"),      # configs[2]: d=512, T=512 fused sequence, global B=256 / 8 GPUs
    "c4": (16, 1024, 768, 12, 12, 64, 1536, "bf16"),   # configs[3]: d=768, 12 layers, T=1024, B=16 per GPU
    "c5": (64, 512, 512, 6, 8, 64, 1024, "mx8"),       # configs[4]: d=512 fp8 MFMA path, B=64 per GPU
}


def _state(D, L, H, dh, M, seed):
    g = torch.Generator().manual_seed(seed)
    sd = oracle.init_transformer_state(D, L, H, dh, M, generator=g)
    for k in sd:  # non-trivial LayerNorm affine so the dgamma / dbeta paths carry signal
        if k.endswith("norm.weight"):
            sd[k] = 1 + 0.1 * torch.randn(D, generator=g)
        if k.endswith("norm.bias"):
            sd[k] = 0.1 * torch.randn(D, generator=g)
    return sd, g


@pytest.mark.parametrize("cfg", list(FULL))
def test_full_size_properties(cfg):
    B, N, D, L, H, dh, M, mode = FULL[cfg]
    sd, g = _state(D, L, H, dh, M, 700 + len(cfg) + B)
    x = torch.randn(B, N, D, generator=g)
    t = make_hip_transformer(sd, D, L, H, dh, M, mode)
    # forward properties (eval)
    t.eval()
    xd = x.to(DEV)
    with torch.no_grad():
        y = t(xd)
        y2 = t(xd)
        ya, yb = t(xd[:B // 2]), t(xd[B // 2:])
        y_small = t(xd[:2])
    assert torch.isfinite(y).all()
    assert torch.equal(y, y2), "repeat forward is not bitwise identical"
    assert torch.equal(y, torch.cat([ya, yb], 0)), "clips are not independent in forward"
    assert torch.equal(y[:2], y_small), "the first clips of the full batch differ from the B=2 run"
    del y2, ya, yb
    # training step properties: gradients of the batch = sums over its halves
    t.train()
    yf, dxf, gf = hip_transformer_run(t, x, SUMSQ)
    gf = {k: v.clone() for k, v in gf.items()}
    assert torch.equal(yf, y), "train-mode forward (dropout 0) differs from eval"
    assert torch.isfinite(dxf).all() and all(torch.isfinite(v).all() for v in gf.values())
    _, dxa, ga = hip_transformer_run(t, x[:B // 2], SUMSQ)
    ga = {k: v.clone() for k, v in ga.items()}
    _, dxb, gb = hip_transformer_run(t, x[B // 2:], SUMSQ)
    # the incoming gradient 2y is the same per clip, every backward op is per clip: bitwise
    assert torch.equal(dxf, torch.cat([dxa, dxb], 0)), "clips are not independent in backward"
    for k in gf:  # token-axis reductions regroup (split-K chunks move with the row count): fp32 summation order only
        check_rel(f"full[{cfg}]:additivity:g.{k}", gf[k], ga[k] + gb[k], 2e-3)
    # repeat backward bitwise (no atomics anywhere on the path)
    _, dxf2, gf2 = hip_transformer_run(t, x, SUMSQ)
    assert torch.equal(dxf, dxf2) and all(torch.equal(gf[k], gf2[k]) for k in gf), "backward is not repeatable bitwise"


@pytest.mark.parametrize("cfg", list(FULL))
def test_config_vs_oracle_small_batch(cfg):
    """same N, D, H, L as the config, B = 2: forward and all gradients against the CPU oracle's autograd"""
    _, N, D, L, H, dh, M, mode = FULL[cfg]
    B = 2
    sd, g = _state(D, L, H, dh, M, 900 + len(cfg) + N)
    x = torch.randn(B, N, D, generator=g)
    y_ref, dx_ref, g_ref = oracle_transformer_run(x, sd, L, H, SQ)
    # parity mode: north_star's rtol 1e-3, in both of its arithmetics (three bf16 products per fp32 product - the default - and
    # the f32-input MFMA); the same stated tolerances, each arithmetic held to 3 x its own calibrated measurement
    for arith in F32_ARITHS:
        with f32_arithmetic(arith):
            t32 = make_hip_transformer(sd, D, L, H, dh, M, "f32")
            y, dx, grads = hip_transformer_run(t32, x, SQ)
        torch.testing.assert_close(y.cpu(), y_ref, rtol=1e-3, atol=5e-5)
        torch.testing.assert_close(dx.cpu(), dx_ref, rtol=1e-3, atol=1e-6 if L <= 6 else 2e-6)
        tag = "cfg_f32" if arith == "f32" else "cfg_f32x3"
        for k, v in g_ref.items():
            check_rel(f"{tag}[{cfg}]:g.{k}", grads[k], v, 2e-3)
        del t32
    # throughput mode
    t = make_hip_transformer(sd, D, L, H, dh, M, mode)
    y, dx, grads = hip_transformer_run(t, x, SQ)
    caps = (1.5e-2, 3e-2, 4e-2) if mode == "bf16" else (5e-2, 1e-1, 1.5e-1)
    check_rel(f"cfg_{mode}[{cfg}]:y", y, y_ref, caps[0])
    check_rel(f"cfg_{mode}[{cfg}]:dx", dx, dx_ref, caps[1])
    for k, v in g_ref.items():
        check_rel(f"cfg_{mode}[{cfg}]:g.{k}", grads[k], v, caps[2])
    if mode == "mx8":  # the stated config-5 tolerance is against the bf16 mode on the same weights
        t16 = make_hip_transformer(sd, D, L, H, dh, M, "bf16")
        y16, dx16, g16 = hip_transformer_run(t16, x, SQ)
        # (round 1 stated 3 % / 6 % / 10 % on 2-layer stacks; the e4m3 operand error compounds over the 6 layers of the
        # full config - measured 3.3 % on y - so the full-depth statement is 5 % / 10 % / 15 %, and 3 x measured)
        check_rel(f"cfg_mx8_vs_bf16[{cfg}]:y", y, y16, 5e-2)
        check_rel(f"cfg_mx8_vs_bf16[{cfg}]:dx", dx, dx16, 1e-1)
        for k in g16:
            check_rel(f"cfg_mx8_vs_bf16[{cfg}]:g.{k}", grads[k], g16[k], 1.5e-1)


def test_c3_c4_model_logits_and_loss_vs_oracle():
    """the synthetic AV model (token fusion + pos-emb + stack + pooled AU logits + AULoss) at the C3 and C4 widths, B = 2:
    parity-mode logits at rtol 1e-3 and the loss, throughput-mode at calibrated bounds"""
    import avformer_amd as A
    for name, (Tv, Ta, D, L, H, dh, M) in {"c3": (384, 128, 512, 6, 8, 64, 1024), "c4": (768, 256, 768, 12, 12, 64, 1536)}.items():
        B = 2
        torch.manual_seed(123)
        m32 = A.SyntheticAVFormer(D, L, H, dh, M, Tv, Ta, compute_dtype="f32").to(DEV)
        m16 = A.SyntheticAVFormer(D, L, H, dh, M, Tv, Ta, compute_dtype="bf16").to(DEV)
        m16.load_state_dict(m32.state_dict())
        g = torch.Generator().manual_seed(125)
        clip = torch.randn(B, Tv, D, generator=g)
        aud = torch.randn(B, Ta, D, generator=g)
        labels = (torch.rand(B, 12, generator=g) > 0.5).float()
        sd = {k: v.detach().cpu() for k, v in m32.state_dict().items()}
        tok = torch.cat([clip, aud], 1) + sd["pos_embedding"]
        tsd = {k[len("transformer."):]: v for k, v in sd.items() if k.startswith("transformer.")}
        logits_ref = oracle.transformer_forward(tok, tsd, L, H).mean(1) @ sd["au_fc.weight"].t() + sd["au_fc.bias"]
        loss_ref = oracle.au_loss(logits_ref, labels)
        batch = {"clip": clip.to(DEV), "audio_features": aud.to(DEV)}
        with torch.no_grad():
            out32, out16 = m32(batch), m16(batch)
            l32, l16 = m32.get_au_loss(out32, labels.to(DEV)), m16.get_au_loss(out16, labels.to(DEV))
        torch.testing.assert_close(out32[:, :12].cpu(), logits_ref, rtol=1e-3, atol=1e-4)
        torch.testing.assert_close(l32.cpu(), loss_ref, rtol=1e-4, atol=1e-5)
        check_abs(f"model[{name}]:logits_maxabs", out16[:, :12], logits_ref, 2e-2)
        check_abs(f"model[{name}]:loss", l16, loss_ref, 5e-3, floor=3e-4)


# ---------------------------------------------------------------------------------------------- attention at C3 / C4
def _attn_ref64(qkv, B, N, H, dh, d_o):
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


@pytest.mark.parametrize("qs", [False, True], ids=["raw_q", "prescaled_q"])
@pytest.mark.parametrize("B,N,H,dh", [(2, 512, 8, 64), (2, 1024, 12, 64), (1, 1000, 12, 64)],
                         ids=["c3_heads", "c4_heads", "c4_ragged"])
def test_attention_bf16_at_config_shapes(B, N, H, dh, qs):
    """C3 (N=512, H=8: the multi-pass head-resident kernels) and C4 (N=1024, H=12: the streaming kernels) against the
    fp64 restatement of heads.py:222-237 - forward, lse and dq / dk / dv"""
    import avformer_amd as A
    ops = A.ops
    g = torch.Generator().manual_seed(N + H)
    qkv = torch.randn(B * N, 3 * H * dh, generator=g).to(torch.bfloat16)
    d_o = torch.randn(B * N, H * dh, generator=g).to(torch.bfloat16)
    ref_in = qkv.float()
    if qs:
        c = math.log2(math.e) / math.sqrt(dh)
        I = H * dh
        dev = qkv.float().clone()
        dev[:, :I] = (dev[:, :I] * c).to(torch.bfloat16).float()
        ref_in = dev.clone()
        ref_in[:, :I] = ref_in[:, :I] / c
        qkv = dev.to(torch.bfloat16)
    o_ref, lse_ref, dqkv_ref = _attn_ref64(ref_in, B, N, H, dh, d_o.float())
    o, lse2 = ops.attn_fwd(qkv.cuda(), B, N, H, dh, q_prescaled=qs)
    tag = f"attn_cfg[{B}x{N}x{H},qs{int(qs)}]"
    check_rel(tag + ":o", o, o_ref, 1e-2)
    torch.testing.assert_close(lse2.cpu(), lse_ref.float(), atol=2e-2, rtol=1e-3)
    dqkv = ops.attn_bwd(qkv.cuda(), o, d_o.cuda(), lse2, B, N, H, dh, q_prescaled=qs)
    I = H * dh
    for name, sl in (("dq", slice(0, I)), ("dk", slice(I, 2 * I)), ("dv", slice(2 * I, 3 * I))):
        check_rel(f"{tag}:{name}", dqkv[:, sl], dqkv_ref[:, sl], 2e-2)


def test_attention_f32_at_c4_shape():
    import avformer_amd as A
    B, N, H, dh = 1, 1024, 12, 64
    g = torch.Generator().manual_seed(77)
    qkv = torch.randn(B * N, 3 * H * dh, generator=g)
    d_o = torch.randn(B * N, H * dh, generator=g)
    o_ref, lse_ref, dqkv_ref = _attn_ref64(qkv, B, N, H, dh, d_o)
    o, lse2 = A.ops.attn_fwd(qkv.cuda(), B, N, H, dh)
    torch.testing.assert_close(o.cpu(), o_ref.float(), atol=2e-5, rtol=1e-4)
    torch.testing.assert_close(lse2.cpu(), lse_ref.float(), atol=1e-4, rtol=1e-5)
    dqkv = A.ops.attn_bwd(qkv.cuda(), o, d_o.cuda(), lse2, B, N, H, dh)
    torch.testing.assert_close(dqkv.cpu(), dqkv_ref.float(), atol=5e-5, rtol=1e-4)


# ---------------------------------------------------------------------------------------------- G1 / G2 / G10 -> HIP ops
def test_g1_attention_fixture_through_hip_ops():
    """G1 (the reference's Attention(dim=32, heads=4, dim_head=8) on x[2,7,32]: y, dx, dW for loss y.pow(2).mean()) fed to the
    operator entry points: to_qkv GEMM -> attention core -> to_out GEMM + bias, and their backward (parity mode)"""
    import avformer_amd as A
    ops = A.ops
    p, gr, r = split_golden(load_golden("g1_attention"))
    H, dh = int(r["heads"]), int(r["dim_head"])
    x = r["x"].reshape(-1, 32).to(DEV)
    B, N = r["x"].shape[:2]
    wqkv, wo, bo = p["to_qkv.weight"].to(DEV), p["to_out.0.weight"].to(DEV), p["to_out.0.bias"].to(DEV)
    qkv = ops.gemm(x, wqkv)                                            # heads.py:221
    o, lse2 = ops.attn_fwd(qkv, B, N, H, dh)                           # heads.py:222-237
    y = ops.gemm(o, wo, epilogue=ops.EPI_BIAS_RES, bias=bo, residual=torch.zeros_like(x))  # heads.py:238
    torch.testing.assert_close(y.cpu().view_as(r["y"]), r["y"], atol=2e-6, rtol=1e-4)
    dy = (2.0 / y.numel()) * y
    d_o = ops.gemm(dy, wo, trans_b=False)
    dqkv = ops.attn_bwd(qkv, o, d_o, lse2, B, N, H, dh)
    dx = ops.gemm(dqkv, wqkv, trans_b=False)
    torch.testing.assert_close(dx.cpu().view_as(r["dx"]), r["dx"], atol=1e-7, rtol=1e-3)
    torch.testing.assert_close(ops.gemm(dqkv, x, trans_a=True, trans_b=False).cpu(), gr["to_qkv.weight"], atol=1e-7, rtol=1e-3)
    torch.testing.assert_close(ops.gemm(dy, o, trans_a=True, trans_b=False).cpu(), gr["to_out.0.weight"], atol=1e-7, rtol=1e-3)
    torch.testing.assert_close(ops.colsum(dy).cpu(), gr["to_out.0.bias"], atol=1e-7, rtol=1e-3)


def test_g2_feedforward_fixture_through_hip_ops():
    """G2 (the reference's FeedForward(32, 64)): Linear+GELU epilogue, Linear+bias, and the dGELU / dW / db backward"""
    import avformer_amd as A
    ops = A.ops
    p, gr, r = split_golden(load_golden("g2_feedforward"))
    x = r["x"].reshape(-1, 32).to(DEV)
    w1, b1, w2, b2 = (p[k].to(DEV) for k in ("net.0.weight", "net.0.bias", "net.3.weight", "net.3.bias"))
    gact, u = ops.gemm(x, w1, epilogue=ops.EPI_BIAS_GELU, bias=b1)     # heads.py:191-193
    y = ops.gemm(gact, w2, epilogue=ops.EPI_BIAS_RES, bias=b2, residual=torch.zeros_like(x))  # heads.py:195
    torch.testing.assert_close(y.cpu().view_as(r["y"]), r["y"], atol=2e-6, rtol=1e-4)
    dy = (2.0 / y.numel()) * y
    du = ops.gemm(dy, w2, trans_b=False, epilogue=ops.EPI_DGELU, aux=u)
    dx = ops.gemm(du, w1, trans_b=False)
    torch.testing.assert_close(dx.cpu().view_as(r["dx"]), r["dx"], atol=1e-7, rtol=1e-3)
    torch.testing.assert_close(ops.gemm(du, x, trans_a=True, trans_b=False).cpu(), gr["net.0.weight"], atol=1e-7, rtol=1e-3)
    torch.testing.assert_close(ops.colsum(du).cpu(), gr["net.0.bias"], atol=1e-7, rtol=1e-3)
    torch.testing.assert_close(ops.gemm(dy, gact, trans_a=True, trans_b=False).cpu(), gr["net.3.weight"], atol=1e-7, rtol=1e-3)
    torch.testing.assert_close(ops.colsum(dy).cpu(), gr["net.3.bias"], atol=1e-7, rtol=1e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_g10_gelu_fixture_through_hip_epilogues(dtype):
    """G10 (the reference's 9-op tanh-GELU on a grid incl. +-large values, and its derivative): the fused GELU and dGELU
    epilogues evaluated on that grid.  The grid enters as column 0 of A (times a unit weight), so acc = u exactly."""
    import avformer_amd as A
    ops = A.ops
    g = load_golden("g10_gelu")
    u = g["u"]
    if dtype == torch.bfloat16:
        keep = u.to(torch.bfloat16).float() == u  # grid points a bf16 operand carries exactly
        assert keep.sum() >= 50
    else:
        keep = torch.ones_like(u, dtype=torch.bool)
    n = u.numel()
    a = torch.zeros(n, 8)
    a[:, 0] = u
    w = torch.zeros(8, 8)
    w[0, 0] = 1.0
    out, aux = ops.gemm(a.to(dtype).to(DEV), w.to(dtype).to(DEV), out_dtype=torch.float32, epilogue=ops.EPI_BIAS_GELU,
                        bias=torch.zeros(8, device=DEV))
    assert torch.equal(aux[:, 0].cpu()[keep], u[keep])
    # parity mode evaluates tanhf; the bf16 path's exp2/rcp form differs by ~1e-6 relative (common.hpp gelu_tanh_fast)
    tol = dict(atol=1e-6, rtol=2e-6) if dtype == torch.float32 else dict(atol=2e-6, rtol=2e-5)
    torch.testing.assert_close(out[:, 0].cpu()[keep], g["y"][keep], **tol)
    # derivative: C = (A W^T) * gelu'(aux) with acc = 1
    ones = torch.zeros(n, 8)
    ones[:, 0] = 1.0
    auxin = torch.zeros(n, 8)
    auxin[:, 0] = u
    d = ops.gemm(ones.to(dtype).to(DEV), w.to(dtype).to(DEV), out_dtype=torch.float32, epilogue=ops.EPI_DGELU,
                 aux=auxin.to(DEV))
    torch.testing.assert_close(d[:, 0].cpu()[keep], g["dy_du"][keep], **(dict(atol=2e-6, rtol=2e-6) if dtype == torch.float32
                                                                          else dict(atol=5e-6, rtol=5e-5)))


def test_g10_gelu_fixture_through_the_bf16x3_epilogues():
    """G10 through the fused GELU / dGELU epilogues of the parity mode's DEFAULT kernel (gemm_f32x3_kernel takes problems of at
    least 96 x 96: the 8-column problem of the test above runs the f32-MFMA kernel and its tanhf).  Here the grid enters a
    128-column problem; the operand split leaves acc = u (1 - 2^-16 at worst), and the epilogue evaluates the exp2 / rcp forms of
    common.hpp: agreement with the reference's 9-op GELU and its derivative at 5e-5 relative - the class of the arithmetic."""
    import avformer_amd as A
    from gpu_util import f32_arithmetic
    ops = A.ops
    g = load_golden("g10_gelu")
    u = g["u"]
    n = u.numel()
    rows = ((n + 127) // 128) * 128
    a = torch.zeros(rows, 32)
    a[:n, 0] = u
    w = torch.zeros(128, 32)
    w[0, 0] = 1.0
    with f32_arithmetic("bf16x3"):
        out, aux = ops.gemm(a.to(DEV), w.to(DEV), epilogue=ops.EPI_BIAS_GELU, bias=torch.zeros(128, device=DEV))
        torch.testing.assert_close(aux[:n, 0].cpu(), u, atol=1e-30, rtol=3e-5)
        torch.testing.assert_close(out[:n, 0].cpu(), g["y"], atol=5e-6, rtol=5e-5)
        assert float(out[:, 1:].abs().max()) == 0.0 and float(aux[:, 1:].abs().max()) == 0.0
        ones = torch.zeros(rows, 32)
        ones[:, 0] = 1.0
        auxin = torch.zeros(rows, 128)
        auxin[:n, 0] = u
        d = ops.gemm(ones.to(DEV), w.to(DEV), epilogue=ops.EPI_DGELU, aux=auxin.to(DEV))
        torch.testing.assert_close(d[:n, 0].cpu(), g["dy_du"], atol=5e-6, rtol=5e-5)
