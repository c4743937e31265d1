"""-m gpu: the bf16 residual stream (``Transformer(..., residual_dtype="bf16")``, cfg.resid_bf16) - the forward stream
x -> x + attn(LN x) -> x + mlp(LN x) stored in bf16 between the kernels - against the CPU oracle and against the fp32-stream
mode of the same build.

Stated tolerance (relative Frobenius error against the fp32 oracle; caps, each case is also held to 3 x its measured value):
outputs 3e-2, input gradient 5e-2, parameter gradients 6e-2 - one extra bf16 rounding (2^-9 relative) per residual add, two
per layer, on top of the throughput mode's bf16 operands."""
import pytest
import torch

import oracle
from gpu_util import DEV, check_abs, check_rel, hip_transformer_run, oracle_transformer_run, rel_fro

pytestmark = pytest.mark.gpu

SQ = lambda y: y.pow(2).mean()


def _make(sd, D, L, H, dh, M, mode="bf16", resid="bf16", dropout=0.0):
    import avformer_amd as A
    t = A.Transformer(D, L, H, dh, M, dropout, compute_dtype=mode, residual_dtype=resid)
    t.load_state_dict(sd, strict=True)
    return t.to(DEV)


CONFIGS = {
    "c1": (4, 64, 128, 2, 8, 32, 256),
    "odd_tokens": (3, 77, 256, 1, 8, 32, 512),
    "tformer_real": (5, 17, 512, 3, 8, 64, 1024),
    "au_head_real": (6, 12, 256, 3, 8, 32, 256),       # <= 16 tokens: the per-operator path (the single-launch layer is fp32-stream)
    "c2_small_batch": (4, 324, 512, 6, 8, 64, 1024),
    "c3_small_batch": (2, 512, 512, 6, 8, 64, 1024),
    "c4_model_short": (2, 200, 768, 2, 12, 64, 1536),
    "wide_tformer": (3, 17, 1536, 1, 8, 64, 1024),
    # >= 8192 rows that are no multiple of 32: the row8 LayerNorm backward's two-batch form (32 rows per workgroup) with a ragged
    # last workgroup, at one and at two 16-byte chunks per lane
    "ragged_8748_rows": (27, 324, 512, 1, 8, 64, 1024),
    "ragged_9009_rows_d1024": (9, 1001, 1024, 1, 8, 64, 1024),
}


@pytest.mark.parametrize("cfg", list(CONFIGS))
def test_resid16_vs_oracle(cfg):
    B, N, D, L, H, dh, M = CONFIGS[cfg]
    g = torch.Generator().manual_seed(321)
    sd = oracle.init_transformer_state(D, L, H, dh, M, generator=g)
    for k in sd:
        if k.endswith("norm.weight"):
            sd[k] = 1 + 0.1 * torch.randn(D, generator=g)
        if k.endswith("norm.bias"):
            sd[k] = 0.1 * torch.randn(D, generator=g)
    x = torch.randn(B, N, D, generator=g)
    y_ref, dx_ref, g_ref = oracle_transformer_run(x, sd, L, H, SQ)
    t = _make(sd, D, L, H, dh, M)
    y, dx, grads = hip_transformer_run(t, x, SQ)
    assert y.dtype == torch.float32 and dx.dtype == torch.float32
    check_rel(f"rs16[{cfg}]:y", y, y_ref, 3e-2)
    check_rel(f"rs16[{cfg}]:dx", dx, dx_ref, 5e-2)
    for k, v in g_ref.items():
        check_rel(f"rs16[{cfg}]:g.{k}", grads[k], v, 6e-2)


def test_resid16_full_c3_properties():
    """full C3 batch: clips independent (bitwise), repeat call bitwise, finite, first clips == small-batch run"""
    B, N, D, L, H, dh, M = 32, 512, 512, 6, 8, 64, 1024
    g = torch.Generator().manual_seed(5)
    sd = oracle.init_transformer_state(D, L, H, dh, M, generator=g)
    t = _make(sd, D, L, H, dh, M).eval()
    x = torch.randn(B, N, D, generator=g).to(DEV)
    with torch.no_grad():
        y, y2 = t(x), t(x)
        ya, yb, ys = t(x[:16]), t(x[16:]), t(x[:2])
    assert torch.isfinite(y).all() and torch.equal(y, y2)
    assert torch.equal(y, torch.cat([ya, yb], 0)) and torch.equal(y[:2], ys)
    t.train()
    _, dx1, g1 = hip_transformer_run(t, x, SQ)
    g1 = {k: v.clone() for k, v in g1.items()}
    _, dx2, g2 = hip_transformer_run(t, x, SQ)
    assert torch.equal(dx1, dx2) and all(torch.equal(g1[k], g2[k]) for k in g1)


@pytest.mark.parametrize("mode", ["bf16", "mx8"])
def test_resid16_model_path_logits_loss_and_gradients(mode):
    """SyntheticAVFormer on the bf16 stream (token build writes bf16, pooled mean reads bf16): logits / loss against the
    oracle, every parameter gradient (incl. pos_embedding through the fused token build) against the fp32-stream mode"""
    import avformer_amd as A
    B, Tv, Ta, D, L, H, dh, M = 4, 196, 128, 512, 2, 8, 64, 1024
    torch.manual_seed(11)
    m32 = A.SyntheticAVFormer(D, L, H, dh, M, Tv, Ta, compute_dtype=mode).to(DEV)
    m16 = A.SyntheticAVFormer(D, L, H, dh, M, Tv, Ta, compute_dtype=mode, residual_dtype="bf16").to(DEV)
    m16.load_state_dict(m32.state_dict())
    g = torch.Generator().manual_seed(12)
    clip, aud = torch.randn(B, Tv, D, generator=g), torch.randn(B, Ta, D, generator=g)
    labels = (torch.rand(B, 12, generator=g) > 0.5).float()
    labels[1] = -1
    sd = {k: v.detach().cpu() for k, v in m32.state_dict().items()}
    tok = torch.cat([clip, aud], 1) + sd["pos_embedding"]
    tsd = {k[len("transformer."):]: v for k, v in sd.items() if k.startswith("transformer.")}
    logits_ref = oracle.transformer_forward(tok, tsd, L, H).mean(1) @ sd["au_fc.weight"].t() + sd["au_fc.bias"]
    loss_ref = oracle.au_loss(logits_ref, labels)
    batch = {"clip": clip.to(DEV).requires_grad_(True), "audio_features": aud.to(DEV).requires_grad_(True)}
    outs = {}
    for name, m in (("f32stream", m32), ("bf16stream", m16)):
        for v in batch.values():
            v.grad = None
        m.zero_grad(set_to_none=True)
        out = m(batch)
        loss = m.get_au_loss(out, labels.to(DEV))
        loss.backward()
        outs[name] = (out.detach(), loss.detach(), {n: p.grad.clone() for n, p in m.named_parameters()},
                      batch["clip"].grad.clone(), batch["audio_features"].grad.clone())
    o16, l16, g16, dc16, da16 = outs["bf16stream"]
    o32, l32, g32, dc32, da32 = outs["f32stream"]
    cap = 4e-2 if mode == "bf16" else 8e-2
    check_abs(f"rs16_model[{mode}]:logits_maxabs", o16[:, :12], logits_ref, cap)
    check_abs(f"rs16_model[{mode}]:loss", l16, loss_ref, 1e-2, floor=5e-4)
    for n in g32:
        check_rel(f"rs16_model[{mode}]:g.{n}", g16[n], g32[n], 8e-2 if mode == "bf16" else 1.5e-1)
    check_rel(f"rs16_model[{mode}]:dclip", dc16, dc32, 8e-2 if mode == "bf16" else 1.5e-1)
    check_rel(f"rs16_model[{mode}]:daudio", da16, da32, 8e-2 if mode == "bf16" else 1.5e-1)


def test_resid16_with_dropout_runs_and_eval_matches():
    """dropout live (fp32 gradient stream, bf16 forward stream): trains without NaN; eval() equals the p = 0 result"""
    B, N, D, L, H, dh, M = 3, 77, 128, 2, 8, 32, 256
    g = torch.Generator().manual_seed(2)
    sd = oracle.init_transformer_state(D, L, H, dh, M, generator=g)
    x = torch.randn(B, N, D, generator=g)
    t = _make(sd, D, L, H, dh, M, dropout=0.25).train()
    y, dx, grads = hip_transformer_run(t, x, SQ)
    assert torch.isfinite(y).all() and torch.isfinite(dx).all() and all(torch.isfinite(v).all() for v in grads.values())
    t0 = _make(sd, D, L, H, dh, M).eval()
    with torch.no_grad():
        assert torch.equal(t.eval()(x.to(DEV)), t0(x.to(DEV)))


def test_resid16_rejects_parity_mode():
    import avformer_amd as A
    with pytest.raises(ValueError, match="residual_dtype"):
        A.Transformer(64, 1, 2, 32, 64, compute_dtype="f32", residual_dtype="bf16")
