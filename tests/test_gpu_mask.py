"""-m gpu: Transformer.forward(x, mask) - the token mask of Attention (reference models/heads.py:225-232).

No caller of the reference passes a mask, but it is part of forward()'s signature.  Fixtures G13 come from the reference's own
Transformer run with a mask (tests/golden/make_golden.py --only g13): clip 0 keeps everything, clip 1 drops three tokens, clip
2 all but the first (a dropped QUERY attends uniformly to all keys - every score is the fill value - and passes no gradient to
q / k, only to v)."""
import pytest
import torch

import oracle
from conftest import load_golden, split_golden
from gpu_util import DEV, check_rel, make_hip_transformer

pytestmark = pytest.mark.gpu

SQ = lambda y: y.pow(2).mean()


def _close(a, b, atol=5e-5, rtol=1e-3):
    if not torch.is_tensor(b):
        b = torch.tensor(b)
    torch.testing.assert_close(a.detach().float().cpu(), b.detach().float().cpu(), atol=atol, rtol=rtol)


def _run(t, x, mask):
    x = x.detach().to(DEV).clone().requires_grad_(True)
    for p in t.parameters():
        p.grad = None
    y = t(x, mask=mask.to(DEV))
    SQ(y).backward()
    torch.cuda.synchronize()
    return y.detach(), x.grad, {k: p.grad for k, p in t.named_parameters()}


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_masked_transformer_f32_vs_reference_golden(tag):
    p, g, r = split_golden(load_golden(f"g13_transformer_mask_{tag}"))
    t = make_hip_transformer(p, r["dim"], r["depth"], r["heads"], r["dim_head"], r["mlp_dim"], "f32")
    y, dx, grads = _run(t, r["x"], r["mask"].bool())
    _close(y, r["y"])
    _close(dx, r["dx"], atol=1e-6)
    for k, v in g.items():
        _close(grads[k], v, atol=2e-6, rtol=2e-3)
    # and the mask matters: without it the output differs
    y_plain = t(r["x"].to(DEV))
    assert (y_plain - y).abs().max() > 1e-3


@pytest.mark.parametrize("tag", ["b", "c"])
def test_masked_transformer_bf16_vs_reference_golden(tag):
    p, g, r = split_golden(load_golden(f"g13_transformer_mask_{tag}"))
    t = make_hip_transformer(p, r["dim"], r["depth"], r["heads"], r["dim_head"], r["mlp_dim"], "bf16")
    y, dx, grads = _run(t, r["x"], r["mask"].bool())
    check_rel(f"mask_bf16[{tag}]:y", y, r["y"], 1.5e-2)
    check_rel(f"mask_bf16[{tag}]:dx", dx, r["dx"], 3e-2)
    for k, v in g.items():
        check_rel(f"mask_bf16[{tag}]:g.{k}", grads[k], v, 4e-2)


@pytest.mark.parametrize("B,N,H,dh", [(3, 40, 2, 64), (2, 100, 4, 32), (4, 7, 3, 16)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_masked_attention_core_vs_fp64(B, N, H, dh, dtype):
    import avformer_amd as A
    if dtype == torch.bfloat16 and dh % 8:
        pytest.skip("bf16 storage: dim_head % 8")
    g = torch.Generator().manual_seed(B * 100 + N)
    I = H * dh
    qkv = torch.randn(B * N, 3 * I, generator=g).to(dtype)
    d_o = torch.randn(B * N, I, generator=g).to(dtype)
    keep = torch.rand(B, N, generator=g) > 0.3
    keep[:, 0] = True
    keep[-1, 1:] = False  # one clip with every other token dropped
    o, lse2 = A.ops.attn_fwd_masked(qkv.to(DEV), keep.to(DEV), B, N, H, dh)
    dqkv = A.ops.attn_bwd_masked(qkv.to(DEV), o, d_o.to(DEV), lse2, keep.to(DEV), B, N, H, dh)
    # fp64 restatement of heads.py:222-237 on the same (storage-rounded) inputs
    x = qkv.double().requires_grad_(True)
    q, k, v = [t.reshape(B, N, H, dh).permute(0, 2, 1, 3) for t in x.split(I, dim=-1)]
    s = (q @ k.transpose(-1, -2)) * dh ** -0.5
    pair = keep[:, None, :, None] & keep[:, None, None, :]
    s = s.masked_fill(~pair, -torch.finfo(torch.float32).max)
    ref = (s.softmax(-1) @ v).permute(0, 2, 1, 3).reshape(B * N, I)
    ref.backward(d_o.double())
    tol = 1e-5 if dtype == torch.float32 else 1.2e-2
    check_rel(f"mask_attn[{B}x{N}x{H}x{dh},{dtype}]:o", o, ref.detach().float(), tol)
    check_rel(f"mask_attn[{B}x{N}x{H}x{dh},{dtype}]:dqkv", dqkv, x.grad.float(), tol * 2)
    # a dropped query passes no gradient to q, and a dropped key receives none into k
    dq = dqkv.float().view(B, N, 3, H, dh)[:, :, 0]
    dk = dqkv.float().view(B, N, 3, H, dh)[:, :, 1]
    assert torch.all(dq[~keep.to(DEV)] == 0) and torch.all(dk[~keep.to(DEV)] == 0)


def test_mask_shape_is_checked():
    import avformer_amd as A
    t = A.Transformer(64, 1, 2, 32, 128, compute_dtype="f32").to(DEV)
    x = torch.randn(2, 9, 64, device=DEV)
    with pytest.raises(AssertionError, match="incorrect dimensions"):
        t(x, mask=torch.ones(2, 9, dtype=torch.bool, device=DEV))  # must have N - 1 entries
    y = t(x, mask=torch.ones(2, 8, dtype=torch.bool, device=DEV))
    torch.testing.assert_close(y, t(x), atol=1e-6, rtol=1e-6)  # an all-True mask is the unmasked result


def test_mask_in_the_mx8_mode_against_the_bf16_mode():
    """with a mask the fp8 mode keeps its MX-FP8 GEMMs and only the attention core moves to the fp32-arithmetic kernels
    (the out-projection falls back to bf16 operands: the masked kernel writes no image)"""
    import avformer_amd as A
    torch.manual_seed(4)
    a = A.Transformer(128, 2, 2, 64, 256, compute_dtype="bf16").to(DEV)
    b = A.Transformer(128, 2, 2, 64, 256, compute_dtype="mx8").to(DEV)
    b.load_state_dict(a.state_dict())
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, 30, 128, generator=g)
    mask = torch.rand(3, 29, generator=g) > 0.3
    outs = [_run(t, x, mask) for t in (a, b)]
    (y0, dx0, g0), (y1, dx1, g1) = outs
    check_rel("mask_mx8:y", y1, y0, 3e-2)
    check_rel("mask_mx8:dx", dx1, dx0, 8e-2)
    for k in g0:
        check_rel(f"mask_mx8:g.{k}", g1[k], g0[k], 1.2e-1)
    sd = {k: v.detach().cpu() for k, v in a.state_dict().items()}
    check_rel("mask_mx8:y_vs_oracle", y1, oracle.transformer_forward(x, sd, 2, 2, mask=mask), 4e-2)


@pytest.mark.parametrize("B,N,H", [(2, 40, 2), (3, 324, 2), (2, 512, 1), (2, 257, 3)])
def test_mask_on_the_mfma_kernels_against_the_oracle(B, N, H):
    """bf16 stacks with dim_head 64 and up to 512 tokens carry the mask on the head-resident forward and the merged backward
    kernel (round 3; before, every masked call ran the fp32-arithmetic attention core): output and every gradient against
    the CPU oracle's autograd, and against the same stack on the fp32-arithmetic core."""
    import avformer_amd as A
    torch.manual_seed(N)
    D, M = 64 * H, 128
    t = A.Transformer(D, 2, H, 64, M, compute_dtype="bf16").to(DEV)
    sd = {k: v.detach().cpu().clone() for k, v in t.state_dict().items()}
    g = torch.Generator().manual_seed(N + 1)
    x = torch.randn(B, N, D, generator=g)
    mask = torch.rand(B, N - 1, generator=g) > 0.3
    mask[-1, 1:] = False          # one clip keeps only the leading token and its first neighbour
    mask[0] = True                # one clip keeps everything
    y, dx, grads = _run(t, x, mask)
    xc = x.clone().requires_grad_(True)
    ps = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    yc = oracle.transformer_forward(xc, ps, 2, H, mask=mask)
    SQ(yc).backward()
    tag = f"mask_mfma[{B}x{N}x{H}]"
    check_rel(tag + ":y", y, yc.detach(), 1.5e-2)
    check_rel(tag + ":dx", dx, xc.grad, 3e-2)
    for k, p in ps.items():
        check_rel(tag + f":g.{k}", grads[k], p.grad, 4e-2)
    assert all(bool(torch.isfinite(v).all()) for v in grads.values())
