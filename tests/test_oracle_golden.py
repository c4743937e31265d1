"""CPU: the oracle (oracle/reference_math.py) against fixtures captured from the reference itself.

Tolerance: fp32 eager vs fp32 eager of the same op sequence -> 2e-5 abs / 1e-4 rel
(fp32-vs-fp64 noise of this block is ~2e-7 relative)."""
import pytest
import torch

import oracle
from conftest import load_golden, split_golden

ATOL, RTOL = 2e-5, 1e-4


def close(a, b, atol=ATOL, rtol=RTOL):
    if not torch.is_tensor(b):
        b = torch.tensor(b, dtype=a.dtype)
    torch.testing.assert_close(a.detach(), b, atol=atol, rtol=rtol)


def run_with_grads(fn, x, params):
    x = x.clone().requires_grad_(True)
    ps = {k: (v.clone().requires_grad_(True)
              if torch.is_tensor(v) and v.is_floating_point() and "running" not in k else v)
          for k, v in params.items()}
    y = fn(x, ps)
    return x, ps, y


def check_grads(ps, grads):
    for k, g in grads.items():
        assert ps[k].grad is not None, k
        close(ps[k].grad, g)


def test_g1_attention():
    p, g, r = split_golden(load_golden("g1_attention"))
    x, ps, y = run_with_grads(lambda x, ps: oracle.attention_forward(
        x, ps["to_qkv.weight"], ps["to_out.0.weight"], ps["to_out.0.bias"], r["heads"]), r["x"], p)
    close(y, r["y"])
    y.pow(2).mean().backward()
    close(x.grad, r["dx"])
    check_grads(ps, g)


def test_g2_feedforward():
    p, g, r = split_golden(load_golden("g2_feedforward"))
    x, ps, y = run_with_grads(lambda x, ps: oracle.feedforward_forward(
        x, ps["net.0.weight"], ps["net.0.bias"], ps["net.3.weight"], ps["net.3.bias"]), r["x"], p)
    close(y, r["y"])
    y.pow(2).mean().backward()
    close(x.grad, r["dx"])
    check_grads(ps, g)


@pytest.mark.parametrize("name", ["g3_transformer_c1", "g4_transformer_n12", "g4_transformer_n17",
                                  "g4_transformer_n49", "g14_transformer_identity_out"])
def test_transformer(name):
    p, g, r = split_golden(load_golden(name))
    x, ps, y = run_with_grads(lambda x, ps: oracle.transformer_forward(x, ps, r["depth"], r["heads"]), r["x"], p)
    close(y, r["y"])
    y.pow(2).mean().backward()
    close(x.grad, r["dx"])
    check_grads(ps, g)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_g13_transformer_with_token_mask(tag):
    """the mask branch of Attention (heads.py:225-232): clip 1 drops three tokens, clip 2 all but the first"""
    p, g, r = split_golden(load_golden(f"g13_transformer_mask_{tag}"))
    mask = r["mask"].bool()
    x, ps, y = run_with_grads(lambda x, ps: oracle.transformer_forward(x, ps, r["depth"], r["heads"], mask=mask), r["x"], p)
    close(y, r["y"])
    y.pow(2).mean().backward()
    close(x.grad, r["dx"])
    check_grads(ps, g)


def test_param_names_match_reference_schema():
    p, _, r = split_golden(load_golden("g3_transformer_c1"))
    names = []
    for i in range(r["depth"]):
        names += oracle.transformer_param_names(i)
    assert names == list(p.keys())
    sd = oracle.init_transformer_state(r["dim"], r["depth"], r["heads"], r["dim_head"], r["mlp_dim"])
    assert {k: tuple(v.shape) for k, v in sd.items()} == {k: tuple(v.shape) for k, v in p.items()}


def test_g5_au_former():
    p, g, r = split_golden(load_golden("g5_au_former"))
    x, ps, (logits, tokens) = run_with_grads(lambda x, ps: oracle.au_former_forward(x, ps), r["x"], p)
    close(logits, r["y"])
    close(tokens, r["y_extra0"])
    logits.pow(2).mean().backward()
    close(x.grad, r["dx"])
    check_grads(ps, g)


@pytest.mark.parametrize("tag", ["eval", "train"])
def test_g15_va_former(tag):
    p, g, r = split_golden(load_golden(f"g15_va_former_{tag}"))
    train = bool(int(r["training"]))
    x, ps, (va, tokens) = run_with_grads(lambda x, ps: oracle.va_former_forward(x, ps, training=train), r["x"], p)
    close(va, r["y"])
    close(tokens, r["y_extra0"])
    va.pow(2).mean().backward()
    close(x.grad, r["dx"])
    check_grads(ps, g)


def test_g6_au_head():
    p, g, r = split_golden(load_golden("g6_au_head"))
    x, ps, y = run_with_grads(lambda x, ps: oracle.au_head_forward(x, ps), r["x"], p)
    close(y, r["y"])
    y.pow(2).mean().backward()
    close(x.grad, r["dx"])
    check_grads(ps, g)


def test_g7_tformer():
    p, g, r = split_golden(load_golden("g7_tformer"))
    x, ps, y = run_with_grads(lambda x, ps: oracle.tformer_forward(
        x, ps, r["num_patches"], r["dim"], r["depth"], r["heads"]), r["x"], p)
    close(y, r["y"])
    y.pow(2).mean().backward()
    close(x.grad, r["dx"])
    check_grads(ps, g)


def test_g11_resformer_tokens():
    p, g, r = split_golden(load_golden("g11_resformer_tokens"))
    x, ps, y = run_with_grads(lambda x, ps: oracle.resformer_tokens_forward(x, ps, r["depth"], r["heads"]), r["x"], p)
    close(y, r["y"])
    y.pow(2).mean().backward()
    close(x.grad, r["dx"])
    check_grads(ps, g)


@pytest.mark.parametrize("tag", ["all", "ign"])
def test_g8_au_loss(tag):
    g = load_golden("g8_au_loss")
    z = g[f"{tag}.z"].clone().requires_grad_(True)
    loss = oracle.au_loss(z, g[f"{tag}.y"])
    close(loss, g[f"{tag}.loss"])
    loss.backward()
    close(z.grad, g[f"{tag}.dz"])
    if tag == "ign":
        assert torch.all(z.grad[[3, 7, 8]] == 0)  # dropped rows get exactly zero gradient


def test_au_loss_all_rows_ignored_is_nan():
    z = torch.zeros(3, 12)
    y = -torch.ones(3, 12)
    assert torch.isnan(oracle.au_loss(z, y))  # mean of empty, as in the reference (loss.py:102)


def test_g9_pipeline():
    g = load_golden("g9_pipeline")
    tf = {k[5:]: v for k, v in g.items() if k.startswith("p.tf.")}
    au = {k[5:]: v for k, v in g.items() if k.startswith("p.au.")}
    tf = {k: v.clone().requires_grad_(True) for k, v in tf.items()}
    au = {k: (v.clone().requires_grad_(True)
              if (torch.is_tensor(v) and "running" not in k and v.is_floating_point()) else v)
          for k, v in au.items()}
    x = g["x"].clone().requires_grad_(True)
    feat = oracle.tformer_forward(x, tf, 8, 32, 1, 8)
    logits, tokens = oracle.au_former_forward(feat, au)
    loss = oracle.au_loss(logits, g["labels"])
    close(feat, g["feat"])
    close(logits, g["logits"])
    close(tokens, g["tokens"])
    close(loss, g["loss"])
    loss.backward()
    close(x.grad, g["dx"])
    for k, v in g.items():
        if k.startswith("g.tf."):
            close(tf[k[5:]].grad, v)
        if k.startswith("g.au."):
            close(au[k[5:]].grad, v)


def test_g10_gelu():
    g = load_golden("g10_gelu")
    u = g["u"].clone().requires_grad_(True)
    y = oracle.gelu_tanh(u)
    close(y, g["y"], atol=1e-6, rtol=1e-6)
    y.sum().backward()
    close(u.grad, g["dy_du"], atol=1e-6, rtol=1e-5)


def test_bf16x3_arithmetic_meets_its_error_contract_on_the_cpu():
    """the arithmetic the parity mode's default kernels implement (oracle/bf16x3.py), emulated on the CPU: against float64 every
    element stays within 3 * 2^-16 * sum_k |a_k| |b_k| - also on operands spanning eight decades - and the typical relative
    Frobenius error is the 4e-6 the GPU kernels measure (tests/test_gpu_ops.py holds the kernels to this emulation)"""
    g = torch.Generator().manual_seed(3)
    for wide in (False, True):
        a = torch.randn(96, 512, generator=g)
        b = torch.randn(512, 80, generator=g)
        if wide:
            a = a * torch.pow(10.0, torch.rand(a.shape, generator=g) * 8 - 4)
            b = b * torch.pow(10.0, torch.rand(b.shape, generator=g) * 8 - 4)
        ref = a.double() @ b.double()
        got = oracle.bf16x3.matmul(a, b)
        mag = a.double().abs() @ b.double().abs()
        assert float(((got - ref).abs() / (oracle.bf16x3.PER_PRODUCT_BOUND * mag)).max()) <= 1.0
        rel = float((got - ref).norm() / ref.norm())
        assert rel < 8e-6, rel
        if not wide:
            assert rel > 1e-6, rel   # (three products, not an exact fp32 product: the emulation emulates)
