"""-m gpu: heads == 1 and dim_head == dim, where the reference's Attention.to_out is nn.Identity (models/heads.py:207,214-217;
never instantiated by the reference's models, but a legal constructor call).  Forward / backward against fixture G14 run in
test_gpu_transformer.py; here: the state_dict carries the reference's keys only, no dropout site after the attention output,
the library optimizer and a checkpoint round trip."""
import pytest
import torch

import oracle
from conftest import load_golden, split_golden
from gpu_util import DEV, make_hip_transformer

pytestmark = pytest.mark.gpu


def test_state_dict_has_the_reference_keys_only():
    import avformer_amd as A
    p, _, r = split_golden(load_golden("g14_transformer_identity_out"))
    t = A.Transformer(r["dim"], r["depth"], r["heads"], r["dim_head"], r["mlp_dim"])
    assert set(t.state_dict().keys()) == set(p.keys())
    assert not any("to_out" in k for k in t.state_dict())
    assert isinstance(t.layers[0][0].fn.fn.to_out, torch.nn.Identity)
    t = t.to(DEV)
    assert t._identity_w.is_cuda and not t._identity_w.requires_grad


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_dropout_has_no_site_after_identity_to_out(mode):
    """replay of the kernels' masks through the oracle with NO factor at site 0 (nn.Identity has no Dropout)"""
    import avformer_amd as A
    B, N, D, L, M, p = 3, 20, 32, 2, 64, 0.3
    g = torch.Generator().manual_seed(8)
    sd = oracle.init_transformer_state(D, L, 1, D, M, generator=g)
    sd = {k: v for k, v in sd.items() if "to_out" not in k}
    x = torch.randn(B, N, D, generator=g)
    t = A.Transformer(D, L, 1, D, M, dropout=p, compute_dtype=mode)
    t.load_state_dict(sd)
    t = t.to(DEV).train()
    xg = x.to(DEV).requires_grad_(True)
    y = t(xg)
    y.pow(2).mean().backward()
    seed = t.last_seed
    drop = [(None,) + tuple(A.ops.dropout_factors(seed, l, s, p, B * N, cols).cpu().view(B, N, cols)
                            for s, cols in ((1, M), (2, D))) for l in range(L)]
    xr = x.clone().requires_grad_(True)
    pr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    yr = oracle.transformer_forward(xr, pr, L, 1, drop=drop)
    yr.pow(2).mean().backward()
    tol = dict(atol=5e-5, rtol=1e-3) if mode == "f32" else dict(atol=6e-2, rtol=6e-2)
    torch.testing.assert_close(y.cpu(), yr.detach(), **tol)
    gt = dict(atol=2e-6, rtol=2e-3) if mode == "f32" else dict(atol=2e-3, rtol=1e-1)
    torch.testing.assert_close(xg.grad.cpu(), xr.grad, **gt)
    for k, prm in t.named_parameters():
        torch.testing.assert_close(prm.grad.cpu(), pr[k].grad, **gt, msg=lambda m, k=k: f"{k}: {m}")


def test_library_adam_and_checkpoint_round_trip():
    import avformer_amd as A
    p, _, r = split_golden(load_golden("g14_transformer_identity_out"))
    t = make_hip_transformer(p, r["dim"], r["depth"], r["heads"], r["dim_head"], r["mlp_dim"], "bf16")
    ref = make_hip_transformer(p, r["dim"], r["depth"], r["heads"], r["dim_head"], r["mlp_dim"], "bf16")
    opt = A.optim.FusedAdam(t, lr=1e-2)
    opt_ref = torch.optim.Adam(ref.parameters(), lr=1e-2)
    x = r["x"].to(DEV)
    for _ in range(3):
        for m, o in ((t, opt), (ref, opt_ref)):
            o.zero_grad(set_to_none=True)
            m(x).pow(2).mean().backward()
            o.step()
    for (k, a), (_, b) in zip(t.named_parameters(), ref.named_parameters()):
        # (Adam turns a last-bit difference of a near-zero gradient element into a visible fraction of one lr = 1e-2 step)
        torch.testing.assert_close(a, b, atol=5e-4, rtol=2e-3, msg=lambda m, k=k: f"{k}: {m}")
    assert torch.equal(t._identity_w, torch.eye(r["dim"], device=DEV))  # the frozen identity is never stepped
    sd = opt.state_dict()  # only real parameters carry optimizer state
    assert len(sd["state"]) == len(list(t.parameters()))
    t2 = make_hip_transformer(t.state_dict(), r["dim"], r["depth"], r["heads"], r["dim_head"], r["mlp_dim"], "bf16")
    torch.testing.assert_close(t2(x), t(x), atol=0, rtol=0)
