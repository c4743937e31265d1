"""LayerNorm folded into the GEMM behind it (cfg.ln_fuse): PreNorm (models/heads.py:178-185) followed by
to_qkv (heads.py:212) / net.0 (heads.py:191) computed as  rstd (x (gamma o W)^T - mean s) + c  on the raw bf16 residual stream.
The folded path must agree with the separate-LayerNorm path of the same mode (same storage types) far more tightly than
either agrees with the fp32 oracle, and stay inside the residual-stream tolerances against the oracle."""
import pytest
import torch

import avformer_amd as A
import oracle
from gpu_util import DEV, rel_fro, check

pytestmark = pytest.mark.gpu


def _pair(dim, depth, heads, dim_head, mlp, dropout=0.0):
    torch.manual_seed(11)
    a = A.Transformer(dim, depth, heads, dim_head, mlp, dropout, compute_dtype="bf16", residual_dtype="bf16").to(DEV)
    b = A.Transformer(dim, depth, heads, dim_head, mlp, dropout, compute_dtype="bf16", residual_dtype="bf16").to(DEV)
    b.load_state_dict(a.state_dict())
    assert a.ln_fuse_ok and b.ln_fuse_ok
    a.ln_fuse, b.ln_fuse = True, False
    # non-trivial LayerNorm parameters and biases (the default init has gamma = 1, beta = 0: the fold would not be exercised)
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for n, p in a.named_parameters():
            if "norm" in n or n.endswith("bias"):
                p.add_(0.3 * torch.randn(p.shape, generator=g).to(DEV))
    b.load_state_dict(a.state_dict())
    return a, b


@pytest.mark.parametrize("B,N,dim,depth,heads,dh,mlp", [(2, 324, 512, 2, 8, 64, 1024), (3, 50, 128, 3, 4, 32, 256),
                                                          (2, 64, 192, 2, 2, 32, 320), (4, 512, 512, 1, 8, 64, 1024)])
def test_folded_layernorm_matches_the_separate_kernels(B, N, dim, depth, heads, dh, mlp):
    a, b = _pair(dim, depth, heads, dh, mlp)
    g = torch.Generator().manual_seed(B * N)
    x = (torch.randn(B, N, dim, generator=g) * 1.5 + 0.7).to(DEV)   # rows with a mean: the mean * s cancellation is live
    outs = []
    for m in (a, b):
        xi = x.clone().requires_grad_(True)
        y = m(xi)
        (y * torch.linspace(-1, 1, dim, device=DEV)).sum().backward()
        outs.append((y.detach(), xi.grad, {k: p.grad.clone() for k, p in m.named_parameters()}))
    (ya, dxa, ga), (yb, dxb, gb) = outs
    tag = f"lnfuse[{B}x{N}x{dim}x{depth}]"
    check(tag + ":y", rel_fro(ya, yb), 8e-3)
    check(tag + ":dx", rel_fro(dxa, dxb), 1.5e-2)
    for k in ga:
        check(f"{tag}:{k}", rel_fro(ga[k], gb[k]), 2.5e-2)


def test_folded_layernorm_against_the_oracle():
    dim, depth, heads, dh, mlp, B, N = 256, 2, 8, 32, 512, 2, 96
    a, _ = _pair(dim, depth, heads, dh, mlp)
    sd = {k: v.detach().cpu() for k, v in a.state_dict().items()}
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, N, dim, generator=g) + 0.5
    xr = x.clone().requires_grad_(True)
    pr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    yr = oracle.transformer_forward(xr, pr, depth, heads)
    yr.pow(2).mean().backward()
    xg = x.to(DEV).requires_grad_(True)
    y = a(xg)
    y.pow(2).mean().backward()
    check("lnfuse_oracle:y", rel_fro(y, yr), 3e-2)
    check("lnfuse_oracle:dx", rel_fro(xg.grad, xr.grad), 5e-2)
    for k, p in a.named_parameters():
        check(f"lnfuse_oracle:{k}", rel_fro(p.grad, pr[k].grad), 6e-2)


def test_folded_layernorm_follows_weight_updates_and_masks():
    """the gamma-scaled images are re-derived after an optimizer step and after a forward that ran without them (mask)"""
    a, b = _pair(128, 2, 4, 32, 256)
    oa, ob = A.optim.FusedAdam(a, lr=1e-2), A.optim.FusedAdam(b, lr=1e-2)
    g = torch.Generator().manual_seed(8)
    x = torch.randn(2, 40, 128, generator=g).to(DEV)
    mask = torch.ones(2, 39, dtype=torch.bool, device=DEV)
    mask[0, 5:9] = False
    for step in range(3):
        for m, o in ((a, oa), (b, ob)):
            o.zero_grad(set_to_none=True)
            m(x).pow(2).mean().backward()
            o.step()
        if step == 1:  # a masked forward (separate LayerNorm kernels, plain images) in between
            b.load_state_dict(a.state_dict())  # (the two paths round differently: re-align the weights for the comparison)
            ya, yb = a(x, mask=mask), b(x, mask=mask)
            assert rel_fro(ya, yb) < 1e-6
    b.load_state_dict(a.state_dict())
    assert rel_fro(a(x), b(x)) < 8e-3   # the folded images follow the optimizer steps and survive the masked forward
