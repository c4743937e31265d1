"""Helpers shared by the -m gpu tests (HIP path vs the CPU oracle)."""
import torch

import avformer_amd as A  # noqa: F401  (alias of the hyphenated package directory)
import oracle

DEV = "cuda"


def rel_fro(a, b):
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def max_abs(a, b):
    return (a.detach().float().cpu() - b.detach().float().cpu()).abs().max().item()


def make_hip_transformer(sd, dim, depth, heads, dim_head, mlp_dim, compute_dtype):
    t = A.Transformer(dim, depth, heads, dim_head, mlp_dim, 0.0, compute_dtype=compute_dtype)
    missing = t.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return t.to(DEV)


def oracle_transformer_run(x, sd, depth, heads, loss_fn):
    """CPU oracle forward + autograd backward -> y, dx, {param: grad}."""
    x = x.detach().cpu().clone().requires_grad_(True)
    ps = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in sd.items()}
    y = oracle.transformer_forward(x, ps, depth, heads)
    loss_fn(y).backward()
    return y.detach(), x.grad, {k: v.grad for k, v in ps.items()}


def hip_transformer_run(t, x, loss_fn):
    x = x.detach().to(DEV).clone().requires_grad_(True)
    for p in t.parameters():
        p.grad = None
    y = t(x)
    loss_fn(y).backward()
    torch.cuda.synchronize()
    return y.detach(), x.grad, {k: p.grad for k, p in t.named_parameters()}
