"""cfg.dw_overlap (weight-gradient launch on the library's side stream): same gradients, bit for bit, as the in-stream order."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _grads(overlap, dev):
    import avformer_amd as A
    torch.manual_seed(3)
    t = A.Transformer(256, 3, 4, 64, 512, 0.0, compute_dtype="bf16", residual_dtype="bf16").to(dev)
    t.dw_overlap = overlap
    x = torch.randn(4, 96, 256, device=dev, requires_grad=True)
    out = []
    for _ in range(2):  # second pass: the side stream and the doubled workspace are reused
        for p in t.parameters():
            p.grad = None
        x.grad = None
        t(x).float().pow(2).mean().backward()
        torch.cuda.synchronize()
        out.append([x.grad.clone()] + [p.grad.clone() for p in t.parameters()])
    return out


def test_dw_overlap_matches_in_stream_order():
    dev = torch.device("cuda:0")
    a, b = _grads(False, dev), _grads(True, dev)
    for ga, gb in zip(a, b):
        for u, v in zip(ga, gb):
            assert torch.equal(u, v)
