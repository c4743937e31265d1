"""Step time (forward + backward) of a bf16 stack at the C2 shape WITH a token mask: MFMA kernels (default) against the
fp32-arithmetic attention core (AVF_ATTN_MASK_MFMA=0)."""
import sys
import time
import torch
sys.path.insert(0, ".")
import avformer_amd as A  # noqa: E402

torch.manual_seed(0)
B, N, D = 32, 324, 512
t = A.Transformer(D, 6, 8, 64, 1024, compute_dtype="bf16", residual_dtype="bf16").cuda()
x = torch.randn(B, N, D, device="cuda", requires_grad=True)
mask = (torch.rand(B, N - 1, device="cuda") > 0.2)


def step(m):
    for p in t.parameters():
        p.grad = None
    t(x, mask=m).float().pow(2).mean().backward()


for name, m in (("masked", mask), ("no mask", None)):
    for _ in range(5):
        step(m)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        step(m)
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per fwd+bwd", flush=True)
