"""CPU emulation of the parity mode's "bf16x3" arithmetic (test infrastructure, like the rest of ``oracle/``).

Not reference math - the reference computes its products in fp32.  This restates what the HIP kernels of
``csrc/gemm_f32.hip`` / ``attn_f32x3.hip`` do to an fp32 matrix product, so that (a) the error contract can be checked on the
CPU, against float64, independently of any kernel, and (b) a kernel can be held to THIS arithmetic (tests/test_gpu_ops.py) far
more tightly than to the exact product - a kernel that dropped a term, or split by truncation, agrees with the exact product to
1e-3 but not with this emulation to 1e-6:

    hi = bf16(x)  (round to nearest even),   lo = bf16(x - hi)  (x - hi is exact in fp32),
    a b  ~  hi_a hi_b + hi_a lo_b + lo_a hi_b        (every bf16 x bf16 product is exact in fp32; accumulation in fp32)

bf16 rounds to 8 significant bits (unit roundoff u = 2^-8): |x - hi| <= u |x|, |x - hi - lo| <= u^2 |x|.  Dropped: lo_a lo_b and the two
residuals, each <= u^2 |a b| = 2^-16 |a b|: <= 3 * 2^-16 = 4.6e-5 relative error per product in the worst case; on random data the
relative Frobenius error of a product sum is ~4e-6.
"""
import torch


def split(x: torch.Tensor):
    """fp32 tensor -> (hi, lo) as fp32 tensors holding bf16 values"""
    x = x.detach().float()
    hi = x.to(torch.bfloat16).float()
    lo = (x - hi).to(torch.bfloat16).float()
    return hi, lo


def matmul(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """a [M, K] @ b [K, N] in the bf16x3 arithmetic, the three partial products summed in float64 (the accumulation error of a
    kernel - fp32, some order - is what a comparison against this leaves)"""
    ah, al = split(a)
    bh, bl = split(b)
    d = torch.float64
    return ah.to(d) @ bh.to(d) + ah.to(d) @ bl.to(d) + al.to(d) @ bh.to(d)


PER_PRODUCT_BOUND = 3.0 * 2.0 ** -16
