"""-m gpu: the MX-FP8 operand path (BASELINE config 5) - avf_quant_mx8 / avf_gemm_mx8_nt through the C ABI.

Checker: oracle/mx8.py, a CPU emulation of the published OCP MXFP8 (e4m3) conversion.
  * quantiser: bit patterns (element bytes and E8M0 scale bytes) must be IDENTICAL to the emulation.
  * GEMM: against the fp32 product of the DEQUANTISED operands.  The matrix core sums the 128 products of one
    instruction after aligning them to the largest (terms ~2^17 below it are truncated, tools/diag/mfma_fp8_probe3.hip),
    so the bound is elementwise  |c - ref| <= 2e-3 * (|A| |B|^T)  (observed <= 4e-4), not an fp32-rounding bound.
  * end to end against the UNQUANTISED fp32 product: relative Frobenius error <= 5 % on N(0,1) data (e4m3 keeps 3
    mantissa bits: ~3 % rms per element, sqrt(2) of that per product of two quantised operands; a sum of independent
    products keeps that relative error - observed 4.2 %) - the tolerance config 5 states for one GEMM.
"""
import pytest
import torch

import oracle
from gpu_util import rel_fro

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    import avformer_amd as A
    assert A.ops.device_ok()
    return A.ops


def _data(rows, cols, seed, kind):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(rows, cols, generator=g)
    if kind == "wide":  # block maxima spread over many binades, some blocks all zero, some values past the e4m3 range
        x = x * torch.exp2(torch.randint(-30, 30, (rows, cols // 32), generator=g).float()).repeat_interleave(32, dim=1)
        x[0, :32] = 0
        x[-1, -32:] = 0
        x[1 % rows, 0] = 3.0e38
    elif kind == "edge":  # maxima just under a power of two: the saturating corner of the conversion (449..511 -> 448)
        x = x.clamp(-0.9, 0.9)
        x[:, ::32] = 1.99
    elif kind == "tiny":
        x = x * 1e-38  # fp32 subnormal block maxima -> scale byte 0
    return x


@pytest.mark.parametrize("rows,cols", [(1, 32), (5, 128), (37, 512), (324, 1024), (130, 96)])
@pytest.mark.parametrize("kind", ["normal", "wide", "edge", "tiny"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_quant_bits(ops, rows, cols, kind, dtype):
    x = _data(rows, cols, rows * 7 + cols, kind).to(dtype)
    q, s = ops.quant_mx8(x.cuda())
    q_ref, s_ref = oracle.mx8_quant(x.float())
    assert torch.equal(s.cpu(), s_ref)
    assert torch.equal(q.cpu(), q_ref)


SHAPES = [(16, 16, 128), (128, 128, 128), (100, 36, 256), (1000, 512, 512), (1296, 1536, 512), (648, 512, 1024),
          (96, 128, 1024), (324, 1024, 512), (7, 4, 384)]


@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16])
def test_gemm_against_dequantised_product(ops, M, N, K, out_dtype):
    g = torch.Generator().manual_seed(M + 3 * N + 5 * K)
    a = torch.randn(M, K, generator=g) * torch.exp2(torch.randint(-3, 4, (M, 1), generator=g).float())
    b = torch.randn(N, K, generator=g) * 0.05
    aq, as_ = ops.quant_mx8(a.cuda())
    bq, bs = ops.quant_mx8(b.cuda())
    c = ops.gemm_mx8(aq, as_, bq, bs, out_dtype=out_dtype).float().cpu()
    ad, bd = oracle.mx8_dequant(aq, as_).double(), oracle.mx8_dequant(bq, bs).double()
    ref = ad @ bd.t()
    bound = 2e-3 * (ad.abs() @ bd.abs().t()) + 1e-30
    if out_dtype == torch.bfloat16:
        bound = bound + ref.abs() * 2.0 ** -8
    excess = ((c.double() - ref).abs() / bound).max().item()
    assert excess <= 1.0, excess


@pytest.mark.parametrize("M,N,K", [(648, 1024, 512), (200, 512, 1024)])
def test_gemm_epilogues(ops, M, N, K):
    import avformer_amd as A
    g = torch.Generator().manual_seed(11)
    a = torch.randn(M, K, generator=g)
    b = torch.randn(N, K, generator=g) * 0.05
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    aq, as_ = ops.quant_mx8(a.cuda())
    bq, bs = ops.quant_mx8(b.cuda())
    ref = (oracle.mx8_dequant(aq, as_).double() @ oracle.mx8_dequant(bq, bs).double().t()).float() + bias
    c, u = ops.gemm_mx8(aq, as_, bq, bs, out_dtype=torch.bfloat16, epilogue=A.ops.EPI_BIAS_GELU, bias=bias.cuda())
    assert rel_fro(u, ref) < 5e-3          # bf16 storage of the saved pre-activation
    assert rel_fro(c, oracle.gelu_tanh(ref)) < 6e-3
    y = ops.gemm_mx8(aq, as_, bq, bs, out_dtype=torch.float32, epilogue=A.ops.EPI_BIAS_RES, bias=bias.cuda(),
                     residual=res.cuda())
    assert rel_fro(y, ref + res) < 1e-3


@pytest.mark.parametrize("M,N,K", [(1296, 1536, 512), (1296, 512, 1024)])
def test_gemm_against_fp32_product(ops, M, N, K):
    """The stated config-5 tolerance: MX-FP8 operands against the unquantised fp32 product."""
    g = torch.Generator().manual_seed(5)
    a = torch.randn(M, K, generator=g)
    b = torch.randn(N, K, generator=g) / K ** 0.5
    aq, as_ = ops.quant_mx8(a.cuda())
    bq, bs = ops.quant_mx8(b.cuda())
    c = ops.gemm_mx8(aq, as_, bq, bs)
    assert rel_fro(c, a @ b.t()) < 0.05


def test_rejects_bad_shapes(ops):
    import avformer_amd as A
    with pytest.raises(A._lib.HipLibraryError):
        ops.quant_mx8(torch.zeros(4, 48, device="cuda"))
    aq, as_ = ops.quant_mx8(torch.zeros(4, 64, device="cuda"))
    with pytest.raises(A._lib.HipLibraryError):
        ops.gemm_mx8(aq, as_, aq, as_)  # K % 128 != 0
