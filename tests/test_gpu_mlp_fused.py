"""The FeedForward sublayer as one launch per direction (csrc/mlp_fused.hip) against the two-GEMM path it replaces
(reference models/heads.py:188-199: Linear -> GELU -> Linear, + residual)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _inputs(R, D, M, res_dtype, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    h = torch.randn(R, D, generator=g).to(DEV).bfloat16()
    w1 = (torch.randn(M, D, generator=g) * D ** -0.5).to(DEV).bfloat16()
    w2 = (torch.randn(D, M, generator=g) * M ** -0.5).to(DEV).bfloat16()
    b1 = (torch.randn(M, generator=g) * 0.1).to(DEV)
    b2 = (torch.randn(D, generator=g) * 0.1).to(DEV)
    x_mid = torch.randn(R, D, generator=g).to(DEV).to(res_dtype)
    return h, w1, b1, w2, b2, x_mid


@pytest.mark.parametrize("R,D,M", [(64, 512, 128), (128, 512, 1024), (640, 256, 512), (192, 768, 384), (10368, 512, 1024)])
@pytest.mark.parametrize("res_dtype", [torch.bfloat16, torch.float32])
def test_fused_forward_matches_two_gemms(R, D, M, res_dtype):
    import avformer_amd as A
    ops = A.ops
    assert ops.mlp_fused_ok(R, D, M)
    h, w1, b1, w2, b2, x_mid = _inputs(R, D, M, res_dtype)
    x_out, u, g = ops.mlp_fused_fwd(h, w1, b1, w2, b2, x_mid)
    g_ref, u_ref = ops.gemm(h, w1, epilogue=ops.EPI_BIAS_GELU, bias=b1)
    out_ref = ops.gemm(g_ref, w2, epilogue=ops.EPI_BIAS_RES, bias=b2, residual=x_mid, out_dtype=res_dtype)
    # same MFMA products, same k order inside a 64-wide step, different accumulation grouping across steps: fp32 round-off
    # in front of the bf16 rounding of u / g, i.e. at most one bf16 ulp on a few elements
    assert (u.float() - u_ref.float()).abs().max() <= 2 ** -7 * u_ref.float().abs().max()
    assert (u != u_ref).float().mean() < 0.02
    assert (g != g_ref).float().mean() < 0.03
    ref64 = (torch.nn.functional.gelu((h.double() @ w1.double().T + b1.double()).bfloat16().double(), approximate="tanh").bfloat16().double()
             @ w2.double().T + b2.double() + x_mid.double())
    e_fused = (x_out.double() - ref64).abs().max().item()
    e_two = (out_ref.double() - ref64).abs().max().item()
    assert e_fused <= max(1.5 * e_two, 2e-2), (e_fused, e_two)


@pytest.mark.parametrize("R,D,M", [(64, 512, 128), (128, 512, 1024), (640, 256, 512), (192, 768, 384), (10368, 512, 1024)])
def test_fused_backward_matches_two_gemms(R, D, M):
    import avformer_amd as A
    ops = A.ops
    h, w1, b1, w2, b2, _ = _inputs(R, D, M, torch.bfloat16, seed=1)
    dy = h  # any bf16 [R, D]
    u = (torch.randn(R, M, device=DEV) * 1.5).bfloat16()
    w2_t = w2.T.contiguous()  # [M, D]
    w1_t = w1.T.contiguous()  # [D, M]
    du, dh, part = ops.mlp_fused_bwd(dy, w2_t, w1_t, u)
    du_ref = ops.gemm(dy, w2_t, epilogue=ops.EPI_DGELU, aux=u)
    dh_ref = ops.gemm(du_ref, w1_t)
    assert (du != du_ref).float().mean() < 0.02
    assert (du.float() - du_ref.float()).abs().max() <= 2 ** -6 * du_ref.float().abs().max()
    dh64 = du_ref.double() @ w1_t.double().T
    e_fused = (dh.double() - dh64).abs().max().item()
    e_two = (dh_ref.double() - dh64).abs().max().item()
    assert e_fused <= max(1.5 * e_two, 2e-2), (e_fused, e_two)
    db1 = part.sum(0)
    # (the kernel sums the fp32 values in front of the bf16 rounding of du: the reference sum differs by the rounding of its R terms)
    ref = du.float().sum(0)
    assert (db1 - ref).abs().max() <= 2 ** -8 * du.float().abs().sum(0).max() + 1e-3


def test_shapes_outside_the_fused_kernel_are_refused():
    import avformer_amd as A
    assert not A.ops.mlp_fused_ok(100, 512, 1024)   # rows % 64
    assert not A.ops.mlp_fused_ok(128, 384, 1024)   # dim
    assert not A.ops.mlp_fused_ok(128, 512, 1000)   # mlp_dim % 128
