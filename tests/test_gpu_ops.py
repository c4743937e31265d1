"""-m gpu: every per-operator C entry point against a plain fp32 restatement of the same op.

Tolerances: parity mode (f32) 2e-5 abs / 1e-4 rel on O(1) data; throughput mode (bf16 operands, fp32
accumulate) is judged by relative Frobenius error <= 1e-2 per op (bf16 has 8 significant bits: 2^-9 = 2e-3
per rounding)."""
import math

import pytest
import torch

import oracle
from conftest import load_golden
from gpu_util import check_abs, check_rel, max_abs, rel_fro

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    import avformer_amd as A
    assert A.ops.device_ok()
    return A.ops


@pytest.fixture(params=["bf16x3", "f32"])
def f32_arith(request):
    """the parity mode's two arithmetics (avf_set_f32_arith): three bf16 products per fp32 product on the bf16 matrix pipe
    (the default) and the f32-input MFMA; every fp32 GEMM / attention test runs in both, at the same tolerances"""
    from avformer_amd import _lib
    prev = _lib.set_f32_arithmetic(request.param)
    yield request.param
    _lib.set_f32_arithmetic(prev)


def _close(a, b, atol=2e-5, rtol=1e-4):
    if not torch.is_tensor(b):
        b = torch.tensor(b)
    torch.testing.assert_close(a.detach().float().cpu(), b.detach().float().cpu(), atol=atol, rtol=rtol)


# ---------------------------------------------------------------------------------------------- LayerNorm
@pytest.mark.parametrize("rows,D", [(7, 32), (100, 128), (37, 48), (64, 512), (5, 1536), (33, 30)])
@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16])
def test_layernorm_fwd(ops, rows, D, out_dtype):
    g = torch.Generator().manual_seed(rows * 1000 + D)
    x = torch.randn(rows, D, generator=g) * 2 + 0.5
    w = torch.randn(D, generator=g)
    b = torch.randn(D, generator=g)
    y, mean, rstd = ops.layernorm_fwd(x.cuda(), w.cuda(), b.cuda(), 1e-5, out_dtype)
    ref = oracle.layernorm(x, w, b)
    if out_dtype == torch.float32:
        _close(y, ref)
    else:
        check_rel(f"ln_fwd[{rows}x{D}]", y, ref, 4e-3)
    _close(mean, x.mean(-1))
    _close(rstd, 1 / torch.sqrt(x.var(-1, unbiased=False) + 1e-5), atol=1e-5, rtol=1e-4)


@pytest.mark.parametrize("rows,D", [(7, 32), (100, 128), (37, 48), (300, 512), (33, 30)])
@pytest.mark.parametrize("dy_dtype", [torch.float32, torch.bfloat16])
def test_layernorm_bwd(ops, rows, D, dy_dtype):
    g = torch.Generator().manual_seed(rows * 1000 + D + 1)
    x = (torch.randn(rows, D, generator=g) * 2 + 0.5).requires_grad_(True)
    w = torch.randn(D, generator=g).requires_grad_(True)
    b = torch.randn(D, generator=g).requires_grad_(True)
    dy = torch.randn(rows, D, generator=g).to(dy_dtype)
    dres = torch.randn(rows, D, generator=g)
    y = oracle.layernorm(x, w, b)
    y.backward(dy.float())
    _, mean, rstd = ops.layernorm_fwd(x.detach().cuda(), w.detach().cuda(), b.detach().cuda())
    dx, dx_lo, dg, db, cs = ops.layernorm_bwd(dy.cuda(), x.detach().cuda(), w.detach().cuda(), mean, rstd,
                                              dres=dres.cuda(), want_lo=True, want_colsum=True)
    ref_dx = x.grad + dres
    _close(dx, ref_dx, atol=5e-5)
    _close(dg, w.grad, atol=2e-4, rtol=2e-4)
    _close(db, b.grad, atol=2e-4, rtol=2e-4)
    _close(cs, ref_dx.sum(0), atol=5e-4, rtol=2e-4)
    check_rel(f"ln_bwd[{rows}x{D},{str(dy_dtype)[6:]}]:dx_lo", dx_lo, ref_dx, 4e-3)


def test_colsum_and_cast(ops):
    g = torch.Generator().manual_seed(5)
    t = torch.randn(1000, 200, generator=g)
    _close(ops.colsum(t.cuda()), t.sum(0), atol=2e-4)
    tb = t.to(torch.bfloat16)
    _close(ops.colsum(tb.cuda()), tb.float().sum(0), atol=2e-4)
    for n in (1, 3, 4, 1023, 4096 + 2):
        v = torch.randn(n, generator=g)
        assert torch.equal(ops.cast_bf16(v.cuda()).cpu(), v.to(torch.bfloat16))
    w = torch.randn(70, 100, generator=g)
    lo, lo_t = ops.prep_weight_bf16(w.cuda())
    assert torch.equal(lo.cpu(), w.to(torch.bfloat16))
    assert torch.equal(lo_t.cpu(), w.t().contiguous().to(torch.bfloat16))


# ---------------------------------------------------------------------------------------------- GEMM
def _gemm_ref(a, b, ta, tb):
    A_ = a.t() if ta else a
    B_ = b.t() if tb else b
    return A_.double() @ B_.double()


@pytest.mark.parametrize("M,N,K", [(5, 12, 8), (64, 64, 16), (130, 70, 52), (300, 256, 128), (257, 96, 260), (200, 102, 77)])
@pytest.mark.parametrize("ta,tb", [(False, True), (False, False), (True, False)])
def test_gemm_f32_forms(ops, f32_arith, M, N, K, ta, tb):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    a = torch.randn((K, M) if ta else (M, K), generator=g)
    b = torch.randn((N, K) if tb else (K, N), generator=g)
    c = ops.gemm(a.cuda(), b.cuda(), trans_a=ta, trans_b=tb)
    _close(c, _gemm_ref(a, b, ta, tb).float(), atol=1e-4 * math.sqrt(K), rtol=1e-4)


@pytest.mark.parametrize("M,N,K,ta,tb", [
    (1536, 512, 4096, True, False),    # a weight gradient: 192 tiles of 64 x 64 -> split-K (f32_split64), both operands row-fast (TN)
    (500, 1000, 2080, True, False),    # ... ragged tiles, K not a multiple of the split chunk
    (512, 1024, 4096, False, True),    # k-fast operands (NT) with a long reduction: the fold applies bias + residual
    (2048, 1024, 768, False, False),   # dX shape (NN), short reduction: unsplit
])
def test_gemm_f32_split64(ops, f32_arith, M, N, K, ta, tb):
    """the parity mode's weight-gradient-shaped GEMMs (round 5): few 64 x 64 tiles, long reduction -> split-K slabs + fold
    (ops.gemm allocates the workspace avf_gemm_workspace_bytes asks for, which is what enables the split), against fp64"""
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    a = torch.randn((K, M) if ta else (M, K), generator=g)
    b = torch.randn((N, K) if tb else (K, N), generator=g)
    ref = _gemm_ref(a, b, ta, tb)
    c = ops.gemm(a.cuda(), b.cuda(), trans_a=ta, trans_b=tb)
    _close(c, ref.float(), atol=1e-4 * math.sqrt(K), rtol=1e-4)
    err = float((c.double().cpu() - ref).norm() / ref.norm())
    # the precision class of each arithmetic: fp32 products in an fmaf chain / three bf16 products (<= 3 * 2^-16 per product in the worst case, ~4e-6 typical)
    assert err < (1e-6 if f32_arith == "f32" else 6e-6), err
    if not ta:  # the fused bias + residual epilogue rides in the fold of a split launch
        bias = torch.randn(N, generator=g)
        res = torch.randn(M, N, generator=g)
        c2 = ops.gemm(a.cuda(), b.cuda(), trans_a=ta, trans_b=tb, epilogue=ops.EPI_BIAS_RES, bias=bias.cuda(), residual=res.cuda())
        _close(c2, (ref + bias.double() + res.double()).float(), atol=1e-4 * math.sqrt(K), rtol=1e-4)


@pytest.mark.parametrize("ta,tb", [(False, True), (False, False), (True, False)])
def test_gemm_f32_bf16x3_error_bound_elementwise(ops, ta, tb):
    """the stated contract of the bf16x3 arithmetic, element by element: |C - A B| <= 3 * 2^-16 * sum_k |a_k| |b_k| (+ the fp32
    accumulation's own K * 2^-24) - on operands whose magnitudes span eight decades inside every row, where a norm-wise check
    would hide a badly rounded small term behind the large ones"""
    from avformer_amd import _lib
    M, N, K = 384, 256, 1024
    g = torch.Generator().manual_seed(77 + int(ta) + 2 * int(tb))
    def wide(shape):
        return torch.randn(shape, generator=g) * torch.pow(10.0, torch.rand(shape, generator=g) * 8 - 4)
    a = wide((K, M) if ta else (M, K))
    b = wide((N, K) if tb else (K, N))
    A_ = (a.t() if ta else a).double()
    B_ = (b.t() if tb else b).double()
    ref = A_ @ B_
    mag = A_.abs() @ B_.abs()
    prev = _lib.set_f32_arithmetic("bf16x3")
    try:
        c = ops.gemm(a.cuda(), b.cuda(), trans_a=ta, trans_b=tb).double().cpu()
    finally:
        _lib.set_f32_arithmetic(prev)
    bound = (oracle.bf16x3.PER_PRODUCT_BOUND + K * 2.0 ** -24) * mag
    worst = float(((c - ref).abs() / bound).max())
    assert worst <= 1.0, worst


@pytest.mark.parametrize("ta,tb", [(False, True), (False, False), (True, False)])
def test_gemm_f32_bf16x3_kernel_implements_the_emulated_arithmetic(ops, ta, tb):
    """the three-product kernel against the CPU emulation of ITS arithmetic (oracle/bf16x3.py): what is left between them is the
    fp32 accumulation order - 1e-6 - where the exact product is 4e-6 away.  A kernel that lost a cross term, or split by
    truncation instead of rounding, stays within 1e-3 of the exact product and fails this by an order of magnitude."""
    from avformer_amd import _lib
    M, N, K = 384, 256, 1024
    g = torch.Generator().manual_seed(91 + int(ta) + 2 * int(tb))
    a = torch.randn((K, M) if ta else (M, K), generator=g)
    b = torch.randn((N, K) if tb else (K, N), generator=g)
    A_, B_ = (a.t() if ta else a), (b.t() if tb else b)
    emu = oracle.bf16x3.matmul(A_, B_)
    exact = A_.double() @ B_.double()
    prev = _lib.set_f32_arithmetic("bf16x3")
    try:
        c = ops.gemm(a.cuda(), b.cuda(), trans_a=ta, trans_b=tb).double().cpu()
    finally:
        _lib.set_f32_arithmetic(prev)
    to_emu = float((c - emu).norm() / emu.norm())
    to_exact = float((c - exact).norm() / exact.norm())
    assert to_emu < 6e-7, (to_emu, to_exact)
    assert 2e-6 < to_exact < 8e-6, (to_emu, to_exact)


@pytest.mark.parametrize("dtype,M,N,K", [(torch.float32, 200, 136, 96), (torch.bfloat16, 200, 136, 96),
                                         (torch.bfloat16, 301, 260, 1088), (torch.bfloat16, 8200, 520, 64)])
def test_gemm_epilogues(ops, f32_arith, dtype, M, N, K):
    if dtype != torch.float32 and f32_arith == "f32":
        pytest.skip("the arithmetic switch only concerns fp32 operands")
    g = torch.Generator().manual_seed(11)
    a = torch.randn(M, K, generator=g).to(dtype)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dtype)
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    base = a.double() @ w.double().t() + bias.double()
    tol = dict(atol=1e-4, rtol=1e-4) if dtype == torch.float32 else dict(atol=3e-2, rtol=2e-2)
    # bias + residual -> fp32
    c = ops.gemm(a.cuda(), w.cuda(), out_dtype=torch.float32, epilogue=ops.EPI_BIAS_RES, bias=bias.cuda(), residual=res.cuda())
    _close(c, (base + res.double()).float(), atol=1e-4 if dtype == torch.float32 else 1e-3, rtol=1e-4)
    # bias + GELU (aux = pre-activation)
    c, aux = ops.gemm(a.cuda(), w.cuda(), epilogue=ops.EPI_BIAS_GELU, bias=bias.cuda())
    _close(aux, base.float(), **tol)
    _close(c, oracle.gelu_tanh(base.float()), **tol)
    # dGELU: C = (A W^T) * gelu'(aux)
    u = torch.randn(M, N, generator=g).to(dtype)
    uu = u.float().clone().requires_grad_(True)
    oracle.gelu_tanh(uu).sum().backward()
    c = ops.gemm(a.cuda(), w.cuda(), epilogue=ops.EPI_DGELU, aux=u.cuda())
    _close(c, ((a.double() @ w.double().t()) * uu.grad.double()).float(), **tol)


@pytest.mark.parametrize("M,N,K", [(500, 96, 256), (130, 260, 72)])
def test_gemm_bf16_operands_fp32_outputs_keep_aux_in_fp32(ops, M, N, K):
    """bf16 operands with c_dtype fp32: the saved pre-activation is stored / read in C's type (found by tests/fuzz/fuzz_gemm.py:
    it used to be written as bf16 into the caller's fp32 buffer)"""
    g = torch.Generator().manual_seed(M + N)
    a = torch.randn(M, K, generator=g).bfloat16()
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).bfloat16()
    bias = torch.randn(N, generator=g)
    base = (a.double() @ w.double().t() + bias.double()).float()
    c, aux = ops.gemm(a.cuda(), w.cuda(), out_dtype=torch.float32, epilogue=ops.EPI_BIAS_GELU, bias=bias.cuda())
    assert aux.dtype == torch.float32
    _close(aux, base, atol=1e-3, rtol=1e-4)
    _close(c, oracle.gelu_tanh(base), atol=3e-3, rtol=2e-3)  # gelu_tanh_fast
    u = torch.randn(M, N, generator=g)
    uu = u.clone().requires_grad_(True)
    oracle.gelu_tanh(uu).sum().backward()
    c = ops.gemm(a.cuda(), w.cuda(), out_dtype=torch.float32, epilogue=ops.EPI_DGELU, aux=u.cuda())
    _close(c, ((a.double() @ w.double().t()) * uu.grad.double()).float(), atol=3e-3, rtol=2e-3)


@pytest.mark.parametrize("M,N,K", [(8, 8, 8), (64, 128, 64), (200, 136, 96), (384, 1536, 512), (1000, 48, 40),
                                   (1000, 512, 1024), (301, 132, 1088), (8200, 1024, 128)])  # 96x128 / 128x128x8w tiles
def test_gemm_bf16_nt(ops, M, N, K):
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16)
    b = torch.randn(N, K, generator=g).to(torch.bfloat16)
    ref = _gemm_ref(a, b, False, True)
    c32 = ops.gemm(a.cuda(), b.cuda(), out_dtype=torch.float32)
    # fp32 accumulation of exact bf16 products: only summation-order noise
    _close(c32, ref.float(), atol=2e-5 * K, rtol=1e-5)
    c16 = ops.gemm(a.cuda(), b.cuda())
    check_rel(f"gemm_nt[{M}x{N}x{K}]", c16, ref, 4e-3)


@pytest.mark.parametrize("M,N,K", [(8, 8, 8), (64, 128, 64), (136, 96, 200), (1536, 512, 2048), (48, 40, 1000),
                                   (512, 512, 16384)])
def test_gemm_bf16_tn(ops, M, N, K):
    """weight-gradient form: C[M,N] = A[K,M]^T B[K,N] (reduction over tokens), incl. the split-K path"""
    g = torch.Generator().manual_seed(M + N + K + 1)
    a = torch.randn(K, M, generator=g).to(torch.bfloat16)
    b = torch.randn(K, N, generator=g).to(torch.bfloat16)
    ref = _gemm_ref(a, b, True, False)
    c = ops.gemm(a.cuda(), b.cuda(), trans_a=True, trans_b=False, out_dtype=torch.float32)
    _close(c, ref.float(), atol=2e-5 * K, rtol=1e-5)


# ---------------------------------------------------------------------------------------------- attention
def _attn_ref(qkv, B, N, H, dh, d_o=None):
    """fp64 restatement of heads.py:222-237 on the packed projection."""
    I = H * dh
    qkv = qkv.double().clone().requires_grad_(True)
    q, k, v = qkv.view(B, N, 3 * I).split(I, dim=-1)
    sh = lambda t: t.reshape(B, N, H, dh).permute(0, 2, 1, 3)
    q, k, v = sh(q), sh(k), sh(v)
    s = (q @ k.transpose(-1, -2)) * dh ** -0.5
    p = s.softmax(-1)
    o = (p @ v).permute(0, 2, 1, 3).reshape(B * N, I)
    lse2 = torch.logsumexp(s, dim=-1) * math.log2(math.e)
    dqkv = None
    if d_o is not None:
        o.backward(d_o.double())
        dqkv = qkv.grad
    return o.detach(), lse2.detach(), dqkv


@pytest.mark.parametrize("B,N,H,dh", [(2, 7, 4, 8), (1, 64, 2, 32), (2, 49, 8, 32), (3, 17, 8, 64), (2, 130, 3, 64),
                                      (1, 324, 2, 64), (2, 12, 8, 16)])
def test_attention_f32(ops, f32_arith, B, N, H, dh):
    g = torch.Generator().manual_seed(B * 100 + N + dh)
    qkv = torch.randn(B * N, 3 * H * dh, generator=g)
    d_o = torch.randn(B * N, H * dh, generator=g)
    o_ref, lse_ref, dqkv_ref = _attn_ref(qkv, B, N, H, dh, d_o)
    o, lse2 = ops.attn_fwd(qkv.cuda(), B, N, H, dh)
    _close(o, o_ref.float())
    _close(lse2, lse_ref.float(), atol=1e-4)
    dqkv = ops.attn_bwd(qkv.cuda(), o, d_o.cuda(), lse2, B, N, H, dh)
    _close(dqkv, dqkv_ref.float(), atol=5e-5, rtol=1e-4)


def _prescale_q(qkv, H, dh):
    """bf16 projection with q' = bf16(q * log2(e)/sqrt(dh)) in the q columns, and the fp32 projection it stands for:
    (q'/c | k | v) exactly, so that the reference sees the very numbers the kernels see"""
    c = math.log2(math.e) / math.sqrt(dh)
    I = H * dh
    dev = qkv.float().clone()
    dev[:, :I] = (dev[:, :I] * c).to(torch.bfloat16).float()
    ref = dev.clone()
    ref[:, :I] = ref[:, :I] / c
    return dev.to(torch.bfloat16), ref


@pytest.mark.parametrize("qs", [False, True], ids=["raw_q", "prescaled_q"])
@pytest.mark.parametrize("B,N,H,dh", [(1, 64, 2, 32), (2, 49, 8, 32), (3, 17, 8, 64), (2, 130, 3, 64), (1, 324, 2, 64),
                                      (2, 12, 8, 32), (1, 512, 2, 64), (1, 200, 1, 32)])
def test_attention_bf16(ops, B, N, H, dh, qs):
    g = torch.Generator().manual_seed(B * 100 + N + dh + 1)
    qkv = torch.randn(B * N, 3 * H * dh, generator=g).to(torch.bfloat16)
    d_o = torch.randn(B * N, H * dh, generator=g).to(torch.bfloat16)
    ref_in = qkv.float()
    if qs:
        qkv, ref_in = _prescale_q(qkv, H, dh)
    o_ref, lse_ref, dqkv_ref = _attn_ref(ref_in, B, N, H, dh, d_o.float())
    o, lse2 = ops.attn_fwd(qkv.cuda(), B, N, H, dh, q_prescaled=qs)
    tag = f"attn[{B}x{N}x{H}x{dh},qs{int(qs)}]"
    check_rel(tag + ":o", o, o_ref, 1e-2)
    _close(lse2, lse_ref.float(), atol=2e-2, rtol=1e-3)
    dqkv = ops.attn_bwd(qkv.cuda(), o, d_o.cuda(), lse2, B, N, H, dh, q_prescaled=qs)
    I = H * dh
    for name, sl in (("dq", slice(0, I)), ("dk", slice(I, 2 * I)), ("dv", slice(2 * I, 3 * I))):
        check_rel(f"{tag}:{name}", dqkv[:, sl], dqkv_ref[:, sl], 2e-2)


@pytest.mark.parametrize("N", [1, 15, 16, 31, 33, 63, 64, 65, 96, 127, 128, 200, 256, 320, 383, 384, 385, 448, 500, 512,
                               575, 576, 577, 640])
@pytest.mark.parametrize("qs", [False, True], ids=["raw_q", "prescaled_q"])
def test_attention_bf16_dh64_lengths(ops, N, qs):
    """dim_head 64 over the sequence lengths where the kernel family changes: one group per wave (N <= 384, 8- and
    12-wave builds), several groups per wave (385..576), the streaming kernels (> 576); tails of 1..63 rows."""
    B, H, dh = 2, 2, 64
    g = torch.Generator().manual_seed(1000 + N)
    qkv = torch.randn(B * N, 3 * H * dh, generator=g).to(torch.bfloat16)
    d_o = torch.randn(B * N, H * dh, generator=g).to(torch.bfloat16)
    ref_in = qkv.float()
    if qs:
        qkv, ref_in = _prescale_q(qkv, H, dh)
    o_ref, lse_ref, dqkv_ref = _attn_ref(ref_in, B, N, H, dh, d_o.float())
    o, lse2 = ops.attn_fwd(qkv.cuda(), B, N, H, dh, q_prescaled=qs)
    assert torch.isfinite(o.float()).all() and torch.isfinite(lse2).all()
    tag = f"attn_len[{N},qs{int(qs)}]"
    check_rel(tag + ":o", o, o_ref, 1e-2)
    _close(lse2, lse_ref.float(), atol=2e-2, rtol=1e-3)
    dqkv = ops.attn_bwd(qkv.cuda(), o, d_o.cuda(), lse2, B, N, H, dh, q_prescaled=qs)
    assert torch.isfinite(dqkv.float()).all()
    I = H * dh
    for name, sl in (("dq", slice(0, I)), ("dk", slice(I, 2 * I)), ("dv", slice(2 * I, 3 * I))):
        if N == 1 and name != "dv":  # a single key: p = 1, dS = 0 exactly -> dq = dk = 0 (no relative error to take)
            assert max_abs(dqkv[:, sl], dqkv_ref[:, sl]) < 1e-2
            continue
        check_rel(f"{tag}:{name}", dqkv[:, sl], dqkv_ref[:, sl], 2e-2)


@pytest.mark.parametrize("qs", [False, True], ids=["raw_q", "prescaled_q"])
@pytest.mark.parametrize("mode", ["ramp", "small_steps", "huge", "descending", "very_negative"])
def test_attention_bf16_rescale_paths(ops, mode, qs):
    """lazy rescaling of the head-resident forward: 'ramp' = the row maximum grows by much more than the threshold in
    every key tile (rescale each tile); 'small_steps' = it grows by less than the threshold per tile (stale maximum,
    probabilities above 1 in the accumulators); 'huge' = scores of magnitude ~1e3 (log2 domain ~1.4e3)."""
    B, N, H, dh = 1, 324, 1, 64
    g = torch.Generator().manual_seed(11)
    q = torch.randn(N, dh, generator=g)
    k = torch.randn(N, dh, generator=g)
    v = torch.randn(N, dh, generator=g)
    u = torch.randn(dh, generator=g)
    u = u / u.norm()
    if mode == "ramp":      # score(q_i, k_j) ~ 8 * 6 * j/64 / 8 ... grows ~6 nats per 64 keys
        q = q * 0.2 + 8.0 * u
        k = k * 0.2 + u[None, :] * (torch.arange(N).float()[:, None] / 64.0) * 6.0
    elif mode == "small_steps":  # ~2.5 nats (3.6 in log2) per tile: below the threshold of 6
        q = q * 0.2 + 8.0 * u
        k = k * 0.2 + u[None, :] * (torch.arange(N).float()[:, None] / 64.0) * 2.5
    elif mode == "descending":  # the first tile holds the row maxima; later tiles fall by ~6 nats per 64 keys
        q = q * 0.2 + 8.0 * u
        k = k * 0.2 - u[None, :] * (torch.arange(N).float()[:, None] / 64.0) * 6.0
    elif mode == "very_negative":  # every score ~ -250 nats: the first tile must centre the maximum far below zero
        q = q * 0.2 + 40.0 * u
        k = k * 0.2 - 50.0 * u[None, :]
    else:
        q = q * 30.0
        k = k * 30.0
    qkv = torch.cat([q, k, v], dim=1).to(torch.bfloat16)
    d_o = torch.randn(N, dh, generator=g).to(torch.bfloat16)
    ref_in = qkv.float()
    if qs:
        qkv, ref_in = _prescale_q(qkv, H, dh)
    o_ref, lse_ref, dqkv_ref = _attn_ref(ref_in, B, N, H, dh, d_o.float())
    o, lse2 = ops.attn_fwd(qkv.cuda(), B, N, H, dh, q_prescaled=qs)
    assert torch.isfinite(o.float()).all() and torch.isfinite(lse2).all()
    check_rel(f"attn_rescale[{mode},qs{int(qs)}]:o", o, o_ref, 1e-2)
    _close(lse2, lse_ref.float(), atol=5e-2, rtol=2e-3)
    dqkv = ops.attn_bwd(qkv.cuda(), o, d_o.cuda(), lse2, B, N, H, dh, q_prescaled=qs)
    assert torch.isfinite(dqkv.float()).all()
    check_rel(f"attn_rescale[{mode},qs{int(qs)}]:dqkv", dqkv, dqkv_ref, 3e-2)


def test_attention_bf16_spiked_scores(ops):
    """online-softmax rescale path: one key dominates late in the sequence (max jumps at a later tile)"""
    B, N, H, dh = 1, 256, 1, 64
    g = torch.Generator().manual_seed(3)
    qkv = torch.randn(B * N, 3 * dh, generator=g)
    qkv[200, dh:2 * dh] = qkv[5, 0:dh] * 6.0  # key 200 aligned with query 5 -> huge score in tile 3
    qkv = qkv.to(torch.bfloat16)
    o_ref, lse_ref, _ = _attn_ref(qkv.float(), B, N, H, dh)
    o, lse2 = ops.attn_fwd(qkv.cuda(), B, N, H, dh)
    check_rel("attn_spiked:o", o, o_ref, 1e-2)
    check_abs("attn_spiked:o_maxabs", o, o_ref, 5e-2)
    _close(lse2, lse_ref.float(), atol=5e-2, rtol=1e-3)


# ---------------------------------------------------------------------------------------------- token plumbing
@pytest.mark.parametrize("B,Tv,Ta,D", [(2, 3, 2, 8), (3, 196, 128, 512), (1, 0, 5, 12), (4, 7, 0, 36), (2, 33, 31, 260)])
def test_fuse_tokens(ops, B, Tv, Ta, D):
    g = torch.Generator().manual_seed(B + Tv + Ta + D)
    clip = torch.randn(B, Tv, D, generator=g).cuda()
    audio = torch.randn(B, Ta, D, generator=g).cuda()
    pos = torch.randn(Tv + Ta, D, generator=g).cuda()
    ref = torch.cat([clip, audio], 1) + pos
    assert torch.equal(ops.fuse_tokens(clip, audio, pos), ref)       # one fp32 add per element: bit-exact
    assert torch.equal(ops.fuse_tokens(clip, audio, None), torch.cat([clip, audio], 1))


@pytest.mark.parametrize("B,T,D", [(2, 5, 8), (32, 324, 512), (3, 1, 12), (1, 1000, 36), (5, 77, 260), (2, 64, 768)])
def test_token_mean_fwd_bwd(ops, B, T, D):
    g = torch.Generator().manual_seed(B + T + D)
    y = torch.randn(B, T, D, generator=g).cuda()
    m = ops.token_mean_fwd(y)
    _close(m, y.double().mean(1).float(), atol=1e-5, rtol=1e-5)
    gr = torch.randn(B, D, generator=g).cuda()
    dy, lo, cs = ops.token_mean_bwd(gr, T, want_bf16=True, want_colsum=True)
    ref = (gr / T)[:, None, :].expand(B, T, D)
    _close(dy, ref, atol=1e-7, rtol=1e-6)
    assert torch.equal(lo, dy.to(torch.bfloat16))
    _close(cs, gr.double().sum(0).float(), atol=1e-5, rtol=1e-5)
    dy2, lo2, cs2 = ops.token_mean_bwd(gr, T)
    assert lo2 is None and cs2 is None and torch.equal(dy2, dy)


# ---------------------------------------------------------------------------------------------- AU loss
@pytest.mark.parametrize("tag", ["all", "ign"])
def test_au_loss_golden(ops, tag):
    import avformer_amd as A
    g = load_golden("g8_au_loss")
    crit = A.AULoss().cuda()
    z = g[f"{tag}.z"].cuda().requires_grad_(True)
    loss = crit(z, g[f"{tag}.y"].cuda())
    loss.backward()
    _close(loss, g[f"{tag}.loss"], atol=1e-6, rtol=1e-5)
    _close(z.grad, g[f"{tag}.dz"], atol=1e-7, rtol=1e-5)


def test_au_loss_strided_and_all_ignored(ops):
    import avformer_amd as A
    crit = A.AULoss().cuda()
    g = torch.Generator().manual_seed(2)
    out = torch.randn(9, 21, generator=g)
    y = (torch.rand(9, 12, generator=g) > 0.5).float()
    y[4] = -1
    ref = oracle.au_loss(out[:, :12], y)
    got = crit(out.cuda()[:, :12], y.cuda())  # strided view, as get_au_loss passes it (avformer.py:116)
    _close(got, ref, atol=1e-6, rtol=1e-5)
    assert torch.isnan(crit(out.cuda()[:, :12], -torch.ones(9, 12).cuda()))


def test_au_loss_on_output_rows_equals_the_sliced_form(ops):
    """AULoss.forward_rows (avf_au_loss_wide: the loss on slots 0..11 of the model's [B,21] rows, the gradient written in that
    layout by the loss kernel) against AULoss on the slice out[:, :12] - value and gradient bit-identical, the other slots' gradient
    exactly zero; ignored rows, the (sum, count) form, every row ignored (NaN), and the fallback for a non-contiguous output"""
    import avformer_amd as A
    crit = A.AULoss().cuda()
    g = torch.Generator().manual_seed(21)
    out = torch.randn(11, 21, generator=g)
    y = (torch.rand(11, 12, generator=g) > 0.5).float()
    y[3] = -1
    y[7] = -1
    a = out.clone().cuda().requires_grad_(True)
    b = out.clone().cuda().requires_grad_(True)
    la = crit(a[:, :12], y.cuda())
    lb = crit.forward_rows(b, y.cuda())
    (3.0 * la).backward()
    (3.0 * lb).backward()
    assert torch.equal(la, lb) and torch.equal(a.grad, b.grad) and float(b.grad[:, 12:].abs().max()) == 0.0
    _close(lb, oracle.au_loss(out[:, :12], y), atol=1e-6, rtol=1e-5)
    sc_n, g_n = ops.au_loss_sum(out.cuda()[:, :12], y.cuda(), crit.pos_weight)
    sc_w, g_w = ops.au_loss_wide(out.cuda(), y.cuda(), crit.pos_weight, sum_mode=True)
    assert torch.equal(sc_n, sc_w) and torch.equal(g_n, g_w[:, :12]) and float(g_w[:, 12:].abs().max()) == 0.0
    assert torch.isnan(crit.forward_rows(out.cuda(), -torch.ones(11, 12).cuda()))
    wide = torch.randn(11, 42, generator=g).cuda()[:, ::2]  # not contiguous: the sliced form serves it
    assert torch.equal(crit.forward_rows(wide, y.cuda()), crit(wide[:, :12], y.cuda()))


@pytest.mark.parametrize("K,shapes", [
    (128, [(64, 64)]),                                               # one 128 x 128 tile, two K-steps
    (4096, [(1536, 512), (1024, 512), (512, 1024), (512, 512)]),     # a d=512 layer: the 256 x 128 kernel (64 tiles)
    (2112, [(2304, 768), (1536, 768), (768, 1536), (768, 768)]),     # d=768 (C4): ragged split-K chunks
    (2048, [(1000, 136), (264, 520), (8, 8)]),                       # ragged tiles in both extents
])
def test_gemm_tn_group(ops, K, shapes):
    """the grouped weight-gradient launch (avf_gemm_tn_group) against fp64 products, through both tile configurations
    (the host picks 256 x 128 for big groups; AVF_TN_BIG forces one or the other in the A/B tools)"""
    g = torch.Generator().manual_seed(K + len(shapes))
    pairs = [(torch.randn(K, m, generator=g).bfloat16(), torch.randn(K, n, generator=g).bfloat16()) for m, n in shapes]
    outs = ops.gemm_tn_group([(a.cuda(), b.cuda()) for a, b in pairs])
    for (a, b), c in zip(pairs, outs):
        ref = a.double().t() @ b.double()
        _close(c, ref.float(), atol=2e-5 * K, rtol=1e-5)


def test_gemm_tn_group_big_tile_on_ragged_shapes():
    """the 256 x 128 kernel forced (AVF_TN_BIG=1, read once per process) onto shapes whose edges are ragged in both tile
    extents and whose K splits unevenly - a child process, as the choice is cached at first use"""
    import os
    import subprocess
    import sys
    code = r'''
import sys, torch
sys.path.insert(0, %r)
import avformer_amd as A
for K, shapes in [(2048, [(1000, 136), (264, 520), (8, 8)]), (192, [(256, 128)]), (4160, [(520, 1032), (1544, 72)]), (64, [(24, 40)])]:
    g = torch.Generator().manual_seed(K)
    pairs = [(torch.randn(K, m, generator=g).bfloat16(), torch.randn(K, n, generator=g).bfloat16()) for m, n in shapes]
    outs = A.ops.gemm_tn_group([(a.cuda(), b.cuda()) for a, b in pairs])
    for (a, b), c in zip(pairs, outs):
        ref = (a.double().t() @ b.double()).float()
        torch.testing.assert_close(c.cpu(), ref, atol=2e-5 * K, rtol=1e-5)
print("TN_BIG_OK")
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, AVF_TUNING="1", AVF_TN_BIG="1"),
                       timeout=600)
    assert r.returncode == 0 and "TN_BIG_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
