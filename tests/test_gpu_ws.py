"""Weight-stationary persistent NT GEMM (csrc/gemm_ws.hip) and the lean epilogue (gemm_nt.hpp): the K = 512 GEMMs of
/root/reference/models/heads.py:191,195,212,215 (nn.Linear forward / dX at dim 512).

The persistent kernel must return bit for bit what the tiled kernel returns (same epilogue arithmetic, same k order of the fp32
accumulation), for every epilogue, both output types, ragged row counts and every ring depth; both are held to an fp64
restatement; the fragment-major weight image is a pure permutation; the column sums of the dGELU form equal the sums of the
stored values."""
import os

import pytest
import torch

import avformer_amd as A
from gpu_util import check

pytestmark = pytest.mark.gpu
ops = A.ops
bf = torch.bfloat16


def _inputs(M, N, epi, od, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    a = torch.randn(M, 512, device="cuda", generator=g).to(bf)
    w = (torch.randn(N, 512, device="cuda", generator=g) / 512 ** 0.5).to(bf)
    bias = torch.randn(N, device="cuda", generator=g) if epi in (ops.EPI_BIAS_RES, ops.EPI_BIAS_GELU) else None
    res = torch.randn(M, N, device="cuda", generator=g).to(od) if epi == ops.EPI_BIAS_RES else None
    aux = torch.randn(M, N, device="cuda", generator=g).to(od) if epi == ops.EPI_DGELU else None
    return a, w, bias, res, aux


def _gelu(u):
    return 0.5 * u * (1.0 + torch.tanh(0.7978845608028654 * (u + 0.044715 * u ** 3)))


def _dgelu(u):
    t = torch.tanh(0.7978845608028654 * (u + 0.044715 * u ** 3))
    return 0.5 * (1 + t) + 0.5 * u * (1 - t * t) * 0.7978845608028654 * (1 + 3 * 0.044715 * u * u)


def _ref64(a, w, bias, res, aux, epi):
    c = a.double() @ w.double().t()
    if bias is not None:
        c = c + bias.double()
    if epi == ops.EPI_BIAS_RES:
        return (c + res.double(),)
    if epi == ops.EPI_BIAS_GELU:
        return (_gelu(c), c)
    if epi == ops.EPI_DGELU:
        return (c * _dgelu(aux.double()),)
    return (c,)


def test_packed_image_is_a_permutation_of_the_weight():
    w = torch.arange(1024 * 512, device="cuda", dtype=torch.float32).remainder(251.0).to(bf).view(1024, 512)
    wp = ops.pack_ws(w)
    assert wp.numel() == w.numel()
    # chunk (panel, wave, j, s, lane) = W[256 panel + 32 wave + 16 j + (lane & 15)][32 s + 8 (lane >> 4) .. + 7]
    v = wp.view(4, 8, 2, 16, 4, 16, 8)  # panel, wave, j, s, lg, li, k8
    back = v.permute(0, 1, 2, 5, 3, 4, 6).reshape(1024, 512)  # -> (panel, wave, j, li) rows x (s, lg, k8) columns
    assert torch.equal(back, w)
    with pytest.raises(ValueError):
        ops.pack_ws(torch.zeros(100, 512, device="cuda", dtype=bf))


@pytest.mark.parametrize("epi_name", ["none", "res", "gelu", "dgelu"])
@pytest.mark.parametrize("out", ["bf16", "f32"])
@pytest.mark.parametrize("M,N", [(16384, 512), (10368, 1024), (4099, 1536), (2048, 256),
                                 # last tile ragged by 16 rows / by 24: as many groups as tiles, one tile each
                                 (2064, 256), (2056, 256)])
def test_persistent_kernel_equals_the_tiled_kernel_bit_for_bit(epi_name, out, M, N):
    epi = {"none": ops.EPI_NONE, "res": ops.EPI_BIAS_RES, "gelu": ops.EPI_BIAS_GELU, "dgelu": ops.EPI_DGELU}[epi_name]
    od = bf if out == "bf16" else torch.float32
    if epi == ops.EPI_DGELU and out == "f32":
        pytest.skip("the layer never asks for an fp32 dGELU product; the persistent kernel refuses it")
    a, w, bias, res, aux = _inputs(M, N, epi, od, seed=M + N)
    r0 = ops.gemm(a, w, out_dtype=od, epilogue=epi, bias=bias, residual=res, aux=aux)
    r1 = ops.gemm_ws(a, ops.pack_ws(w), N, out_dtype=od, epilogue=epi, bias=bias, residual=res, aux=aux)
    r0 = r0 if isinstance(r0, tuple) else (r0,)
    r1 = r1 if isinstance(r1, tuple) else (r1,)
    assert len(r0) == len(r1)
    for x, y in zip(r0, r1):
        assert torch.equal(x, y), float((x.float() - y.float()).abs().max())
    ref = _ref64(a, w, bias, res, aux, epi)
    for i, (y, r) in enumerate(zip(r1, ref)):
        err = float((y.double() - r).norm() / r.norm())
        check(f"ws_{epi_name}_{out}_{M}x{N}_{i}", err, 1e-2 if out == "bf16" else 2e-5)


def test_column_sums_of_the_dgelu_form():
    M, N = 10368, 1024
    a, w, _, _, aux = _inputs(M, N, ops.EPI_DGELU, bf, seed=5)
    c, cs = ops.gemm_ws(a, ops.pack_ws(w), N, out_dtype=bf, epilogue=ops.EPI_DGELU, aux=aux, want_colsum=True)
    ref = _ref64(a, w, None, None, aux, ops.EPI_DGELU)[0]
    # the sums are taken of the fp32 values BEFORE their rounding to bf16 (as the tiled kernel's), in fp32
    err = float((cs.double() - ref.sum(0)).abs().max() / ref.abs().sum(0).max())
    assert err < 2e-5, err
    assert torch.equal(c, ops.gemm(a, w, out_dtype=bf, epilogue=ops.EPI_DGELU, aux=aux))


def test_dgelu_form_emits_the_mx_fp8_image_of_its_product():
    """the fp8 mode's dGELU GEMM runs on this kernel with bf16 operands and writes the MX-FP8 image of du for the fp8 GEMM
    behind it.  Exact check: sparse {-1, 0, 1} operands make every accumulator a small integer and u = 30 makes gelu'(u) = 1
    exactly, so the fp32 value behind C IS the stored bf16 value and the image must be the OCP conversion of C bit for bit."""
    import oracle
    M, N = 4096, 1024
    g = torch.Generator(device="cuda").manual_seed(11)
    a = ((torch.rand(M, 512, device="cuda", generator=g) < 1 / 16).float() * torch.randint(0, 2, (M, 512), device="cuda", generator=g).mul(2).sub(1)).to(bf)
    w = torch.randint(-1, 2, (N, 512), device="cuda", generator=g).to(bf)
    u = torch.full((M, N), 30.0, device="cuda").to(bf)
    c, cs, q, sc = ops.gemm_ws(a, ops.pack_ws(w), N, out_dtype=bf, epilogue=ops.EPI_DGELU, aux=u, want_colsum=True, want_image=True)
    ref = a.double() @ w.double().t()
    assert torch.equal(c.double(), ref)
    q_ref, s_ref = oracle.mx8_quant(c.float().cpu())
    assert torch.equal(sc.cpu(), s_ref)
    assert torch.equal(q.cpu(), q_ref)
    assert torch.equal(cs.double(), ref.sum(0))


def test_unfit_shapes_are_refused_and_the_layer_falls_back():
    a = torch.randn(1024, 512, device="cuda").to(bf)  # fewer than 2048 rows
    w = torch.randn(512, 512, device="cuda").to(bf)
    with pytest.raises(RuntimeError, match="weight-stationary"):
        ops.gemm_ws(a, ops.pack_ws(w), 512)


@pytest.mark.parametrize("dropout", [0.0, 0.2])
def test_stack_with_and_without_the_persistent_kernel(tmp_path, dropout):
    """AVF_NT_WS / AVF_NT_LEAN are read once per process (and exist under AVF_TUNING=1 only): three child processes run the same
    2-layer stack (forward + backward) on (a) the persistent kernel + lean epilogue, (b) the tiled kernel + lean epilogue,
    (c) the tiled kernel + the general run-time-option epilogue; then one FusedAdam step - the only writer of the
    fragment-major weight images from then on - and a second forward.  dropout = 0.2 (the reference's heads.py:277): the
    lean epilogue's compile-time dropout site (DROP) against the general epilogue's, same seed (torch.manual_seed), same
    masks.  Every tensor is bit-identical except the gradient of net.0's bias, whose column sums go through per-tile (tiled
    kernel) or per-workgroup (persistent kernel) fp32 partial sums (and what that bias feeds after the step: the second
    forward is held to 1e-3)."""
    import subprocess
    import sys
    code = (
        "import os, torch, avformer_amd as A\n"
        "torch.manual_seed(3)\n"
        f"m = A.Transformer(512, 2, 8, 64, 1024, {dropout}, compute_dtype='bf16', residual_dtype='bf16').cuda().train()\n"
        "x = torch.randn(8, 324, 512, device='cuda', requires_grad=True)\n"
        "y = m(x); y.float().pow(2).mean().backward()\n"
        "d = {'y': y.detach().float().cpu(), 'dx': x.grad.cpu(), 'seed': torch.tensor(float(m.last_seed % 65536))}\n"
        "d.update({n: p.grad.clone().cpu() for n, p in m.named_parameters()})\n"
        # ... then one library Adam step (which rewrites the weight images, the fragment-major ones included) and a forward on them
        "opt = A.optim.FusedAdam(m, lr=1e-3)\n"
        "opt.step()\n"
        "m.eval()\n"
        "with torch.no_grad():\n"
        "    d['y_after_step'] = m(x.detach()).float().cpu()\n"
        "torch.save(d, os.environ['AVF_TEST_OUT'])\n")
    outs = []
    for ws, lean in (("1", "1"), ("0", "1"), ("0", "0")):
        path = str(tmp_path / f"ws{ws}{lean}.pt")
        env = dict(os.environ, AVF_TUNING="1", AVF_NT_WS=ws, AVF_NT_LEAN=lean, AVF_TEST_OUT=path)  # (switches exist under AVF_TUNING=1 only)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(torch.load(path))
    on = outs[0]
    for off in outs[1:]:
        assert on.keys() == off.keys()
        for k in on:
            if k.endswith("net.0.bias"):
                assert torch.allclose(on[k], off[k], rtol=1e-4, atol=1e-6), k
            elif k == "y_after_step":  # (Adam turns the last-bit difference of d net.0.bias into a different update of that bias)
                assert (on[k] - off[k]).norm() <= 1e-3 * off[k].norm(), k
            else:
                assert torch.equal(on[k], off[k]), k


def test_weight_warmup_of_the_tiled_kernel_changes_no_bit(tmp_path):
    """AVF_NT_WPF (gemm_bf16.hip): the tiled NT kernel's workgroups request the whole weight image up front in launches that
    leave workgroup slots empty (default), never (0) or in every launch (2).  The loads feed nothing: three child processes
    run the same GEMMs - both 8-wave tiles, partly filled and full grids, ragged M / N, a one-K-step problem, every lean
    epilogue, the 2-layer stack at C2's token count - and every output must be bit-identical."""
    import subprocess
    import sys
    code = (
        "import os, torch, avformer_amd as A\n"
        "ops = A.ops\n"
        "torch.manual_seed(11)\n"
        "d = {}\n"
        "for i, (M, N, K) in enumerate(((10368, 512, 1024), (16384, 512, 1536), (4099, 520, 64), (2592, 384, 512), (20000, 1024, 192))):\n"
        "    a = torch.randn(M, K, device='cuda').bfloat16(); w = (torch.randn(N, K, device='cuda') / K ** 0.5).bfloat16()\n"
        "    bias = torch.randn(N, device='cuda'); res = torch.randn(M, N, device='cuda').bfloat16()\n"
        "    d[f'plain{i}'] = ops.gemm(a, w, out_dtype=torch.bfloat16).float().cpu()\n"
        "    d[f'res{i}'] = ops.gemm(a, w, out_dtype=torch.bfloat16, epilogue=A._lib.EPI_BIAS_RES, bias=bias, residual=res).float().cpu()\n"
        "    d[f'f32{i}'] = ops.gemm(a, w, out_dtype=torch.float32).cpu()\n"
        "m = A.Transformer(512, 2, 8, 64, 1024, 0.0, compute_dtype='bf16', residual_dtype='bf16').cuda().train()\n"
        "x = torch.randn(8, 324, 512, device='cuda', requires_grad=True)\n"
        "y = m(x); y.float().pow(2).mean().backward()\n"
        "d['y'] = y.detach().float().cpu(); d['dx'] = x.grad.cpu()\n"
        "d.update({n: p.grad.clone().cpu() for n, p in m.named_parameters()})\n"
        "torch.save(d, os.environ['AVF_TEST_OUT'])\n")
    outs = []
    for wpf in ("0", "1", "2"):
        path = str(tmp_path / f"wpf{wpf}.pt")
        env = dict(os.environ, AVF_TUNING="1", AVF_NT_WPF=wpf, AVF_NT_WS="0", AVF_TEST_OUT=path)  # (every NT GEMM on the tiled kernel)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(torch.load(path))
    for other in outs[1:]:
        assert outs[0].keys() == other.keys()
        for k in outs[0]:
            assert torch.equal(outs[0][k], other[k]), k


# ---------------------------------------------------------------------------------------------- against the ORACLE, >= 2048 rows
# The B = 2 oracle tests of test_gpu_configs.py have 1024 token rows and therefore run the tiled kernel; these hold the
# PERSISTENT kernel itself to the oracle's math (oracle/reference_math.py = /root/reference/models/heads.py:164-256), so that a
# divergence of the two kernels cannot hide behind their bit-identity tests.
def test_stack_on_the_persistent_kernel_vs_oracle_2560_rows():
    """d = 512, 2 layers, B = 5 x 512 tokens = 2560 rows: QKV (plain), MLP1 (bias + GELU) and dGELU (+ column sums) run on the
    persistent kernel (d_o - plain, N = 512 - only from three row tiles per persistent workgroup up, i.e. 12288 rows; its epilogue
    form is held to the oracle by the 4128-row test below); forward and every gradient against the oracle's autograd, both
    residual streams"""
    import oracle
    from gpu_util import check_rel, hip_transformer_run, oracle_transformer_run
    D, L, H, dh, M, B, N = 512, 2, 8, 64, 1024, 5, 512
    g = torch.Generator().manual_seed(2560)
    sd = oracle.init_transformer_state(D, L, H, dh, M, generator=g)
    for k in sd:
        if k.endswith("norm.weight"):
            sd[k] = 1 + 0.1 * torch.randn(D, generator=g)
        if k.endswith("norm.bias"):
            sd[k] = 0.1 * torch.randn(D, generator=g)
    x = torch.randn(B, N, D, generator=g)
    loss = lambda y: y.float().pow(2).mean()
    y_ref, dx_ref, g_ref = oracle_transformer_run(x, sd, L, H, loss)
    for resid, caps in (("f32", (1.5e-2, 3e-2, 4e-2)), ("bf16", (3e-2, 5e-2, 6e-2))):
        t = A.Transformer(D, L, H, dh, M, 0.0, compute_dtype="bf16", residual_dtype=resid)
        t.load_state_dict(sd, strict=True)
        t = t.cuda()
        assert ops.gemm_ws_used(B * N, 3 * H * dh, D), "this shape is expected on the persistent kernel"
        y, dx, grads = hip_transformer_run(t, x, loss)
        check_rel(f"ws_oracle[{resid}]:y", y, y_ref, caps[0])
        check_rel(f"ws_oracle[{resid}]:dx", dx, dx_ref, caps[1])
        for k, v in g_ref.items():
            check_rel(f"ws_oracle[{resid}]:g.{k}", grads[k], v, caps[2])


@pytest.mark.parametrize("epi", ["none", "bias_res", "bias_gelu", "dgelu"])
def test_each_persistent_epilogue_vs_oracle_math(epi):
    """one GEMM per epilogue on the persistent kernel (forced: avf_gemm_nt_ws), 4128 rows (ragged last tile), against the
    oracle's own functions evaluated in fp32 on the CPU from the SAME bf16-rounded operands: what is left is the fp32
    accumulation order and the bf16 rounding of the stored result (2^-8 relative)"""
    import oracle
    import torch.nn.functional as F
    from gpu_util import check_rel
    E = {"none": ops.EPI_NONE, "bias_res": ops.EPI_BIAS_RES, "bias_gelu": ops.EPI_BIAS_GELU, "dgelu": ops.EPI_DGELU}[epi]
    M, N = 4128, 1024 if epi in ("bias_gelu", "dgelu") else 512
    a, w, bias, res, aux = _inputs(M, N, E, bf, seed=11)
    out = ops.gemm_ws(a, ops.pack_ws(w), N, out_dtype=bf, epilogue=E, bias=bias, residual=res, aux=aux,
                      want_colsum=(epi == "dgelu"))
    af, wf = a.float().cpu(), w.float().cpu()
    lin = F.linear(af, wf, None if bias is None else bias.cpu())  # nn.Linear, heads.py:191,195,212,215
    if epi == "none":
        check_rel("ws_epi_oracle:none", out, lin, 4e-3)
    elif epi == "bias_res":  # Residual(fn)(x) = fn(x) + x, heads.py:175
        check_rel("ws_epi_oracle:bias_res", out, lin + res.float().cpu(), 4e-3)
    elif epi == "bias_gelu":  # GELU of heads.py:166 (the oracle's 9-op restatement); the saved pre-activation is u itself
        c, u = out
        check_rel("ws_epi_oracle:bias_gelu:u", u, lin, 4e-3)
        check_rel("ws_epi_oracle:bias_gelu:g", c, oracle.gelu_tanh(lin), 6e-3)
    else:  # autograd of that GELU: d/du, times the incoming gradient (the GEMM's product)
        c, cs = out
        u = aux.float().cpu().requires_grad_(True)
        oracle.gelu_tanh(u).backward(lin)
        check_rel("ws_epi_oracle:dgelu", c, u.grad, 6e-3)
        check_rel("ws_epi_oracle:dgelu:colsum", cs, u.grad.sum(0), 1e-3)  # (summed in fp32 before the rounding of the stored values)
