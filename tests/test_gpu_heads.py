"""-m gpu: the token producers / consumers either side of the stack (csrc/heads.hip) against plain
fp32 restatements: BatchNorm1d (training + eval, running statistics), the per-token dots, cls / positional assembly, the
feature-axis fusion, the feature-map <-> token permutes, the padded logits Linear; then AU_former in TRAINING mode (batch
statistics) against autograd through the oracle - the fixtures G5 / G6 / G7 / G9 / G11 (tests/test_gpu_transformer.py) hold
the same code in eval mode to the reference's own outputs."""
import copy

import pytest
import torch

import oracle
from gpu_util import DEV, rel_fro

pytestmark = pytest.mark.gpu


def _close(a, b, atol=2e-6, rtol=1e-5):
    torch.testing.assert_close(a.detach().float().cpu(), b.detach().float().cpu(), atol=atol, rtol=rtol)


@pytest.fixture(scope="module")
def ops():
    import avformer_amd as A
    return A.ops


@pytest.mark.parametrize("B,C", [(5, 7), (64, 512), (2, 300), (1, 16)])
@pytest.mark.parametrize("training", [True, False])
def test_bn1d_fwd_bwd(ops, B, C, training):
    if B == 1 and training:
        pytest.skip("nn.BatchNorm1d refuses a single training row")
    g = torch.Generator().manual_seed(B * 10 + C)
    bn = torch.nn.BatchNorm1d(C)
    with torch.no_grad():
        bn.weight.copy_(torch.randn(C, generator=g))
        bn.bias.copy_(torch.randn(C, generator=g))
        bn.running_mean.copy_(torch.randn(C, generator=g))
        bn.running_var.copy_(torch.rand(C, generator=g) + 0.5)
    bn.train(training)
    ref = copy.deepcopy(bn)
    x = (torch.randn(B, C, generator=g) * 2 + 1).requires_grad_(True)
    dy = torch.randn(B, C, generator=g)
    y_ref = ref(x)
    y_ref.backward(dy)
    d = bn.to(DEV)
    y, mean, invstd = ops.bn1d_fwd(x.detach().to(DEV), d.weight.detach(), d.bias.detach(), d.running_mean, d.running_var,
                                   d.num_batches_tracked if training else None, d.eps, d.momentum, training)
    _close(y, y_ref, atol=1e-5, rtol=1e-5)
    _close(d.running_mean, ref.running_mean, atol=1e-6)
    _close(d.running_var, ref.running_var, atol=1e-6, rtol=1e-5)
    assert int(d.num_batches_tracked) == int(ref.num_batches_tracked)
    dx, dg, db = ops.bn1d_bwd(x.detach().to(DEV), dy.to(DEV), d.weight.detach(), mean, invstd, training)
    _close(dx, x.grad, atol=2e-5, rtol=1e-4)
    _close(dg, ref.weight.grad, atol=2e-5, rtol=1e-4)
    _close(db, ref.bias.grad, atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("B,T,E,pad", [(3, 12, 128, None), (64, 12, 256, 21), (2, 5, 33, 8)])
def test_token_dots(ops, B, T, E, pad):
    g = torch.Generator().manual_seed(B + T + E)
    tok = torch.randn(B, T, E, generator=g)
    w = torch.randn(T, E, generator=g)
    out = ops.token_dots_fwd(tok.to(DEV), w.to(DEV), pad)
    ref = (tok * w).sum(-1)
    _close(out[:, :T], ref, atol=1e-5)
    if pad:
        assert out.shape == (B, pad) and float(out[:, T:].abs().sum()) == 0.0
    dout = torch.randn(B, pad or T, generator=g)
    dtok, dw = ops.token_dots_bwd(dout.to(DEV), tok.to(DEV), w.to(DEV))
    _close(dtok, dout[:, :T, None] * w[None], atol=1e-6)
    _close(dw, (dout[:, :T, None] * tok).sum(0), atol=1e-5)


def test_assemble_cat_transpose(ops):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(4, 16, 24, generator=g)
    cls = torch.randn(1, 24, generator=g)
    pos = torch.randn(17, 24, generator=g)
    out = ops.assemble_tokens(x.to(DEV), cls.to(DEV), pos.to(DEV))
    assert torch.equal(out.cpu(), torch.cat([cls.expand(4, 1, 24), x], 1) + pos)
    assert torch.equal(ops.assemble_tokens(x.to(DEV), None, pos[:16].to(DEV)).cpu(), x + pos[:16])
    a, v = torch.randn(5, 12, 128, generator=g), torch.randn(5, 12, 64, generator=g)
    p2 = torch.randn(12, 192, generator=g)
    assert torch.equal(ops.cat_features(a.to(DEV), v.to(DEV), p2.to(DEV)).cpu(), torch.cat([a, v], 2) + p2)
    m = torch.randn(3, 70, 49, generator=g)   # [B, C, S]: ragged 32 x 32 tiles
    p3 = torch.randn(49, 70, generator=g)
    assert torch.equal(ops.transpose_add(m.to(DEV), p3.to(DEV)).cpu(), m.permute(0, 2, 1) + p3)
    assert torch.equal(ops.transpose_add(m.to(DEV), None).cpu(), m.permute(0, 2, 1).contiguous())


def test_small_gemm_tiles_and_strided_operands(ops):
    """the 32 x 32 configuration of the parity GEMM (picked for small grids), a broadcast residual row (ld 0), a strided
    output view and strided A / B operands"""
    g = torch.Generator().manual_seed(4)
    y = torch.randn(64, 512, generator=g)
    w = torch.randn(1536, 512, generator=g) / 512 ** 0.5
    b = torch.randn(1536, generator=g)
    pos = torch.randn(1536, generator=g)
    c = ops.gemm(y.to(DEV), w.to(DEV), epilogue=ops.EPI_BIAS_RES, bias=b.to(DEV), residual=pos.to(DEV), residual_ld=0)
    _close(c, (y.double() @ w.double().t() + b.double() + pos.double()).float(), atol=2e-5, rtol=1e-5)
    out = torch.full((64, 21), 7.0, device=DEV)
    w12 = torch.randn(12, 512, generator=g)
    ops.gemm(y.to(DEV), w12.to(DEV), bias=b[:12].to(DEV), out=out[:, :12])
    _close(out[:, :12], (y.double() @ w12.double().t() + b[:12].double()).float(), atol=2e-4, rtol=1e-5)
    assert bool((out[:, 12:] == 7.0).all())
    dl = out[:, :12]   # row stride 21
    _close(ops.gemm(dl, y.to(DEV), trans_a=True, trans_b=False), (dl.cpu().double().t() @ y.double()).float(), atol=2e-3, rtol=1e-5)
    _close(ops.colsum(dl), dl.cpu().double().sum(0).float(), atol=1e-3, rtol=1e-5)


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_au_former_training_mode_vs_oracle(mode):
    """AU_former with BATCH statistics (train mode, dropout 0): logits, tokens, every gradient incl. BatchNorm's against
    autograd through the oracle (fixture G5 holds eval mode to the reference itself)"""
    import avformer_amd as A
    torch.manual_seed(1)
    m = A.AU_former(input_dim=64, emb_dim=128, compute_dtype=mode).to(DEV).train()
    g = torch.Generator().manual_seed(2)
    x = torch.randn(9, 64, generator=g) * 1.5 + 0.3
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    xr = x.clone().requires_grad_(True)
    pr = {k: (v.requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v) for k, v in sd.items()}
    lr, tr = oracle.au_former_forward(xr, pr, training=True)
    (lr.pow(2).mean() + tr.pow(2).mean()).backward()
    xg = x.to(DEV).requires_grad_(True)
    logits, tokens = m(xg)
    (logits.pow(2).mean() + tokens.pow(2).mean()).backward()
    tol = 2e-5 if mode == "f32" else 2e-2
    assert rel_fro(logits, lr) < tol and rel_fro(tokens, tr) < tol, (rel_fro(logits, lr), rel_fro(tokens, tr))
    assert rel_fro(xg.grad, xr.grad) < (1e-4 if mode == "f32" else 4e-2), rel_fro(xg.grad, xr.grad)
    for n, p in m.named_parameters():
        e = rel_fro(p.grad, pr[n].grad)
        assert e < (2e-4 if mode == "f32" else 6e-2), (n, e)
    # running statistics moved as nn.BatchNorm1d moves them
    torch.testing.assert_close(m.AU_BN1.running_mean.cpu(), 0.9 * sd["AU_BN1.running_mean"] + 0.1 * x.mean(0), atol=1e-6, rtol=1e-5)
    torch.testing.assert_close(m.AU_BN1.running_var.cpu(), 0.9 * sd["AU_BN1.running_var"] + 0.1 * x.var(0, unbiased=True),
                               atol=1e-6, rtol=1e-5)
    assert int(m.AU_BN1.num_batches_tracked) == 1


def test_flat_parameter_groups_survive_device_moves_and_state_dict():
    """the 12 projection weights are slices of one buffer (no per-step concatenation); .to() / load_state_dict / deepcopy
    keep the model correct (the group re-packs itself when the storages are no longer adjacent)"""
    import avformer_amd as A
    torch.manual_seed(3)
    m = A.AU_former(input_dim=32, emb_dim=128, compute_dtype="f32").to(DEV).eval()
    x = torch.randn(4, 32, device=DEV)
    y0, t0 = m(x)
    W = m._proj_w.get()
    assert all(p.data_ptr() == W.data_ptr() + i * W[0].numel() * 4 for i, p in enumerate(m._proj_w.params))
    sd = copy.deepcopy(m.state_dict())
    m2 = A.AU_former(input_dim=32, emb_dim=128, compute_dtype="f32")
    m2.load_state_dict(sd)
    m2 = m2.to(DEV).eval()
    y2, _ = m2(x)
    assert torch.equal(y0, y2)
    m3 = copy.deepcopy(m).cpu().to(DEV)      # storages scattered by the round trip
    y3, _ = m3(x)
    assert torch.equal(y0, y3)
    with torch.no_grad():                     # in-place optimizer-style update through the Parameter objects
        for p in m._proj_w.params:
            p.mul_(0.5)
    y4, _ = m(x)
    assert not torch.equal(y4, y0)


@pytest.mark.parametrize("B,K,O,width", [(32, 512, 12, 21), (5, 128, 12, 21), (64, 768, 7, 7), (1, 4096, 3, 8)])
def test_linear_pad_one_launch_forward_backward(B, K, O, width):
    """the AU logits Linear written into the reference's zero-padded [B, 21] row (avformer.py:101-105) as one launch each way"""
    import avformer_amd as A
    g = torch.Generator().manual_seed(B + K)
    x = torch.randn(B, K, generator=g)
    w = torch.randn(O, K, generator=g) / K ** 0.5
    b = torch.randn(O, generator=g)
    dout = torch.randn(B, width, generator=g)
    out = A.ops.linear_pad_fwd(x.cuda(), w.cuda(), b.cuda(), width)
    ref = x.double() @ w.double().t() + b.double()
    torch.testing.assert_close(out[:, :O].cpu().double(), ref, atol=2e-6, rtol=1e-5)
    assert torch.all(out[:, O:] == 0)
    dx, dw, db = A.ops.linear_pad_bwd(dout.cuda(), x.cuda(), w.cuda())
    dl = dout[:, :O].double()
    torch.testing.assert_close(dx.cpu().double(), dl @ w.double(), atol=2e-6, rtol=1e-5)
    torch.testing.assert_close(dw.cpu().double(), dl.t() @ x.double(), atol=5e-6, rtol=1e-5)
    torch.testing.assert_close(db.cpu().double(), dl.sum(0), atol=5e-6, rtol=1e-5)
    dx2, dw2, db2 = A.ops.linear_pad_bwd(dout.cuda(), x.cuda(), w.cuda(), need_dx=False, need_db=False)
    assert dx2 is None and db2 is None and torch.equal(dw2, dw)


@pytest.mark.parametrize("name", ["sformer", "vformer", "tformer"])
def test_former_registry_models_run_a_training_step(name):
    """train.py:292-303 registry entries on the device: [B,21] layout, AU loss, gradients reach the token sections and heads.
    Without a backbone the models take the stage-3 feature map ([B*16, 256, 7, 7]) the reference's ResNet stem would produce."""
    import avformer_amd as A
    torch.manual_seed(0)
    m = A.models.build_model(name, task="AU").cuda()
    B = 2
    frames = B if name == "sformer" else B * 16
    x = {"clip": torch.randn(frames, 256, 7, 7, device="cuda")}
    y = (torch.rand(B, 12, device="cuda") > 0.5).float()
    out = m(x)
    assert out.shape == (B, 21) and torch.isfinite(out).all()
    m.get_au_loss(out, y).backward()
    tok = m.base_model if name == "sformer" else m.video_model.s_former
    assert tok.pos_embedding.grad is not None and torch.isfinite(tok.pos_embedding.grad).all()
    g = [p.grad for n, p in m.named_parameters() if "to_qkv" in n and not n.startswith("va_head.")]  # (VA head: task 'VA' only)
    assert g and all(t is not None and torch.isfinite(t).all() and float(t.abs().sum()) > 0 for t in g)


def test_sformer_va_task_runs_va_former():
    """registry entry ``sformer`` with task='VA' (sformer.py:358, 378-380): the last two columns of the [B,21] row come from
    ``VA_former`` on the frame feature; the CCC loss of get_va_loss reaches its stack and the token section"""
    import avformer_amd as A
    torch.manual_seed(0)
    m = A.models.build_model("sformer", task="VA").cuda()
    B = 4
    x = {"clip": torch.randn(B, 256, 7, 7, device="cuda")}
    out = m(x)
    assert out.shape == (B, 21) and torch.isfinite(out).all()
    m.eval()                                                  # (dropout 0.2 is live in train mode: compare in eval mode)
    out = m(x)
    va, _ = m.va_head(m.base_model(x["clip"]))
    assert torch.equal(out[:, 19:21], va.to(out.dtype))
    m.train()
    y = torch.rand(B, 2, device="cuda") * 2 - 1
    m.get_va_loss(m(x), y).backward()
    g = [p.grad for n, p in m.named_parameters() if n.startswith("va_head.") and "to_qkv" in n]
    assert g and all(t is not None and torch.isfinite(t).all() and float(t.abs().sum()) > 0 for t in g)
    assert m.base_model.pos_embedding.grad is not None
