"""-m gpu: dropout (nn.Dropout after to_out, after GELU, after net.3 - reference heads.py:194-196,216).

The kernels draw their masks from a counter-based hash of (seed, layer, site, element), so torch's RNG stream cannot
be bit-matched; instead (1) the mask statistics are checked, (2) the exact masks are extracted
through the C ABI and REPLAYED through the CPU oracle: forward, input gradient and every parameter gradient of the
HIP path must agree with the oracle run under the same masks (this also proves forward and backward regenerate the
same masks), (3) eval() is the identity, as nn.Dropout."""
import pytest
import torch

import oracle
from gpu_util import DEV, check_rel, make_hip_transformer, rel_fro

pytestmark = pytest.mark.gpu


def test_mask_statistics():
    import avformer_amd as A
    p = 0.2
    f = A.ops.dropout_factors(seed=12345, layer=1, site=2, p=p, rows=2048, cols=512)
    vals = torch.unique(f).cpu()
    assert vals.numel() == 2 and vals[0] == 0
    scale = vals[1].item()
    keep = (f > 0).float().mean().item()
    assert abs(keep - (1 - p)) < 3e-3               # 1M draws: sigma = 4e-4
    assert abs(scale * keep - 1.0) < 3e-3           # unbiased: E[factor] = 1
    # per-row and per-column rates are flat too (no striping from the 4-element grouping)
    assert ((f > 0).float().mean(0) - (1 - p)).abs().max() < 0.05
    assert ((f > 0).float().mean(1) - (1 - p)).abs().max() < 0.08
    # different site / layer / seed -> different masks, same key -> same mask
    g = A.ops.dropout_factors(seed=12345, layer=1, site=1, p=p, rows=2048, cols=512)
    h = A.ops.dropout_factors(seed=12346, layer=1, site=2, p=p, rows=2048, cols=512)
    again = A.ops.dropout_factors(seed=12345, layer=1, site=2, p=p, rows=2048, cols=512)
    assert torch.equal(f, again)
    for other in (g, h):
        agree = ((f > 0) == (other > 0)).float().mean().item()
        assert abs(agree - (p * p + (1 - p) ** 2)) < 5e-3   # independent masks


@pytest.mark.parametrize("cfg", [(3, 77, 128, 2, 8, 32, 256), (2, 324, 512, 2, 8, 64, 1024),
                                 (4, 12, 128, 2, 8, 32, 256),       # takes the single-launch forward
                                 (8, 324, 512, 2, 8, 64, 1024),     # 2592 rows: the persistent GEMM + the lean epilogue's
                                 (8, 324, 512, 2, 8, 64, 1024, "bf16"),   # ... dropout site (DROP), both residual streams
                                 (3, 200, 256, 4, 4, 64, 512, "bf16")])   # 4 layers: MIDDLE layers both receive and hand on the two
                                                                          # bf16 gradient images (stream + masked) of round 5
def test_mask_replay_against_oracle(cfg):
    import avformer_amd as A
    B, N, D, L, H, dh, M = cfg[:7]
    resid = cfg[7] if len(cfg) > 7 else "f32"
    p = 0.25
    g = torch.Generator().manual_seed(5)
    sd = oracle.init_transformer_state(D, L, H, dh, M, generator=g)
    x = torch.randn(B, N, D, generator=g)
    t = A.Transformer(D, L, H, dh, M, dropout=p, compute_dtype="bf16", residual_dtype=resid)
    if B * N >= 2048 and D == 512:
        assert A.ops.gemm_ws_used(B * N, M, D, A.ops.EPI_BIAS_GELU), "expected on the persistent kernel"
    t.load_state_dict(sd)
    t = t.to(DEV).train()
    xg = x.to(DEV).requires_grad_(True)
    y = t(xg)
    y.pow(2).mean().backward()
    seed = t.last_seed
    assert seed != 0
    R = B * N
    drop = [tuple(A.ops.dropout_factors(seed, l, s, p, R, cols).cpu().view(B, N, cols)
                  for s, cols in ((0, D), (1, M), (2, D))) for l in range(L)]
    xr = x.clone().requires_grad_(True)
    pr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    yr = oracle.transformer_forward(xr, pr, L, H, drop=drop)
    yr.pow(2).mean().backward()
    # masked activations differ from the unmasked forward by O(1); agreement at bf16 level proves the same masks
    tag = "dropout_replay[" + "x".join(map(str, cfg)) + "]"
    caps = (1.5e-2, 3e-2, 4e-2) if resid == "f32" else (3e-2, 5e-2, 6e-2)
    check_rel(tag + ":y", y, yr, caps[0])
    check_rel(tag + ":dx", xg.grad, xr.grad, caps[1])
    for k, prm in t.named_parameters():
        check_rel(f"{tag}:g.{k}", prm.grad, pr[k].grad, caps[2])
    y_nodrop = oracle.transformer_forward(x, sd, L, H)
    assert rel_fro(yr, y_nodrop) > 0.1               # the masks really changed the result
    # a second forward draws a new seed -> different output; eval() -> deterministic, equals the p=0 math
    y2 = t(xg)
    assert t.last_seed != seed and rel_fro(y2, y) > 0.05
    t.eval()
    with torch.no_grad():
        ye = t(xg)
    check_rel(tag + ":eval_y", ye, y_nodrop, caps[0])


@pytest.mark.parametrize("cfg", [(3, 77, 128, 2, 8, 32, 256), (2, 100, 64, 3, 2, 16, 96), (4, 12, 128, 2, 8, 32, 256),
                                 (2, 324, 256, 2, 4, 64, 512)])  # round 6: dim_head 64, 648 rows - the three-product GEMM and attention kernels
def test_mask_replay_against_oracle_fp32_mode(cfg):
    """the fp32 parity mode with live dropout (round 2): the same counter-based masks ride in the fp32 GEMM epilogues and in
    fp32 masked copies of the two gradients a Linear behind a dropout site sees; replayed through the oracle the whole
    forward / backward must agree at fp32 level - a far tighter check of the mask bookkeeping (which site masks which
    gradient, bias gradients of MASKED sums, the residual stream never masked) than the bf16 replay above."""
    import avformer_amd as A
    B, N, D, L, H, dh, M = cfg
    p = 0.3
    g = torch.Generator().manual_seed(6)
    sd = oracle.init_transformer_state(D, L, H, dh, M, generator=g)
    x = torch.randn(B, N, D, generator=g)
    t = A.Transformer(D, L, H, dh, M, dropout=p, compute_dtype="f32")
    t.load_state_dict(sd)
    t = t.to(DEV).train()
    xg = x.to(DEV).requires_grad_(True)
    y = t(xg)
    y.pow(2).mean().backward()
    seed = t.last_seed
    R = B * N
    drop = [tuple(A.ops.dropout_factors(seed, l, s, p, R, cols).cpu().view(B, N, cols)
                  for s, cols in ((0, D), (1, M), (2, D))) for l in range(L)]
    xr = x.clone().requires_grad_(True)
    pr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    yr = oracle.transformer_forward(xr, pr, L, H, drop=drop)
    yr.pow(2).mean().backward()
    torch.testing.assert_close(y.cpu(), yr.detach(), atol=5e-5, rtol=1e-3)
    torch.testing.assert_close(xg.grad.cpu(), xr.grad, atol=1e-6, rtol=2e-3)
    for k, prm in t.named_parameters():
        torch.testing.assert_close(prm.grad.cpu(), pr[k].grad, atol=2e-6, rtol=2e-3, msg=lambda m, k=k: f"{k}: {m}")
    assert rel_fro(yr, oracle.transformer_forward(x, sd, L, H)) > 0.1  # the masks really changed the result
    t.eval()
    with torch.no_grad():
        torch.testing.assert_close(t(xg).cpu(), oracle.transformer_forward(x, sd, L, H), atol=5e-5, rtol=1e-3)


def test_avformer_model_trains_with_its_reference_dropout():
    """the reference instantiates its heads with dropout=0.2 (avformer.py:48,87): train() must run on the HIP path"""
    import avformer_amd as A
    torch.manual_seed(0)
    model = A.build_model("avformer", task="AU").to(DEV).train()
    x = {"clip": torch.randn(8, 512, device=DEV), "audio_features": torch.randn(8, 512, device=DEV)}
    labels = (torch.rand(8, 12, device=DEV) > 0.5).float()
    loss = model.get_au_loss(model(x), labels)
    loss.backward()
    assert torch.isfinite(loss)
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)


def test_dropout_p_below_mask_resolution_is_refused():
    """0 < p < 2^-17 rounds to a dead mask (thresh16 == 0) while still selecting the live-dropout buffers: the library refuses
    the configuration instead of running with the two predicates in disagreement"""
    import avformer_amd as A
    t = A.Transformer(128, 1, 8, 32, 256, dropout=1e-6).cuda().train()
    with pytest.raises(RuntimeError, match="below the mask resolution"):
        t(torch.randn(2, 12, 128, device="cuda"))
