"""FusedAdam (avf_layer_adam_step) against torch.optim.Adam on the same model: parameters, optimizer state and the
bf16 weight copies the next forward uses."""
import copy

import pytest
import torch

from gpu_util import DEV, rel_fro

pytestmark = pytest.mark.gpu


def _models(mode, dims=(64, 2, 2, 32, 96, 9, 7)):
    import avformer_amd as A
    D, L, H, dh, M, Tv, Ta = dims
    torch.manual_seed(3)
    a = A.SyntheticAVFormer(D, L, H, dh, M, Tv, Ta, compute_dtype=mode).to(DEV)
    b = copy.deepcopy(a)
    return A, a, b, (Tv, Ta, D)


@pytest.mark.parametrize("mode,dims", [("f32", (64, 2, 2, 32, 96, 9, 7)), ("bf16", (64, 2, 2, 32, 96, 9, 7)),
                                       ("bf16", (40, 1, 1, 32, 56, 5, 4)),   # partial 64x64 tiles
                                       ("f32", (36, 1, 3, 16, 52, 5, 4))])   # widths 36 / 52: partial tiles, 4-element granularity
def test_fused_adam_matches_torch_adam(mode, dims):
    A, ma, mb, (Tv, Ta, D) = _models(mode, dims)
    kw = dict(lr=3e-3, betas=(0.9, 0.98), eps=1e-8, weight_decay=1e-2)
    oa = A.optim.FusedAdam(ma, **kw)
    ob = torch.optim.Adam(mb.parameters(), **kw)
    g = torch.Generator().manual_seed(4)
    for it in range(4):
        batch = {"clip": torch.randn(3, Tv, D, generator=g).to(DEV), "audio_features": torch.randn(3, Ta, D, generator=g).to(DEV)}
        labels = (torch.rand(3, 12, generator=g) > 0.5).float().to(DEV)
        for m, o in ((ma, oa), (mb, ob)):
            o.zero_grad(set_to_none=True)
            m.get_au_loss(m(batch), labels).backward()
        # same gradients into both optimizers (removes the bf16 forward noise from the comparison)
        for pa, pb in zip(ma.parameters(), mb.parameters()):
            pb.grad = pa.grad.detach().clone()
        oa.step()
        ob.step()
        for (n, pa), pb in zip(ma.named_parameters(), mb.parameters()):
            assert rel_fro(pa, pb) < 2e-6, (it, n, rel_fro(pa, pb))
    # optimizer state has torch's keys and values
    for pa, pb in zip(ma.parameters(), mb.parameters()):
        sa, sb = oa.state[pa], ob.state[pb]
        assert float(sa["step"]) == float(sb["step"]) == 4.0
        assert rel_fro(sa["exp_avg"], sb["exp_avg"]) < 1e-5
        assert rel_fro(sa["exp_avg_sq"], sb["exp_avg_sq"]) < 1e-5
    if mode == "bf16":
        # the copies written by the optimizer are exactly what a fresh preparation pass produces
        st = ma.transformer
        assert st._lowp_ready
        kept = [b.clone() for b in st._lowp_bufs]
        st.refresh_weights()
        with torch.no_grad():
            ma(batch)
        for x, y in zip(kept, st._lowp_bufs):
            assert torch.equal(x, y)


def test_fused_adam_keeps_the_fragment_major_images_in_sync():
    """dim = 512: the Adam kernel's scattered stores (pack_ws_off, optim.hip) are the only writer of the fragment-major weight
    images the persistent GEMM reads (gemm_ws.hip) after the first step.  After a few steps every byte of a layer's low-precision
    buffer - row-major, transposed AND fragment-major images - must equal what a fresh avf_layer_prepare_weights derives from
    the fp32 masters, and each fragment-major image must be pack_ws() of a bf16 rounding of its master."""
    A, ma, _, (Tv, Ta, D) = _models("bf16", (512, 2, 8, 64, 1024, 9, 7))
    opt = A.optim.FusedAdam(ma, lr=3e-3, betas=(0.9, 0.98), eps=1e-8, weight_decay=1e-2)
    g = torch.Generator().manual_seed(5)
    for _ in range(3):
        batch = {"clip": torch.randn(3, Tv, D, generator=g).to(DEV), "audio_features": torch.randn(3, Ta, D, generator=g).to(DEV)}
        labels = (torch.rand(3, 12, generator=g) > 0.5).float().to(DEV)
        opt.zero_grad(set_to_none=True)
        ma.get_au_loss(ma(batch), labels).backward()
        opt.step()
    st = ma.transformer
    assert st._lowp_ready
    kept = [b.clone() for b in st._lowp_bufs]
    st.refresh_weights()
    with torch.no_grad():
        ma(batch)
    for x, y in zip(kept, st._lowp_bufs):
        assert torch.equal(x, y)
    # the fragment-major image of W1 (net.0.weight [1024, 512]) is somewhere in the buffer, byte for byte
    w1 = dict(st.named_parameters())["layers.0.1.fn.fn.net.0.weight"].detach()
    img = A.ops.pack_ws(w1.to(torch.bfloat16).contiguous()).view(torch.uint8)
    buf = kept[0].view(torch.uint8)
    n = img.numel()
    head = img[:64]
    cand = (buf[: buf.numel() - n + 1].unfold(0, 64, 256) == head).all(1).nonzero().flatten() * 256  # (images are 256-byte aligned)
    assert any(torch.equal(buf[int(o): int(o) + n], img) for o in cand.tolist()), "no fragment-major image of W1 in the low-precision buffer"


def test_fused_adam_skips_weight_prep_and_state_dict_roundtrip():
    A, ma, mb, (Tv, Ta, D) = _models("bf16")
    oa = A.optim.FusedAdam(ma, lr=1e-3, weight_decay=5e-5)
    g = torch.Generator().manual_seed(5)
    batch = {"clip": torch.randn(2, Tv, D, generator=g).to(DEV), "audio_features": torch.randn(2, Ta, D, generator=g).to(DEV)}
    labels = (torch.rand(2, 12, generator=g) > 0.5).float().to(DEV)

    def one(m, o):
        o.zero_grad(set_to_none=True)
        loss = m.get_au_loss(m(batch), labels)
        loss.backward()
        o.step()
        return float(loss.detach())

    l0 = one(ma, oa)
    l1 = one(ma, oa)  # this forward consumed the optimizer-written copies
    assert l1 < l0
    # resume: a second optimizer loaded from the state dict continues identically
    mb.load_state_dict(ma.state_dict())
    ob = A.optim.FusedAdam(mb, lr=1e-3, weight_decay=5e-5)
    ob.load_state_dict(copy.deepcopy(oa.state_dict()))
    la, lb = one(ma, oa), one(mb, ob)
    assert abs(la - lb) < 1e-6 * max(1.0, abs(la))
    for pa, pb in zip(ma.parameters(), mb.parameters()):
        assert rel_fro(pa, pb) < 1e-6


def test_load_state_dict_after_step_invalidates_bf16_copies():
    """optimizer step -> load_state_dict -> forward must use the LOADED weights, not the copies the step wrote"""
    A, ma, mb, (Tv, Ta, D) = _models("bf16")
    oa = A.optim.FusedAdam(ma, lr=1e-2)
    g = torch.Generator().manual_seed(8)
    batch = {"clip": torch.randn(2, Tv, D, generator=g).to(DEV), "audio_features": torch.randn(2, Ta, D, generator=g).to(DEV)}
    labels = (torch.rand(2, 12, generator=g) > 0.5).float().to(DEV)
    saved = copy.deepcopy(ma.state_dict())
    ma.get_au_loss(ma(batch), labels).backward()
    oa.step()                      # weights move; bf16 copies rewritten, marked ready
    ma.load_state_dict(saved)      # back to the initial weights, in place (same storage)
    with torch.no_grad():
        out_a = ma(batch)
        out_b = mb(batch)          # mb still holds the initial weights
    assert torch.equal(out_a, out_b)


def test_training_trajectory_bf16_library_adam_tracks_f32_torch_adam():
    """40 optimisation steps on a fixed batch: the throughput path (bf16 kernels, pre-scaled q, bf16 weight copies written by
    the library's Adam and consumed by the next forward) must follow the parity path (f32 kernels, torch.optim.Adam) -
    a stale or mis-scaled weight copy anywhere in that loop shows up as a diverging loss curve."""
    import avformer_amd as A
    D, L, H, dh, M, Tv, Ta = 64, 2, 2, 32, 128, 20, 12
    torch.manual_seed(21)
    m32 = A.SyntheticAVFormer(D, L, H, dh, M, Tv, Ta, compute_dtype="f32").to(DEV)
    m16 = A.SyntheticAVFormer(D, L, H, dh, M, Tv, Ta, compute_dtype="bf16").to(DEV)
    m16.load_state_dict(m32.state_dict())
    o32 = torch.optim.Adam(m32.parameters(), lr=2e-3, weight_decay=5e-5)
    o16 = A.optim.FusedAdam(m16, lr=2e-3, weight_decay=5e-5)
    g = torch.Generator().manual_seed(22)
    batch = {"clip": torch.randn(16, Tv, D, generator=g).to(DEV), "audio_features": torch.randn(16, Ta, D, generator=g).to(DEV)}
    labels = (torch.rand(16, 12, generator=g) > 0.5).float().to(DEV)
    curves = {}
    for name, m, o in (("f32", m32, o32), ("bf16", m16, o16)):
        losses = []
        for _ in range(40):
            o.zero_grad(set_to_none=True)
            loss = m.get_au_loss(m(batch), labels)
            loss.backward()
            o.step()
            losses.append(float(loss.detach()))
        curves[name] = losses
    a, b = curves["f32"], curves["bf16"]
    assert a[-1] < 0.6 * a[0] and b[-1] < 0.6 * b[0], (a[0], a[-1], b[0], b[-1])  # both fit the batch
    for i in (0, 5, 10, 20, 39):
        assert abs(a[i] - b[i]) < 0.05 * a[i] + 5e-3, (i, a[i], b[i])


def test_frozen_stack_and_partly_frozen_layer_state_dict_roundtrip():
    """the reference's default setup: pretrained branches frozen (avformer.py:76-85).  A fully frozen stack gets no
    optimizer state and no launch; frozen tensors of a partly trainable stack stay out of ``state`` - so
    ``state_dict()`` / ``load_state_dict()`` work (KeyError in torch's param_mappings before)."""
    import avformer_amd as A
    torch.manual_seed(0)
    m = A.build_model("avformer", task="AU", compute_dtype="bf16").to(DEV).train()
    for p in m.audio_model.parameters():     # whole branch (one stack) frozen
        p.requires_grad = False
    vt = m.video_model.au_head.corr_transformer
    for p in vt.layer_parameters(0):         # first layer of the video stack frozen, second trainable
        p.requires_grad = False
    frozen_before = {n: p.detach().clone() for n, p in m.named_parameters() if not p.requires_grad}
    opt = A.optim.FusedAdam(m, lr=1e-2, weight_decay=1e-2)
    g = torch.Generator().manual_seed(1)
    batch = {"clip": torch.randn(8, 512, generator=g).to(DEV), "audio_features": torch.randn(8, 512, generator=g).to(DEV)}
    labels = (torch.rand(8, 12, generator=g) > 0.5).float().to(DEV)
    for _ in range(2):
        opt.zero_grad(set_to_none=True)
        m.get_au_loss(m(batch), labels).backward()
        opt.step()
    torch.cuda.synchronize()
    for n, p in m.named_parameters():
        if not p.requires_grad:
            assert torch.equal(p, frozen_before[n]), f"frozen tensor {n} was updated"
            assert p not in opt.state, f"optimizer state created for the frozen tensor {n}"
    sd = opt.state_dict()                     # raised KeyError before the fix
    # (the per-token logit weights of the two branch heads get no gradient: avformer.py:100 uses their tokens only)
    assert len(sd["state"]) == sum(1 for p in m.parameters() if p.requires_grad and p.grad is not None)
    opt2 = A.optim.FusedAdam(m, lr=1e-2, weight_decay=1e-2)
    opt2.load_state_dict(sd)
    opt2.zero_grad(set_to_none=True)
    m.get_au_loss(m(batch), labels).backward()
    opt2.step()
    torch.cuda.synchronize()
    assert all(torch.isfinite(p).all() for p in m.parameters())
    # the frozen audio stack caches its bf16 images across forwards (nothing retrains it); an in-place edit is noticed
    at = m.audio_model.au_head.corr_transformer
    kept = at._lowp_versions
    with torch.no_grad():
        m(batch)
    assert at._lowp_versions is kept
    with torch.no_grad():
        at.layer_parameters(0)[2].mul_(1.5)   # version counter moves -> images are re-derived
        before = m(batch)
        at.layer_parameters(0)[2].div_(1.5)
        after = m(batch)
    assert not torch.equal(before, after)


def test_inplace_weight_edit_after_fused_step_is_not_ignored():
    """after FusedAdam.step() the next forward skips the weight preparation; an in-place edit of a master in
    between (EMA swap, clipping) must void that skip"""
    A, ma, _, (Tv, Ta, D) = _models("bf16")
    opt = A.optim.FusedAdam(ma, lr=1e-3)
    g = torch.Generator().manual_seed(4)
    batch = {"clip": torch.randn(3, Tv, D, generator=g).to(DEV), "audio_features": torch.randn(3, Ta, D, generator=g).to(DEV)}
    labels = (torch.rand(3, 12, generator=g) > 0.5).float().to(DEV)
    ma.get_au_loss(ma(batch), labels).backward()
    opt.step()
    w = ma.transformer.layer_parameters(0)[7]   # net.0.weight
    with torch.no_grad():
        w.mul_(0.0)
        y_zeroed = ma(batch)
    ma.transformer.refresh_weights()
    with torch.no_grad():
        y_fresh = ma(batch)
    assert torch.equal(y_zeroed, y_fresh)


def test_adam_descriptor_table_selftest_and_abort():
    """the ~12.6 KB by-value descriptor table reaches the kernel whole (143 one-element tensors, each checked), and an aborted
    session launches nothing: avf_adam_batch_abort leaves parameters untouched and a later session works"""
    import ctypes as C
    import avformer_amd as A
    from avformer_amd import _lib
    lib = _lib.load()
    _lib.check(lib.avf_selftest_adam_table(C.c_void_p(torch.cuda.current_stream().cuda_stream)), "selftest_adam_table")
    torch.manual_seed(0)
    t = A.Transformer(128, 1, 8, 32, 256).cuda()
    opt = A.optim.FusedAdam(t, lr=1e-2)
    x = torch.randn(2, 12, 128, device="cuda")
    t(x).float().pow(2).mean().backward()
    before = [p.detach().clone() for p in t.parameters()]
    step_before = None if opt._step_dev is None else float(opt._step_dev)
    orig = opt._step_body
    def boom(*a, **k):
        orig(*a, **k)              # the table has been collected ...
        raise RuntimeError("boom")  # ... and the caller fails before _end
    opt._step_body = boom
    with pytest.raises(RuntimeError, match="boom"):
        opt.step()
    torch.cuda.synchronize()
    assert all(torch.equal(a, p.detach()) for a, p in zip(before, t.parameters())), "an aborted session launched its table"
    assert float(opt._step_dev) == (0.0 if step_before is None else step_before)
    opt._step_body = orig
    opt.step()                     # the session machinery is usable again
    torch.cuda.synchronize()
    assert any(not torch.equal(a, p.detach()) for a, p in zip(before, t.parameters()))
