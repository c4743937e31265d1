#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REFERENCE implementation.

Run in the build container only (it needs /root/reference, which never travels to the GPU box):

    python tests/golden/make_golden.py

It imports the reference's own modules *unmodified* (models/heads.py, models/loss.py standalone;
models/tformer.py and models/vformer.py through a synthetic package whose __path__ is the
reference's models/ directory, with an empty stub registered for the absent ``torchvision``,
which the classes used here never touch) and records inputs, parameters, outputs and gradients
as fp32 ``.npz`` files.  The fixtures are data only; no reference source text is stored.

Fixture list: G1 attention, G2 feed-forward, G3 transformer at config C1,
G4 transformer with inner != dim at N in {12, 17, 49}, G5 AU_former (eval), G6 tformer_AU_head
(emb 64), G7 TFormer, G8 AULoss with/without ignored rows, G9 tiny pipeline TFormer -> AU_former
-> AULoss with gradients, G10 tanh-GELU on a grid, G11 the token section of ResFormer.forward, G12 the evaluation
score of metrics/accf1.py (MultiLabelAccF1, the reference's own sklearn-backed implementation) on seeded batches, G13 the
Transformer with a token mask (the branch heads.py:225-232), G14 the Transformer whose to_out is nn.Identity (heads.py:207),
G15 VA_former (heads.py:341-372) in eval and in train mode.

    python tests/golden/make_golden.py --only g12        (re)generates just that fixture
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
REF = os.environ.get("AVF_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def _load_standalone(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    # torchvision is imported at the top of tformer/vformer but unused by the classes we need
    for n in ("torchvision", "torchvision.models"):
        if n not in sys.modules:
            sys.modules[n] = types.ModuleType(n)
    sys.modules["torchvision"].models = sys.modules["torchvision.models"]
    pkg = types.ModuleType("refmodels")
    pkg.__path__ = [os.path.join(REF, "models")]
    sys.modules["refmodels"] = pkg
    heads = _load_standalone("refmodels.heads", os.path.join(REF, "models", "heads.py"))
    loss = _load_standalone("refmodels.loss", os.path.join(REF, "models", "loss.py"))
    tformer = _load_standalone("refmodels.tformer", os.path.join(REF, "models", "tformer.py"))
    vformer = _load_standalone("refmodels.vformer", os.path.join(REF, "models", "vformer.py"))
    return heads, loss, tformer, vformer


def npify(d):
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **npify(arrays))
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


def module_io(mod, x, loss_fn, prefix_params="p.", prefix_grads="g."):
    """Forward + backward; returns dict with params, grads, y, dx."""
    x = x.clone().requires_grad_(True)
    y = mod(x)
    if isinstance(y, tuple):
        y_main = y[0]
    else:
        y_main = y
    loss = loss_fn(y_main)
    loss.backward()
    out = {"x": x.detach(), "y": y_main.detach(), "dx": x.grad, "loss": loss.detach()}
    if isinstance(y, tuple):
        for i, extra in enumerate(y[1:]):
            out[f"y_extra{i}"] = extra.detach()
    for k, v in mod.state_dict().items():
        out[prefix_params + k] = v
    for k, p in mod.named_parameters():
        if p.grad is not None:
            out[prefix_grads + k] = p.grad
    return out


def main():
    heads, loss_mod, tformer, vformer = load_reference()
    sq = lambda y: y.pow(2).mean()

    # G1: one Attention (dim 32, 4 heads x 8)
    torch.manual_seed(1001)
    att = heads.Attention(dim=32, heads=4, dim_head=8)
    save("g1_attention", heads=4, dim_head=8, **module_io(att, torch.randn(2, 7, 32), sq))

    # G2: one FeedForward (32 -> 64 -> 32)
    torch.manual_seed(1002)
    ff = heads.FeedForward(32, 64)
    save("g2_feedforward", **module_io(ff, torch.randn(2, 7, 32), sq))

    # G3: Transformer at BASELINE config C1 (B4 N64 D128 L2 H8 dh32 M256)
    torch.manual_seed(1003)
    tr = heads.Transformer(128, 2, 8, 32, 256)
    save("g3_transformer_c1", dim=128, depth=2, heads=8, dim_head=32, mlp_dim=256,
         **module_io(tr, torch.randn(4, 64, 128), sq))

    # G4: inner != dim, N in {12, 17, 49}
    for n, (dim, depth, h, dh, mlp) in {12: (64, 2, 8, 32, 128), 17: (32, 2, 8, 64, 64),
                                        49: (48, 1, 8, 32, 96)}.items():
        torch.manual_seed(1004 + n)
        tr = heads.Transformer(dim, depth, h, dh, mlp)
        save(f"g4_transformer_n{n}", dim=dim, depth=depth, heads=h, dim_head=dh, mlp_dim=mlp,
             **module_io(tr, torch.randn(3, n, dim), sq))

    # G5: AU_former(input_dim=64, emb_dim=32) in eval mode (running stats perturbed so BN is not
    # the identity)
    torch.manual_seed(1005)
    auf = heads.AU_former(input_dim=64, emb_dim=32)
    auf.AU_BN1.running_mean.normal_(0, 0.5)
    auf.AU_BN1.running_var.uniform_(0.5, 2.0)
    auf.eval()
    save("g5_au_former", input_dim=64, emb_dim=32, **module_io(auf, torch.randn(5, 64), sq))

    # G6: tformer_AU_head(emb_dim=64) eval  (the stand-in for avformer's missing former_AU_head,
    # which avformer.py:87 instantiates with emb_dim=256; 64 keeps the fixture small)
    torch.manual_seed(1006)
    hd = tformer.tformer_AU_head(emb_dim=64)
    hd.eval()
    save("g6_au_head", emb_dim=64, **module_io(hd, torch.randn(3, 12, 64), sq))

    # G7: TFormer(num_patches=16, dim=64, depth=2, heads=8, mlp 128, dim_head 32)
    torch.manual_seed(1007)
    tf = vformer.TFormer(num_patches=16, dim=64, depth=2, heads=8, mlp_dim=128, dim_head=32)
    save("g7_tformer", num_patches=16, dim=64, depth=2, heads=8, dim_head=32, mlp_dim=128,
         **module_io(tf, torch.randn(3, 16, 64), sq))

    # G8: AULoss with and without ignored (-1) rows.  The ctor calls torch.cuda.current_device()
    # (loss.py:73); on a CPU-only host we let it return 'cpu'.
    torch.cuda.current_device = lambda: "cpu"
    crit = loss_mod.AULoss()
    torch.manual_seed(1008)
    z = torch.randn(16, 12) * 3
    y = (torch.rand(16, 12) > 0.5).float()
    out = {}
    for tag, yy in (("all", y.clone()), ("ign", y.clone())):
        if tag == "ign":
            yy[3] = -1
            yy[7] = -1
            yy[8, 0] = -1          # only the first label decides (loss.py:85-88)
            yy[9, 5] = -1          # a -1 elsewhere does NOT drop the row (its BCE target is -1)
        zz = z.clone().requires_grad_(True)
        l = crit(zz, yy)
        l.backward()
        out.update({f"{tag}.z": z, f"{tag}.y": yy, f"{tag}.loss": l.detach(), f"{tag}.dz": zz.grad})
    save("g8_au_loss", **out)

    # G9: tiny pipeline  [B,T,D] -> TFormer -> AU_former(eval) -> AULoss, all gradients
    torch.manual_seed(1009)
    tf = vformer.TFormer(num_patches=8, dim=32, depth=1, heads=8, mlp_dim=64, dim_head=32)
    auf = heads.AU_former(input_dim=32, emb_dim=32)
    auf.AU_BN1.running_mean.normal_(0, 0.3)
    auf.AU_BN1.running_var.uniform_(0.7, 1.5)
    auf.eval()
    x = torch.randn(6, 8, 32, requires_grad=True)
    labels = (torch.rand(6, 12) > 0.5).float()
    labels[2] = -1
    feat = tf(x)
    logits, tokens = auf(feat)
    l = crit(logits, labels)
    l.backward()
    out = {"x": x.detach(), "labels": labels, "feat": feat.detach(), "logits": logits.detach(),
           "tokens": tokens.detach(), "loss": l.detach(), "dx": x.grad}
    for k, v in tf.state_dict().items():
        out["p.tf." + k] = v
    for k, v in auf.state_dict().items():
        out["p.au." + k] = v
    for k, p in tf.named_parameters():
        out["g.tf." + k] = p.grad
    for k, p in auf.named_parameters():
        if p.grad is not None:
            out["g.au." + k] = p.grad
    save("g9_pipeline", **out)

    # G11: the token section of ResFormer.forward (vformer.py:245-259 = sformer.py:313-327): stage-3 feature map
    # [B', C, h, w] -> tokens [B', h*w, C] + pos_embedding -> spatial_transformer -> back to [B', C, h, w].  The module
    # is the reference's own ResFormer (its conv stages are constructed but not run); small dims keep the fixture small.
    torch.manual_seed(1011)
    rf = vformer.ResFormer(vformer.BasicBlock, [1, 1, 1, 1], num_patches=9, dim=32, depth=1, heads=8, mlp_dim=64, dim_head=32)
    xmap = torch.randn(4, 32, 3, 3, requires_grad=True)
    b_l, c, h, w = xmap.shape
    t = xmap.reshape((b_l, c, h * w)).permute(0, 2, 1)
    t = t + rf.pos_embedding[:, :t.shape[1]]
    t = rf.spatial_transformer(t)
    ymap = t.permute(0, 2, 1).reshape((b_l, c, h, w))
    l = sq(ymap)
    l.backward()
    out = {"x": xmap.detach(), "y": ymap.detach(), "dx": xmap.grad, "loss": l.detach(),
           "p.pos_embedding": rf.pos_embedding.detach(), "g.pos_embedding": rf.pos_embedding.grad}
    for k, v in rf.spatial_transformer.state_dict().items():
        out["p.spatial_transformer." + k] = v
    for k, q in rf.spatial_transformer.named_parameters():
        out["g.spatial_transformer." + k] = q.grad
    save("g11_resformer_tokens", num_patches=9, dim=32, depth=1, heads=8, dim_head=32, mlp_dim=64, **out)

    # G10: tanh-GELU on a grid incl. large magnitudes, with its derivative
    g = heads.GELU()
    u = torch.cat([torch.linspace(-12, 12, 481), torch.tensor([-60.0, -30.0, 30.0, 60.0, 0.0, 1e-4, -1e-4])])
    u = u.clone().requires_grad_(True)
    yv = g(u)
    yv.sum().backward()
    save("g10_gelu", u=u.detach(), y=yv.detach(), dy_du=u.grad)


def metric_fixture():
    """G12: the reference's MultiLabelAccF1 (metrics/accf1.py:47-77; numpy + sklearn only) fed as train.py:155-163 feeds
    it - round(sigmoid(logits)) per batch, labels in {0, 1, -1} - over several seeded cases incl. an AU without a single
    positive label and an all-negative / nothing-predicted case (sklearn's zero_division path)."""
    import warnings
    accf1 = _load_standalone("ref_accf1", os.path.join(REF, "metrics", "accf1.py"))
    rng = np.random.default_rng(12)
    out = {}
    cases = {"mixed": (7, 37), "single_batch": (1, 64), "no_positive_au": (3, 20), "all_negative": (2, 8)}
    for name, (nb, bs) in cases.items():
        m = accf1.MultiLabelAccF1(ignore_index=-1)
        logits = rng.normal(size=(nb, bs, 21)).astype(np.float32)
        y = (rng.random((nb, bs, 12)) > 0.6).astype(np.float32)
        y[rng.random((nb, bs, 12)) < 0.1] = -1
        if name == "no_positive_au":
            y[:, :, 5] = 0
        if name == "all_negative":
            y[:] = 0
            logits[:] = -3.0
        for b in range(nb):
            pred = np.round(torch.sigmoid(torch.from_numpy(logits[b, :, :12])).numpy())  # train.py:155
            m.update(pred, y[b])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            acc, f1 = m.get()
        out[f"{name}.logits"], out[f"{name}.labels"] = logits, y
        out[f"{name}.acc"], out[f"{name}.f1"] = np.float64(acc), np.float64(f1)
    np.savez(os.path.join(OUT, "g12_metric.npz"), **out)
    print("g12_metric", {k: float(v) for k, v in out.items() if k.endswith((".acc", ".f1"))})


def mask_fixture():
    """G13: Transformer.forward(x, mask) - the key/query mask branch of Attention (models/heads.py:225-232), which no caller
    of the reference uses but which is part of forward()'s signature.  mask is [B, N-1] bool (the reference pads a True for
    the first token); clip 0 keeps everything, clip 1 drops three tokens, clip 2 drops all but one."""
    heads, _, _, _ = load_reference()
    sq = lambda y: y.pow(2).mean()
    for tag, (dim, depth, h, dh, mlp, n) in {"a": (32, 2, 4, 8, 64, 9), "b": (64, 2, 2, 32, 128, 12),
                                            "c": (64, 1, 2, 64, 128, 40)}.items():
        torch.manual_seed(1300 + n)
        tr = heads.Transformer(dim, depth, h, dh, mlp)
        x = torch.randn(3, n, dim)
        mask = torch.ones(3, n - 1, dtype=torch.bool)
        mask[1, [1, 4, n - 2]] = False
        mask[2, 1:] = False
        xx = x.clone().requires_grad_(True)
        y = tr(xx, mask=mask)
        loss = sq(y)
        loss.backward()
        out = {"x": x, "mask": mask, "y": y.detach(), "dx": xx.grad, "loss": loss.detach()}
        for k, v in tr.state_dict().items():
            out["p." + k] = v
        for k, p in tr.named_parameters():
            out["g." + k] = p.grad
        save(f"g13_transformer_mask_{tag}", dim=dim, depth=depth, heads=h, dim_head=dh, mlp_dim=mlp, **out)


def identity_fixture():
    """G14: Transformer with heads == 1 and dim_head == dim, where Attention.to_out is nn.Identity (models/heads.py:207,
    214-217): no out-projection parameters in the state_dict, no Dropout after the attention output."""
    heads, _, _, _ = load_reference()
    torch.manual_seed(1400)
    tr = heads.Transformer(32, 2, 1, 32, 64)
    assert isinstance(tr.layers[0][0].fn.fn.to_out, torch.nn.Identity)
    save("g14_transformer_identity_out", dim=32, depth=2, heads=1, dim_head=32, mlp_dim=64,
         **module_io(tr, torch.randn(3, 10, 32), lambda y: y.pow(2).mean()))


def va_former_fixture():
    """G15: VA_former (models/heads.py:341-372; SpatialFormer's ``va_head``, sformer.py:358, 378-380): BatchNorm1d -> two
    Linear(in, E) -> [B, 2, E] + pos_embedding -> Transformer(E, depth 2, 8 heads of 32, mlp 128) -> one bias-free Linear(E, 1)
    per token.  Two cases at the reference's own emb_dim = 128 (input_dim 64 keeps the files small): eval mode with perturbed
    running statistics, and train mode (batch statistics; dropout 0 so the outputs are deterministic)."""
    heads, _, _, _ = load_reference()
    sq = lambda y: y.pow(2).mean()
    for tag, train in (("eval", False), ("train", True)):
        torch.manual_seed(1500 + int(train))
        m = heads.VA_former(input_dim=64, emb_dim=128)
        m.VA_BN1.running_mean.normal_(0, 0.5)
        m.VA_BN1.running_var.uniform_(0.5, 2.0)
        m.train(train)
        save(f"g15_va_former_{tag}", input_dim=64, emb_dim=128, training=int(train), **module_io(m, torch.randn(6, 64), sq))


if __name__ == "__main__":
    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None
    if only == "g12":
        metric_fixture()
    elif only == "g13":
        mask_fixture()
    elif only == "g14":
        identity_fixture()
    elif only == "g15":
        va_former_fixture()
    else:
        main()
        metric_fixture()
        mask_fixture()
        identity_fixture()
        va_former_fixture()
