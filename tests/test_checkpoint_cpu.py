"""CPU: checkpoint compatibility with the reference's save/resume/pretrain conventions."""
import os

import torch

import avformer_amd as A
from conftest import load_golden, split_golden


def _same(a, b):
    return all(torch.equal(a[k], b[k]) for k in a) and a.keys() == b.keys()


def test_save_resume_roundtrip(tmp_path):
    torch.manual_seed(0)
    m = A.build_model("avformer", task="AU")
    path = A.checkpoint.save_checkpoint(m.state_dict(), str(tmp_path / "pretrain"), "latest.pth")  # train.py:247
    assert os.path.exists(path)
    torch.manual_seed(1)
    m2 = A.build_model("avformer", task="AU")
    assert not _same(m.state_dict(), m2.state_dict())
    res = A.checkpoint.resume(m2, str(tmp_path / "pretrain"))  # train.py:323-326
    assert not res.missing_keys and not res.unexpected_keys
    assert _same(m.state_dict(), m2.state_dict())
    assert A.checkpoint.resume(m2, str(tmp_path / "nowhere")) is None  # absent file: skipped, not a crash


def test_module_prefix_and_base_model_rename(tmp_path):
    """DataParallel-style 'module.' prefixes (avformer.py:32) and 'base_model.' -> 's_former.' (vformer.py:349)"""
    torch.manual_seed(0)
    src = A.TFormer(16, 64, 2, 8, 128, 32)

    class Video(torch.nn.Module):  # stands for the reference VideoModel: s_former (backbone) + t_former
        def __init__(self):
            super().__init__()
            self.s_former = torch.nn.Linear(8, 8)
            self.t_former = A.TFormer(16, 64, 2, 8, 128, 32)

    sd = {"module.t_former." + k: v for k, v in src.state_dict().items()}
    sd["module.base_model.weight"] = torch.full((8, 8), 3.0)
    sd["module.base_model.bias"] = torch.full((8,), 4.0)
    torch.save({"state_dict": sd}, tmp_path / "w.pth")  # vformer.py:335 nests under 'state_dict'
    dst = Video()
    res = A.checkpoint.load_pretrain(dst, str(tmp_path / "w.pth"), freeze=True)
    assert not res.missing_keys and not res.unexpected_keys
    assert _same(src.state_dict(), dst.t_former.state_dict())
    assert torch.all(dst.s_former.weight == 3.0) and torch.all(dst.s_former.bias == 4.0)
    assert all(not p.requires_grad for p in dst.parameters())


def test_reference_weights_load_into_heads():
    """state dicts captured from the reference's own modules (golden fixtures) load with nothing missing"""
    for name, mod in (("g5_au_former", A.AU_former(input_dim=64, emb_dim=32)), ("g6_au_head", A.former_AU_head(emb_dim=64)),
                      ("g7_tformer", A.TFormer(16, 64, 2, 8, 128, 32)), ("g3_transformer_c1", A.Transformer(128, 2, 8, 32, 256))):
        p, _, _ = split_golden(load_golden(name))
        p = {k: v for k, v in p.items() if torch.is_tensor(v)}
        res = mod.load_state_dict(A.checkpoint.remap_state_dict({"module." + k: v for k, v in p.items()}), strict=False)
        assert not res.unexpected_keys and all("num_batches_tracked" in k for k in res.missing_keys), (name, res)


def test_flat_group_follows_replaced_parameters():
    """the packed weight groups of the AU heads resolve their Parameters on the owning modules at every use, so a
    replaced Parameter (load_state_dict(assign=True), manual reassignment) is the one that is packed and differentiated"""
    import torch
    import avformer_amd as A
    head = A.heads.tformer_AU_head(emb_dim=16)
    g = head._last_w
    flat0 = g.get()
    assert all(p.data_ptr() == flat0[i].data_ptr() for i, p in enumerate(g.params))
    new = torch.nn.Parameter(torch.full((1, 16), 3.0))
    head.AU_linear_last5.weight = new
    assert g.params[4] is new
    flat1 = g.get()                      # addresses no longer adjacent: re-packed from the CURRENT objects
    assert torch.equal(flat1[4], torch.full((1, 16), 3.0)) and new.data_ptr() == flat1[4].data_ptr()
    sd = {k: torch.zeros_like(v) for k, v in head.state_dict().items()}
    head.load_state_dict(sd, assign=True)
    assert float(g.get().abs().sum()) == 0.0
