"""CPU (no GPU needed): the C-ABI shared library builds, loads and exports every symbol that
include/avformer_hip.h declares; size queries (pure host code) behave; the Python host side mirrors the
reference's parameter schema; the product path fails loudly without a GPU."""
import ctypes
import os
import re

import pytest
import torch

import avformer_amd as A
from conftest import load_golden, split_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    A._build.build()
    return A._lib.load()


def test_header_symbols_all_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "avformer_hip.h")).read()
    declared = set(re.findall(r"\b(avf_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"avf_layer_cfg", "avf_layer_params", "avf_layer_grads"}
    assert declared == set(A._lib.SIGNATURES), declared ^ set(A._lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.avf_version() == 1
    # the binding's struct declarations match the compiled header
    assert lib.avf_sizeof_layer_cfg() == ctypes.sizeof(A._lib.LayerCfg)
    assert lib.avf_sizeof_layer_params() == ctypes.sizeof(A._lib.LayerPtrs)


def test_size_queries_and_config_validation(lib):
    cfg = A._lib.LayerCfg(32, 512, 512, 8, 64, 1024, A._lib.BF16, 1, 1e-5, 0.0)
    saved = lib.avf_layer_saved_bytes(ctypes.byref(cfg))
    R = 32 * 512
    # bf16: h1,h2 (D) + qkv (3I) + o (I) + u,g (M) at 2 B, x_mid fp32, 4 stats vectors + lse
    expect = R * 2 * (512 * 2 + 1536 + 512 + 2 * 1024) + R * 512 * 4 + 4 * R * 4 + 32 * 8 * 512 * 4
    assert expect <= saved <= expect + 64 * 256
    assert lib.avf_layer_lowp_bytes(ctypes.byref(cfg)) >= 2 * 2 * (1536 * 512 + 512 * 512 + 2 * 512 * 1024)
    assert lib.avf_layer_workspace_bytes(ctypes.byref(cfg)) > 0
    cfg32 = A._lib.LayerCfg(4, 64, 128, 8, 32, 256, A._lib.F32, 1, 1e-5, 0.0)
    assert lib.avf_layer_lowp_bytes(ctypes.byref(cfg32)) == 0
    bad = A._lib.LayerCfg(4, 64, 128, 8, 48, 256, A._lib.BF16, 1, 1e-5, 0.0)
    assert lib.avf_layer_saved_bytes(ctypes.byref(bad)) == 0
    assert b"dim_head" in lib.avf_last_error()
    drop = A._lib.LayerCfg(4, 64, 128, 8, 32, 256, A._lib.F32, 1, 1e-5, 0.2)  # fp32 mode with live dropout (round 2)
    assert lib.avf_layer_saved_bytes(ctypes.byref(drop)) > 0
    nodrop = A._lib.LayerCfg(4, 64, 128, 8, 32, 256, A._lib.F32, 1, 1e-5, 0.0)
    # ... which needs two fp32 masked gradient copies in the workspace
    assert lib.avf_layer_workspace_bytes(ctypes.byref(drop)) >= lib.avf_layer_workspace_bytes(ctypes.byref(nodrop)) + 2 * 256 * 128 * 4
    bad_drop = A._lib.LayerCfg(4, 64, 130, 8, 32, 256, A._lib.F32, 1, 1e-5, 0.2)  # dropout needs dim % 4 == 0
    assert lib.avf_layer_saved_bytes(ctypes.byref(bad_drop)) == 0
    assert b"dropout" in lib.avf_last_error()
    assert lib.avf_gemm_workspace_bytes(A._lib.BF16, 1, 0, 1536, 512, 16384) > 0
    # fp32: split-K slabs of the weight-gradient shapes (round 5; optional - without a workspace avf_gemm runs unsplit) and of
    # skinny shapes; nothing for a shape whose tiles already fill the chip
    assert lib.avf_gemm_workspace_bytes(A._lib.F32, 1, 0, 1536, 512, 16384) % (1536 * 512 * 4) == 0
    assert lib.avf_gemm_workspace_bytes(A._lib.F32, 1, 0, 1536, 512, 16384) > 0
    assert lib.avf_gemm_workspace_bytes(A._lib.F32, 0, 1, 16384, 1536, 512) == 0


def test_state_dict_schema_matches_reference_fixture():
    p, _, r = split_golden(load_golden("g3_transformer_c1"))
    t = A.Transformer(r["dim"], r["depth"], r["heads"], r["dim_head"], r["mlp_dim"])
    sd = t.state_dict()
    assert list(sd.keys()) == list(p.keys())
    assert all(sd[k].shape == p[k].shape for k in p)
    t.load_state_dict(p, strict=True)


def test_same_seed_same_init_as_reference_fixture():
    """the holders are built in the reference's RNG order: seeding like make_golden.py reproduces its weights"""
    p, _, r = split_golden(load_golden("g3_transformer_c1"))
    torch.manual_seed(1003)
    t = A.Transformer(r["dim"], r["depth"], r["heads"], r["dim_head"], r["mlp_dim"])
    for k, v in t.state_dict().items():
        assert torch.equal(v, p[k]), k


def test_head_schemas_match_reference_fixtures():
    for name, mod in (("g5_au_former", A.AU_former(input_dim=64, emb_dim=32)),
                      ("g6_au_head", A.tformer_AU_head(emb_dim=64)),
                      ("g15_va_former_eval", A.VA_former(input_dim=64, emb_dim=128)),
                      ("g7_tformer", A.TFormer(16, 64, 2, 8, 128, 32))):
        p, _, _ = split_golden(load_golden(name))
        ref_keys = [k for k, v in p.items()]
        assert list(mod.state_dict().keys()) == ref_keys, name


def test_no_cpu_fallback():
    t = A.Transformer(32, 1, 8, 32, 64)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        t(torch.randn(1, 4, 32))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        A.AULoss()(torch.zeros(2, 12), torch.zeros(2, 12))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        A.ops.layernorm_fwd(torch.zeros(2, 8), torch.ones(8), torch.zeros(8))


def test_dropout_configuration():
    t = A.Transformer(32, 1, 8, 32, 64, dropout=0.2, compute_dtype="f32")
    t.train()
    assert abs(t._cfg(1, 4).dropout_p - 0.2) < 1e-7  # live in both modes since round 2
    t.eval()
    assert t._cfg(1, 4).dropout_p == 0.0
    t = A.Transformer(32, 1, 8, 32, 64, dropout=0.2)  # bf16: dropout is live in train(), identity in eval()
    t.train()
    seed_t = torch.zeros(1, dtype=torch.int64)
    c = t._cfg(1, 4, layer=3, seed_t=seed_t)
    assert abs(c.dropout_p - 0.2) < 1e-7 and c.layer_index == 3 and c.seed_dev == seed_t.data_ptr()
    s1 = t._advance_seed(torch.device("cpu"))
    s2 = t._advance_seed(torch.device("cpu"))
    assert int(s2) == int(s1) + 1 and t.last_seed == int(s2) & 0xFFFFFFFFFFFFFFFF
    t.eval()
    assert t._cfg(1, 4, seed_t=seed_t).dropout_p == 0.0 and t._advance_seed(torch.device("cpu")) is None


def test_registry_has_the_former_entries_of_train_py():
    """train.py:292-303: avformer / vformer / tformer / sformer resolve; constructor kwargs, .modes, .task and the
    reference's parameter names (checkpoints load with strict=False) are in place - no GPU needed to construct"""
    for name in ("sformer", "vformer", "tformer"):
        m = A.models.build_model(name, modality="A;V;M", task="AU")
        assert m.modes == ["clip"] and m.task == "AU"
        assert hasattr(m, "get_au_loss") and hasattr(m, "get_ex_loss") and hasattr(m, "get_va_loss")
    sd = A.models.build_model("sformer").state_dict()
    assert "base_model.pos_embedding" in sd and "base_model.spatial_transformer.layers.0.0.fn.fn.to_qkv.weight" in sd
    assert "au_head.AU_linear_p1.weight" in sd and "fc.1.weight" in sd
    sd = A.models.build_model("tformer").state_dict()
    assert "video_model.s_former.pos_embedding" in sd and "video_model.t_former.cls_token" in sd
    assert sd["video_model.t_former.pos_embedding"].shape == (1, 17, 1536) and "au_head.AU_linear_last12.weight" in sd
    sd = A.models.build_model("vformer").state_dict()
    assert sd["video_model.t_former.pos_embedding"].shape == (1, 17, 512) and sd["fc.3.weight"].shape == (21, 256)
    with pytest.raises(KeyError):
        A.models.build_model("emonet")


def test_registry():
    assert "avformer" in A.MODEL_REGISTRY
    m = A.build_model("avformer", modality="A;V;M", task="AU")
    assert m.modes == ['clip', 'audio_features'] and m.task == "AU"
    with pytest.raises(KeyError):
        A.build_model("resnet")
