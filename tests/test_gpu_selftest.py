"""-m gpu: hardware facts the bf16 kernels are built on (MFMA fragment maps, ds_read_b64_tr_b16)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    import avformer_amd as A
    assert A.ops.device_ok(), "libavformer_hip: no usable gfx950 device"
    return A.ops


def test_mfma_bf16_16x16x32_fragment_map(ops):
    g = torch.Generator().manual_seed(7)
    # small integers: exactly representable in bf16, products/sums exact in fp32; ASYMMETRIC operands
    a = torch.randint(-8, 9, (16, 32), generator=g).float()
    b = torch.randint(-8, 9, (32, 16), generator=g).float()
    c = ops.selftest_mfma_bf16(a.to(torch.bfloat16).cuda(), b.to(torch.bfloat16).cuda()).cpu()
    assert torch.equal(c, a @ b)


def test_mfma_f32_16x16x4_fragment_map(ops):
    g = torch.Generator().manual_seed(8)
    a = torch.randint(-50, 51, (16, 4), generator=g).float()
    b = torch.randint(-50, 51, (4, 16), generator=g).float()
    c = ops.selftest_mfma_f32(a.cuda(), b.cuda()).cpu()
    assert torch.equal(c, a @ b)


def test_ds_read_tr16_b64_semantics(ops):
    # (2r+1) * 2^(c-8): odd part and exponent identify (r, c) uniquely; exact in bf16 (6 significant bits)
    exact = (2 * torch.arange(32).view(32, 1) + 1).float() * torch.pow(2.0, torch.arange(16).view(1, 16).float() - 8)
    tile = exact.to(torch.bfloat16)
    assert torch.equal(tile.float(), exact) and exact.unique().numel() == 512
    out = ops.selftest_tr16(tile.cuda()).cpu().float()  # [64 lanes, 8]
    exp = torch.empty(64, 8)
    for l in range(64):
        i, g = l & 15, l >> 4
        for h in range(2):
            for e in range(4):
                exp[l, h * 4 + e] = tile[16 * h + 4 * g + e, i].float()
    assert torch.equal(out, exp)
