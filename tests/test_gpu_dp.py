"""-m gpu: the data-parallel wrapper on real device tensors with the RCCL backend (a 1-rank group is all a one-GPU
box allows; multi-rank semantics are covered on CPU by tests/test_dp_gloo.py).  Exercises the CUDA-only parts:
communication stream, event chaining from the backward hook, record_stream, work.wait() on the compute stream."""
import os

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def test_dp_single_rank_nccl_matches_plain_step():
    import avformer_amd as A
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    try:
        torch.manual_seed(3)
        kw = dict(dim=128, depth=3, heads=8, dim_head=32, mlp_dim=256, t_video=20, t_audio=13, compute_dtype="bf16")
        m_dp = A.SyntheticAVFormer(**kw).cuda()
        m_ref = A.SyntheticAVFormer(**kw).cuda()
        m_ref.load_state_dict(m_dp.state_dict())
        dp = A.dp.DataParallel(m_dp)
        g = torch.Generator().manual_seed(4)
        batch = {"clip": torch.randn(6, 20, 128, generator=g).cuda(), "audio_features": torch.randn(6, 13, 128, generator=g).cuda()}
        labels = (torch.rand(6, 12, generator=g) > 0.5).float().cuda()
        for _ in range(2):  # second iteration: buffers recycled through the caching allocator across streams
            for m in (m_dp, m_ref):
                m.zero_grad(set_to_none=True)
                m.get_au_loss(m(batch), labels).backward()
            dp.finish()
            torch.cuda.synchronize()
            for (n, p), (_, q) in zip(m_dp.named_parameters(), m_ref.named_parameters()):
                assert p.grad is not None and torch.equal(p.grad, q.grad), n
        assert len(dp._pending) == 0
    finally:
        if created:
            dist.destroy_process_group()
