"""-m gpu: a whole training step (zero_grad, forward, AULoss, backward, Adam) captured into one hipGraph and replayed.
Checks (1) capture safety of every library call, (2) replay == eager bit for bit without dropout, (3) with dropout the
device-resident seed advances inside the graph, so each replay draws new masks."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _loss(m, b):
    return m.get_au_loss(m({"clip": b["clip"], "audio_features": b["audio_features"]}), b["labels"])


def _batch(seed, B=16):
    g = torch.Generator().manual_seed(seed)
    return {"clip": torch.randn(B, 512, generator=g).cuda(), "audio_features": torch.randn(B, 512, generator=g).cuda(),
            "labels": (torch.rand(B, 12, generator=g) > 0.5).float().cuda()}


def test_graph_replay_equals_eager_without_dropout():
    import avformer_amd as A
    torch.manual_seed(0)
    m_g = A.build_model("avformer", task="AU").cuda().eval()   # eval(): BatchNorm running stats + no dropout
    for p in m_g.parameters():
        p.requires_grad_(True)
    m_e = copy.deepcopy(m_g)
    opt_g = torch.optim.Adam(m_g.parameters(), lr=1e-3, fused=True, capturable=True)
    opt_e = torch.optim.Adam(m_e.parameters(), lr=1e-3, fused=True, capturable=True)
    gs = A.graphs.GraphedTrainStep(m_g, opt_g, _loss, _batch(1), warmup=2)
    # bring the eager twin to the same state: the 2 warm-up steps ran on batch(1) (capture records, it does not execute)
    for _ in range(2):
        opt_e.zero_grad(set_to_none=True)
        _loss(m_e, _batch(1)).backward()
        opt_e.step()
    for i in range(3):
        b = _batch(10 + i)
        lg = gs(b).clone()
        opt_e.zero_grad(set_to_none=True)
        le = _loss(m_e, b)
        le.backward()
        opt_e.step()
        torch.cuda.synchronize()
        assert torch.equal(lg, le.detach()), (i, lg.item(), le.item())
    for (n, p), (_, q) in zip(m_g.named_parameters(), m_e.named_parameters()):
        assert torch.equal(p, q), n


def test_graph_replay_draws_fresh_dropout_masks():
    import avformer_amd as A
    torch.manual_seed(0)
    m = A.build_model("avformer", task="AU").cuda().train()
    opt = torch.optim.Adam(m.parameters(), lr=0.0, fused=True, capturable=True)   # lr 0: only the masks change
    b = _batch(3, B=32)
    gs = A.graphs.GraphedTrainStep(m, opt, _loss, b, warmup=2)
    head = m.au_head.corr_transformer
    seeds, losses = [], []
    for _ in range(4):
        losses.append(gs(b).item())
        seeds.append(head.last_seed)
    assert seeds == [seeds[0] + i for i in range(4)]        # the captured in-place add advances the device seed
    assert len(set(round(l, 6) for l in losses)) == 4       # same inputs, same weights, different masks


@pytest.mark.parametrize("mode", ["bf16", "mx8"])
def test_graph_replay_of_the_synthetic_model_step(mode):
    """the C2-style step (SyntheticAVFormer + AULoss + the library Adam) as bench.py replays it; mx8: the weight
    re-quantisation, the producer-side image emitters and the top layer's quantiser pass must all be capture-safe"""
    import avformer_amd as A
    torch.manual_seed(1)
    mk = lambda: A.SyntheticAVFormer(256, 2, 4, 64, 512, 24, 16, task="AU", compute_dtype=mode).cuda()
    m_g = mk()
    m_e = mk()
    m_e.load_state_dict(m_g.state_dict())
    opt_g = A.optim.FusedAdam(m_g, lr=1e-3)
    opt_e = A.optim.FusedAdam(m_e, lr=1e-3)

    def batch(seed, B=6):
        g = torch.Generator().manual_seed(seed)
        return {"clip": torch.randn(B, 24, 256, generator=g).cuda(), "audio_features": torch.randn(B, 16, 256, generator=g).cuda(),
                "labels": (torch.rand(B, 12, generator=g) > 0.5).float().cuda()}

    gs = A.graphs.GraphedTrainStep(m_g, opt_g, _loss, batch(1), warmup=2)
    for _ in range(2):
        opt_e.zero_grad(set_to_none=True)
        _loss(m_e, batch(1)).backward()
        opt_e.step()
    for i in range(3):
        b = batch(20 + i)
        lg = gs(b).clone()
        opt_e.zero_grad(set_to_none=True)
        le = _loss(m_e, b)
        le.backward()
        opt_e.step()
        torch.cuda.synchronize()
        assert torch.equal(lg, le.detach()), (i, lg.item(), le.item())
    for (n, p), (_, q) in zip(m_g.named_parameters(), m_e.named_parameters()):
        assert torch.equal(p, q), n
