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
            # the wrapped model takes the (sum, count) form of AULoss (one more fp32 rounding in d loss / d logits, which a
            # bf16 cast further down may amplify to one bf16 ulp on single elements): equal to rounding, not bitwise
            for (n, p), (_, q) in zip(m_dp.named_parameters(), m_ref.named_parameters()):
                assert p.grad is not None, n
                err = ((p.grad - q.grad).norm() / (q.grad.norm() + 1e-30)).item()
                assert err < 1e-3, (n, err)
        assert len(dp._pending) == 0
        # a wrapped model is captured only together with its wrapper (finish() must be part of the recorded step)
        with pytest.raises(RuntimeError, match="dp="):
            A.graphs.GraphedTrainStep(m_dp, torch.optim.Adam(m_dp.parameters()), lambda m, b: m.get_au_loss(m(b), labels), batch)
    finally:
        if created:
            dist.destroy_process_group()


def test_dp_step_captured_into_a_hipgraph_equals_the_eager_dp_step():
    """forward + loss reduction + backward + bucketed RCCL all-reduces + finish() + Adam recorded into ONE
    hipGraph (1-rank RCCL group: all a one-GPU box allows).  Replays must leave exactly the parameters the eager
    data-parallel loop leaves - bit for bit (same kernels, same order, no dropout) - over several steps on changing batches."""
    import avformer_amd as A
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29534")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    try:
        torch.manual_seed(5)
        kw = dict(dim=128, depth=3, heads=8, dim_head=32, mlp_dim=256, t_video=20, t_audio=13, compute_dtype="bf16")
        m_g = A.SyntheticAVFormer(**kw).cuda()
        m_e = A.SyntheticAVFormer(**kw).cuda()
        m_e.load_state_dict(m_g.state_dict())
        dp_g, dp_e = A.dp.DataParallel(m_g), A.dp.DataParallel(m_e)
        opt_g = A.optim.FusedAdam(m_g, lr=1e-3, weight_decay=5e-5)
        opt_e = A.optim.FusedAdam(m_e, lr=1e-3, weight_decay=5e-5)
        for o in (opt_g, opt_e):
            for gd in o.param_groups:
                gd["capturable"] = True  # the group of the parameters outside the stacks steps through torch's Adam
        g = torch.Generator().manual_seed(6)

        def make_batch():
            b = {"clip": torch.randn(6, 20, 128, generator=g).cuda(), "audio_features": torch.randn(6, 13, 128, generator=g).cuda(),
                 "labels": (torch.rand(6, 12, generator=g) > 0.5).float().cuda()}
            b["labels"][0] = -1
            return b

        loss_fn = lambda m, b: m.get_au_loss(m({"clip": b["clip"], "audio_features": b["audio_features"]}), b["labels"])
        first = make_batch()
        step = A.graphs.GraphedTrainStep(m_g, opt_g, loss_fn, first, warmup=2, dp=dp_g)   # 2 warm-up steps (capture records, it does not run)
        for _ in range(2):                                                                  # the same two steps, eagerly
            opt_e.zero_grad(set_to_none=True)
            loss_fn(m_e, first).backward()
            dp_e.finish()
            opt_e.step()
        for _ in range(3):
            b = make_batch()
            lg = step(b)
            opt_e.zero_grad(set_to_none=True)
            le = loss_fn(m_e, b)
            le.backward()
            dp_e.finish()
            opt_e.step()
            torch.cuda.synchronize()
            assert torch.equal(lg, le.detach()), (lg.item(), le.item())
        for (n, p), (_, q) in zip(m_g.named_parameters(), m_e.named_parameters()):
            assert torch.equal(p, q), n
    finally:
        if created:
            dist.destroy_process_group()


def test_au_loss_sum_count_form_matches_mean_form():
    """avf_au_loss_sum (numerator, denominator of loss.py:85-102) against the fused mean kernel and the oracle, incl. a
    shard whose rows are all ignored (sum 0, count 0, zero gradient - no NaN leaks into the other ranks' reduction)"""
    import avformer_amd as A
    import oracle
    g = torch.Generator().manual_seed(12)
    z = torch.randn(9, 12, generator=g)
    y = (torch.rand(9, 12, generator=g) > 0.5).float()
    y[[1, 4]] = -1
    pw = torch.tensor(A.loss.AU_POS_WEIGHT)
    sc, grad = A.ops.au_loss_sum(z.cuda(), y.cuda(), pw.cuda())
    loss, grad_mean = A.ops.au_loss(z.cuda(), y.cuda(), pw.cuda())
    assert sc[1].item() == 7.0
    torch.testing.assert_close((sc[0] / sc[1]).cpu(), oracle.au_loss(z, y), rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(grad / sc[1], grad_mean, rtol=1e-6, atol=1e-9)
    sc0, grad0 = A.ops.au_loss_sum(z.cuda(), -torch.ones(9, 12).cuda(), pw.cuda())
    assert sc0.tolist() == [0.0, 0.0] and float(grad0.abs().sum()) == 0.0


def _rccl_worker(rank, world, port, q, backend="nccl"):
    import avformer_amd as A
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    if backend == "nccl":
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:  # gloo over GPU tensors: every rank on cuda:0 (a one-GPU box); the collectives stage through the host
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(3 + rank)  # different init per rank: the wrapper broadcasts rank 0's
        kw = dict(dim=128, depth=3, heads=8, dim_head=32, mlp_dim=256, t_video=20, t_audio=13, compute_dtype="f32")
        m_dp = A.SyntheticAVFormer(**kw).cuda()
        dp = A.dp.DataParallel(m_dp)
        m_ref = A.SyntheticAVFormer(**kw).cuda()
        m_ref.load_state_dict(m_dp.state_dict())
        g = torch.Generator().manual_seed(4)
        clip, aud = torch.randn(8, 20, 128, generator=g).cuda(), torch.randn(8, 13, 128, generator=g).cuda()
        labels = (torch.rand(8, 12, generator=g) > 0.5).float()
        labels[[0, 1, 2]] = -1  # all in rank 0's shard: unequal kept counts
        labels = labels.cuda()
        m_ref.get_au_loss(m_ref({"clip": clip, "audio_features": aud}), labels).backward()
        sl = slice(rank * 4, rank * 4 + 4)
        loss = m_dp.get_au_loss(m_dp({"clip": clip[sl], "audio_features": aud[sl]}), labels[sl])
        loss.backward()
        dp.finish()
        torch.cuda.synchronize()
        worst = max(((p.grad - r.grad).norm() / (r.grad.norm() + 1e-30)).item()
                    for p, r in zip(m_dp.parameters(), m_ref.parameters()))
        q.put((rank, worst))
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (the driver's 8-GPU node; a gpurun box has one)")
def test_dp_two_ranks_rccl_match_single_process():
    """two processes, one GPU each, RCCL: averaged gradients (with unequal ignored rows per rank) == the single-process
    gradients on the concatenated batch, parity mode"""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rccl_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, worst in res:
        assert worst < 1e-4, (rank, worst)


def test_dp_two_ranks_on_one_gpu_match_single_process():
    """the same two-rank case on a ONE-GPU box: two processes, both on cuda:0, a gloo group over the GPU tensors (RCCL refuses
    two ranks on one device).  Everything but the transport is the product path - HIP forward / backward per rank, the
    per-layer hooks, bucket merging, the (sum, count) loss reduction, finish() - and the averaged gradients must equal the
    single-process gradients on the concatenated batch, with unequal ignored rows per rank."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rccl_worker, args=(r, 2, port, q, "gloo")) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, worst in res:
        assert worst < 1e-4, (rank, worst)


def test_bench_two_rank_control_flow_rehearsal_on_one_gpu():
    """`python bench.py --gpus 2` end to end on a ONE-GPU box: its own launcher starts two ranks, both on cuda:0, the collectives
    through gloo (AVF_BENCH_ONE_DEVICE / AVF_BENCH_BACKEND: rehearsal switches of bench.py).  The numbers mean nothing (two ranks
    share one GPU); what is checked is that the multi-rank control flow - rendezvous, rank assertion, the collectively decided
    untimed loops, every timed region under the data-parallel wrapper, the data-parallel report - runs to ONE json line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, AVF_BENCH_ONE_DEVICE="1", AVF_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
                        "--no-cpu-baseline", "--no-kernel-events"], capture_output=True, text=True, env=env, cwd=root, timeout=500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["config"]["global_batch"] == 64
    dp = d["data_parallel"]
    assert dp["rccl_ranks"] == 2 and dp["gradient_collectives_per_step"] >= 2 and dp["loss_collectives_per_step"] == 1
    assert sum(dp["bucket_bytes"]) == dp["gradient_bytes_per_step"] > 50_000_000  # 12.6 M stack parameters + the head, fp32
    # --launch auto with two ranks: the eager region is timed first, then the captured step is ATTEMPTED; gloo's collectives
    # cannot be recorded into a hipGraph, so the attempt must fail cleanly on both ranks and the eager measurement stand
    assert d["launch"].startswith("eager (graph capture failed") and d["value"] > 0
    assert d["eager_ms_per_step"] is not None and abs(d["eager_ms_per_step"] - d["ms_per_step"]) < 1e-6


def test_bench_launch_auto_success_branch_on_one_rank_rccl():
    """the multi-rank default launch mode (--launch auto: eager region first, then the captured data-parallel step under a
    watchdog, adopted only if its first replay validates against the eager loss) with its SUCCESS branch forced on one rank:
    process group over RCCL with world size 1, all-reduces captured into the hipGraph, `value` from the replayed region and
    the eager region's number beside it"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, AVF_BENCH_FORCE_DP="1", AVF_BENCH_AUTO_DP="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_PORT=str(_free_tcp_port()))
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "6", "--warmup", "2", "--no-cpu-baseline",
                        "--no-kernel-events", "--no-extra"], capture_output=True, text=True, env=env, cwd=root, timeout=500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["launch"].startswith("hipGraph replay"), d["launch"]
    assert d["eager_ms_per_step"] is not None and d["ms_per_step"] > 0 and d["data_parallel"]["rccl_ranks"] == 1


def test_bench_recovers_from_a_failed_capture():
    """a capture that fails half-way (AVF_BENCH_FAIL_CAPTURE=1 issues a device synchronisation while the step is being recorded -
    the kind of error a collective that cannot be recorded raises) must leave the process usable: the current stream restored,
    the invalidated capture closed, the sticky HIP error cleared (avf_hip_error_reset) - the eager measurement stands and every
    later part of the run (data-parallel report, other regions) still works"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, AVF_BENCH_FORCE_DP="1", AVF_BENCH_AUTO_DP="1", AVF_BENCH_FAIL_CAPTURE="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_PORT=str(_free_tcp_port()))
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "6", "--warmup", "2", "--no-cpu-baseline",
                        "--no-kernel-events", "--no-extra"], capture_output=True, text=True, env=env, cwd=root, timeout=500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["launch"].startswith("eager (graph capture failed"), d["launch"]
    assert d["value"] > 0 and abs(d["eager_ms_per_step"] - d["ms_per_step"]) < 1e-6
    assert d["data_parallel"]["rccl_ranks"] == 1 and d["data_parallel"]["gradient_collectives_per_step"] >= 2


def test_bench_reports_the_eager_measurement_when_the_graph_attempt_crashes():
    """a crash INSIDE the runtime during the optional graph attempt (AVF_BENCH_FAIL_CAPTURE=segv raises SIGSEGV in the middle of
    the capture) cannot be handled as an exception: the armed crash line (avf_crash_line_arm) prints the eager measurement taken
    before the attempt and the process leaves with status 0 - one JSON line, `launch` says what happened"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, AVF_BENCH_FORCE_DP="1", AVF_BENCH_AUTO_DP="1", AVF_BENCH_FAIL_CAPTURE="segv",
               HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_PORT=str(_free_tcp_port()))
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "6", "--warmup", "2", "--no-cpu-baseline",
                        "--no-kernel-events", "--no-extra"], capture_output=True, text=True, env=env, cwd=root, timeout=500)
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert "crashed inside the runtime" in d["launch"] and d["value"] > 0 and d["n_gpus"] == 1


def _free_tcp_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port
