"""CPU, world_size=2 over gloo: the data-parallel wrapper (dp.DataParallel) - per-layer gradient buckets handed
over by the transformer's backward hook, the extra bucket for parameters outside the stacks, averaging, and the
parameter broadcast.  The layer math on CPU is the ORACLE (checker role only); the property tested is the one the
8-GPU path relies on: N-rank averaged gradients == single-process gradients on the concatenated batch (the hot path
has no cross-clip coupling)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import avformer_amd as A
import oracle

D, L, H, DH, M = 32, 2, 4, 8, 64


class OracleStack(A.Transformer):
    """Same parameter holders / hook protocol as the HIP Transformer, CPU math from the oracle (tests only)."""

    def forward(self, x, mask=None):
        sd = dict(self.state_dict(keep_vars=True))
        return oracle.transformer_forward(x, sd, self.depth, self.heads)

    def emulate_backward_hooks(self):
        # what _StackFn.backward does on the GPU: per layer (reverse order) one flat fp32 bucket whose views are .grad;
        # the buckets are consecutive slices of one allocation (the wrapper merges adjacent ones into one collective)
        per_layer = sum(p.numel() for p in self.layer_parameters(0))
        flat_all = torch.empty(self.depth * per_layer)
        for l in reversed(range(self.depth)):
            ps = self.layer_parameters(l)
            flat = flat_all[l * per_layer:(l + 1) * per_layer]
            flat.copy_(torch.cat([p.grad.reshape(-1) for p in ps]))
            off = 0
            for p in ps:
                n = p.numel()
                p.grad = flat[off:off + n].view_as(p)
                off += n
            if self._grad_hook is not None:
                self._grad_hook(l, flat)


class TinyModel(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.pos = torch.nn.Parameter(torch.randn(1, 6, D) * 0.1)
        self.stack = OracleStack(D, L, H, DH, M)
        self.fc = torch.nn.Linear(D, 12)

    def forward(self, x):
        return self.fc(self.stack(x + self.pos).mean(1))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out, bucket_layers=2):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)  # deliberately different init per rank: the wrapper must broadcast rank 0's
        model = TinyModel()
        dp = A.dp.DataParallel(model, bucket_layers=bucket_layers)
        launched = []
        real_launch = dp._launch
        dp._launch = lambda t: (launched.append(t.numel()), real_launch(t))[1]
        g = torch.Generator().manual_seed(7)
        x = torch.randn(8, 6, D, generator=g)
        y = (torch.rand(8, 12, generator=g) > 0.5).float()
        # reference: full batch, single process, rank 0's (broadcast) weights
        ref = TinyModel()
        ref.load_state_dict(model.state_dict())
        oracle.au_loss(ref(x), y).backward()
        # data parallel: each rank takes its half
        sl = slice(rank * 4, rank * 4 + 4)
        model.zero_grad(set_to_none=True)
        oracle.au_loss(model(x[sl]), y[sl]).backward()
        model.stack.emulate_backward_hooks()
        dp.finish()
        worst = 0.0
        for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
            worst = max(worst, (p.grad - q.grad).abs().max().item() / (q.grad.abs().max().item() + 1e-12))
        same_w = all(torch.equal(a, b) for a, b in zip(model.state_dict().values(), ref.state_dict().values()))
        st = dp.stats()  # the counters bench.py reports for a multi-GPU run
        assert st["collectives"] == len(launched) and len(st["bucket_bytes"]) == len(launched) and min(st["bucket_bytes"]) > 0
        out.put((rank, worst, same_w, len(dp._pending), len(launched)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("bucket_layers", [1, 2])
def test_dp_two_ranks_match_single_process(bucket_layers):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out, bucket_layers)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, worst, same_w, pending, ncoll in res:
        assert worst < 1e-5, (rank, worst)
        assert same_w and pending == 0
        # two layers: one collective each, or one merged collective; plus the bucket of the parameters outside the stack
        assert ncoll == (3 if bucket_layers == 1 else 2), ncoll


def _worker_byte_rule(rank, world, port, out, depth, layers_per_bucket_bytes):
    """the byte rule of the bucket merger on a `depth`-layer stack: bucket_bytes = layers_per_bucket_bytes x (one layer's bytes)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(100)

        class Deep(TinyModel):
            def __init__(self):
                torch.nn.Module.__init__(self)
                self.pos = torch.nn.Parameter(torch.randn(1, 6, D) * 0.1)
                self.stack = OracleStack(D, depth, H, DH, M)
                self.fc = torch.nn.Linear(D, 12)

        model = Deep()
        layer_bytes = 4 * sum(p.numel() for p in model.stack.layer_parameters(0))
        dp = A.dp.DataParallel(model, bucket_bytes=int(layers_per_bucket_bytes * layer_bytes))
        launched = []
        real_launch = dp._launch
        dp._launch = lambda t: (launched.append(t.numel() * 4), real_launch(t))[1]
        g = torch.Generator().manual_seed(7)
        x = torch.randn(8, 6, D, generator=g)
        y = (torch.rand(8, 12, generator=g) > 0.5).float()
        ref = Deep()
        ref.load_state_dict(model.state_dict())
        oracle.au_loss(ref(x), y).backward()
        sl = slice(rank * 4, rank * 4 + 4)
        oracle.au_loss(model(x[sl]), y[sl]).backward()
        model.stack.emulate_backward_hooks()
        dp.finish()
        worst = max((p.grad - q.grad).abs().max().item() / (q.grad.abs().max().item() + 1e-12)
                    for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()))
        out.put((rank, worst, launched, layer_bytes))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("depth,ratio,expect_layers", [
    (3, 0.85, [1, 1, 1]),     # BASELINE config 4's case: one layer (18.9 MB at d = 768) EXCEEDS the 16 MiB rule -> one collective per layer
    (4, 1.9, [2, 2]),         # configs 2 / 3 / 5: two layers of 8.4 MB per 16 MiB bucket
    (5, 1.9, [2, 2, 1]),      # an odd layer count: the bottom layer (layer 0) flushes what is held
    (3, 100.0, [3]),          # a whole small stack (the 12-token heads) in one collective
])
def test_dp_bucket_byte_rule(depth, ratio, expect_layers):
    """the merger's byte rule (dp.DataParallel, bucket_bytes): which layers share a collective when a layer is larger than the
    rule (C4), when two fit (C2 / C3 / C5), with a remainder, and when the whole stack fits - and the averaged gradients equal the
    single-process ones in every case"""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_byte_rule, args=(r, 2, port, out, depth, ratio)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, worst, launched, layer_bytes in res:
        assert worst < 1e-5, (rank, worst)
        stack_colls, rest = launched[:-1], launched[-1]      # the last collective carries the parameters outside the stack
        assert [b // layer_bytes for b in stack_colls] == expect_layers and all(b % layer_bytes == 0 for b in stack_colls), launched
        assert 0 < rest < layer_bytes


class OracleAULoss(torch.nn.Module):
    """CPU stand-in for loss.AULoss with the same data-parallel protocol: when the wrapper has set ``global_mean`` it hands
    over (sum over kept rows of the row mean, kept rows) instead of dividing locally"""

    def __init__(self):
        super().__init__()
        self.global_mean = None
        self.reduce_eval = False

    def forward(self, z, y):
        keep = y[:, 0] != -1
        k = keep.sum()
        s = oracle.au_loss(z[keep], y[keep]) * k if int(k) > 0 else z.sum() * 0.0
        # the same gate as loss.AULoss.forward: the collective only for the training-mode loss with gradients enabled
        if self.global_mean is not None and ((self.training and torch.is_grad_enabled()) or self.reduce_eval):
            return self.global_mean(s, k.float())
        return s / k


class TinyModelWithLoss(TinyModel):
    def __init__(self):
        super().__init__()
        self.loss_AU = OracleAULoss()


def _worker_unequal(rank, world, port, out, ignored):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(100)
        model = TinyModelWithLoss()
        dp = A.dp.DataParallel(model)
        assert model.loss_AU.global_mean is not None  # the wrapper found the ratio loss
        g = torch.Generator().manual_seed(9)
        x = torch.randn(8, 6, D, generator=g)
        y = (torch.rand(8, 12, generator=g) > 0.5).float()
        y[ignored] = -1  # all of them in rank 0's half: the ranks keep different numbers of rows
        ref = TinyModelWithLoss()
        ref.load_state_dict(model.state_dict())
        loss_ref = oracle.au_loss(ref(x), y)  # loss.py:85-102 on the global batch
        loss_ref.backward()
        sl = slice(rank * 4, rank * 4 + 4)
        model.zero_grad(set_to_none=True)
        loss = model.loss_AU(model(x[sl]), y[sl])
        loss.backward()
        model.stack.emulate_backward_hooks()
        dp.finish()
        worst = 0.0
        for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
            worst = max(worst, (p.grad - q.grad).abs().max().item() / (q.grad.abs().max().item() + 1e-12))
        # what plain mean-of-means would have given (the round-1 behaviour) - must differ, or the case proves nothing
        out.put((rank, worst, abs(loss.item() - loss_ref.item()) / abs(loss_ref.item())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("ignored", [[0, 2, 3], [0, 1, 2, 3]], ids=["3_of_4_ignored_on_rank0", "rank0_all_ignored"])
def test_dp_global_mean_loss_with_unequal_ignored_rows(ignored):
    """AULoss is a ratio (loss.py:85-102): with different numbers of ignored rows per rank the mean of the per-rank means is
    not the global mean.  The wrapper reduces (sum, count): loss and averaged gradients equal the single-process ones."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_unequal, args=(r, 2, port, out, ignored)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, worst, loss_err in res:
        assert worst < 1e-5, (rank, worst)
        assert loss_err < 1e-6, (rank, loss_err)


def _worker_eval_loss(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(100)
        model = TinyModelWithLoss()
        A.dp.DataParallel(model)
        g = torch.Generator().manual_seed(9)
        x = torch.randn(4, 6, D, generator=g)
        y = (torch.rand(4, 12, generator=g) > 0.5).float()
        local = oracle.au_loss(model(x), y).item()
        got = []
        if rank == 0:  # a validation loss on ONE rank: eval mode, then train mode under no_grad - neither may be a collective
            model.eval()
            got.append(model.loss_AU(model(x), y).item())
            model.train()
            with torch.no_grad():
                got.append(model.loss_AU(model(x), y).item())
        dist.barrier()
        out.put((rank, local, got))
    finally:
        dist.destroy_process_group()


def test_dp_eval_loss_on_one_rank_is_local():
    """a loss call that not every rank makes (rank-0 validation) must not enter the (sum, count) all-reduce"""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_eval_loss, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, local, got in res:
        for v in got:
            assert abs(v - local) < 1e-6


def test_auloss_gate_matches_stand_in():
    """the product's AULoss applies the same gate as the stand-in above (source check: the HIP loss cannot run here)"""
    import inspect
    src = inspect.getsource(A.loss.AULoss.forward)
    assert "self.training and torch.is_grad_enabled()" in src and "self.reduce_eval" in src


def test_dp_requires_process_group():
    with pytest.raises(RuntimeError, match="torch.distributed"):
        A.dp.DataParallel(torch.nn.Linear(2, 2))
