#!/usr/bin/env python3
"""Host-side time of the main segments of one C2-shaped step at B=1 (GPU never back-pressures): perf_counter around the
Python entry points, including the ones that run on the autograd thread.  AVF_BENCH_FORCE_DP=1 adds the data-parallel
wrapper (1-rank RCCL group)."""
import collections, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import avformer_amd as A
from avformer_amd import transformer as T, optim as O, dp as DP, models as M

acc = collections.defaultdict(float)


def timed(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[name] += time.perf_counter() - t0
    return w


T._StackFn.forward = staticmethod(timed("stack.forward", T._StackFn.forward))
T._StackFn._backward = staticmethod(timed("stack.backward", T._StackFn._backward))
O.FusedAdam.step = timed("adam.step", O.FusedAdam.step)
DP.DataParallel._on_layer_grads = timed("dp.hook", DP.DataParallel._on_layer_grads)
DP.DataParallel.finish = timed("dp.finish", DP.DataParallel.finish)
M._FuseTokens.forward = staticmethod(timed("fuse_tokens.fwd", M._FuseTokens.forward))
M._FuseTokens.backward = staticmethod(timed("fuse_tokens.bwd", M._FuseTokens.backward))

use_dp = os.environ.get("AVF_BENCH_FORCE_DP") == "1"
if use_dp:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
torch.manual_seed(0)
B, Tv, Ta, D = 1, 196, 128, 512
model = A.SyntheticAVFormer(D, 6, 8, 64, 1024, Tv, Ta, task="AU").cuda()
opt = A.optim.FusedAdam(model, lr=5e-4, weight_decay=5e-5)
dp = A.dp.DataParallel(model) if use_dp else None
batch = {"clip": torch.randn(B, Tv, D, device="cuda"), "audio_features": torch.randn(B, Ta, D, device="cuda")}
y = (torch.rand(B, 12, device="cuda") > 0.5).float()


def step():
    opt.zero_grad(set_to_none=True)
    loss = model.get_au_loss(model(batch), y)
    loss.backward()
    if dp is not None: dp.finish()
    opt.step()


for _ in range(10): step()
torch.cuda.synchronize(); acc.clear()
n = 50
t0 = time.perf_counter()
for _ in range(n): step()
total = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"host per step {total / n * 1e3:.3f} ms")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]): print(f"  {k:18s} {v / n * 1e3:7.3f} ms")
print(f"  {'other':18s} {(total - sum(acc.values())) / n * 1e3:7.3f} ms  (loss, linear head, autograd engine, zero_grad)")
