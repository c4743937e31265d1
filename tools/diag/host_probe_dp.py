"""Development aid: host-issue time against device time of the data-parallel eager step (1-rank RCCL group on one GPU)."""
import os, sys, time
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist
import bench
import avformer_amd as A


class Args:
    batch = 0; residual = "bf16"; no_optimizer = False; torch_adam = False
    config = sys.argv[1] if len(sys.argv) > 1 else "c2"


torch.cuda.set_device(0)
dist.init_process_group("nccl")
dev = torch.device("cuda:0")
for use_dp in (False, True, False, True):
    r = bench.Region(A, torch, dist, Args.config, "bf16", Args, dev, 0, 1, use_dp)
    for _ in range(30):
        r.step()
    torch.cuda.synchronize()
    n = 30
    t0 = time.perf_counter()
    for _ in range(n):
        r.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"dp={use_dp}: host issue {(t1 - t0) / n * 1e3:.3f} ms/step, with sync {(t2 - t0) / n * 1e3:.3f} ms/step")
    del r
# where the data-parallel host time goes: wrap the hook and finish() with timers / a profiler
import cProfile, pstats
r = bench.Region(A, torch, dist, Args.config, "bf16", Args, dev, 0, 1, True)
for _ in range(20):
    r.step()
torch.cuda.synchronize()
pr = cProfile.Profile()
wall = {"hook": 0.0, "finish": 0.0, "loss": 0.0}
dp = r.dp
orig_hook, orig_finish, orig_gm = dp._on_layer_grads, dp.finish, dp.global_mean


def hook(l, f):
    t = time.perf_counter(); pr.enable()
    try:
        return orig_hook(l, f)
    finally:
        pr.disable(); wall["hook"] += time.perf_counter() - t


def finish():
    t = time.perf_counter(); pr.enable()
    try:
        return orig_finish()
    finally:
        pr.disable(); wall["finish"] += time.perf_counter() - t


def gm(s, k):
    t = time.perf_counter(); pr.enable()
    try:
        return orig_gm(s, k)
    finally:
        pr.disable(); wall["loss"] += time.perf_counter() - t


for st in dp._stacks:
    st.set_grad_hook(hook)
dp.finish = finish
for m in r.model.modules():
    if hasattr(m, "global_mean"):
        m.global_mean = gm
n = 20
for _ in range(n):
    r.step()
torch.cuda.synchronize()
print({k: round(v / n * 1e3, 3) for k, v in wall.items()}, "ms/step")
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
dist.destroy_process_group()
