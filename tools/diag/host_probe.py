"""Development aid: host time to ISSUE one training step (no device sync inside) against the device time, and a cProfile of
the issue loop.  python tools/diag/host_probe.py c4"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402
import avformer_amd as A  # noqa: E402


class Args:
    batch = 0
    residual = "f32"
    no_optimizer = False
    torch_adam = False
    config = sys.argv[1] if len(sys.argv) > 1 else "c4"


dev = torch.device("cuda:0")
r = bench.Region(A, torch, None, Args.config, "bf16", Args, dev, 0, 1, False)
for _ in range(5):
    r.step()
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for _ in range(n):
    r.step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{Args.config}: host issue {(t1 - t0) / n * 1e3:.2f} ms/step, with final sync {(t2 - t0) / n * 1e3:.2f} ms/step")
# per-step host times over a longer run (allocator / sync hiccups show as outliers)
ts = []
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(40):
    a = time.perf_counter()
    r.step()
    ts.append((time.perf_counter() - a) * 1e3)
torch.cuda.synchronize()
print(f"40 steps: {(time.perf_counter() - t0) / 40 * 1e3:.2f} ms/step; host per step: min {min(ts):.2f} median {sorted(ts)[20]:.2f} "
      f"max {max(ts):.2f}; steps over 2x median: {[round(t, 1) for t in ts if t > 2 * sorted(ts)[20]]}")
print("allocator:", {k: v for k, v in torch.cuda.memory_stats().items() if k in ("num_alloc_retries", "num_device_alloc", "num_device_free", "reserved_bytes.all.peak", "allocated_bytes.all.peak")})
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    r.step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
