"""Host time to ISSUE one training step (eager launches), against the device time of the step.

    python tools/diag/host_issue.py [--config c2] [--dp]      (--dp: a 1-rank RCCL process group, gradient hooks + collectives)

The loop issues K steps without waiting for the device (the HIP queue absorbs them) and reads the host clock, then drains the
device: `host_ms` is what Python + the HIP runtime spend per step, `device_ms` the same K steps fenced.  A step whose host time
exceeds its device time is host-bound when launched eagerly (the multi-rank default of bench.py).
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2")
    ap.add_argument("--dp", action="store_true")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--profile", action="store_true", help="cProfile of the issue loop (top 30 by own time)")
    ap.add_argument("--calls", action="store_true", help="host time inside every C-ABI entry point and every collective")
    a = ap.parse_args()
    import torch
    import torch.distributed as dist
    import avformer_amd as A
    import bench
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    if a.dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1)
    args = argparse.Namespace(batch=0, config=a.config, residual="bf16", no_optimizer=False, torch_adam=False, dropout=0.0)
    r = bench.Region(A, torch, dist, a.config, "bf16", args, dev, 0, 1, a.dp)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(10):
            r.step()
        torch.cuda.synchronize()
        res = []
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(a.steps):
                r.step()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            res.append(((t1 - t0) / a.steps * 1e3, (t2 - t0) / a.steps * 1e3))
        if a.profile:
            import cProfile
            import pstats
            pr = cProfile.Profile()
            pr.enable()
            for _ in range(a.steps):
                r.step()
            pr.disable()
            torch.cuda.synchronize()
            pstats.Stats(pr).sort_stats("tottime").print_stats(30)
        if a.calls:
            import collections
            acc = collections.defaultdict(lambda: [0, 0.0])
            lib = A._lib.load()

            def wrap(obj, name, key):
                fn = getattr(obj, name)

                def w(*x, **k):
                    t = time.perf_counter()
                    try:
                        return fn(*x, **k)
                    finally:
                        e = acc[key]
                        e[0] += 1
                        e[1] += time.perf_counter() - t
                setattr(obj, name, w)
            for n in dir(lib):
                if n.startswith("avf_") and callable(getattr(lib, n)):
                    wrap(lib, n, n)
            wrap(dist, "all_reduce", "dist.all_reduce")
            if r.dp is not None:
                wrap(r.dp, "_on_layer_grads", "dp._on_layer_grads (incl. its all_reduce)")
                for st in r.dp._stacks:
                    st.set_grad_hook(r.dp._on_layer_grads)
                wrap(r.dp, "finish", "dp.finish")
            t0 = time.perf_counter()
            for _ in range(a.steps):
                r.step()
            tot = time.perf_counter() - t0
            torch.cuda.synchronize()
            print(f"per step: {tot / a.steps * 1e3:.3f} ms host; inside the wrapped calls:")
            for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
                print(f"  {k:45s} {n / a.steps:6.1f} calls  {t / a.steps * 1e3:7.3f} ms")
    for h, d in res:
        print(f"{a.config} dp={a.dp}: host issue {h:.3f} ms/step, issued + drained {d:.3f} ms/step")
    if a.dp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
