"""Development aid: where the HOST time of one C2 training step goes (main thread by section; the autograd thread's Python
backward functions wrapped in cProfile)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import avformer_amd as A


class Args:
    batch = 0; residual = "f32"; no_optimizer = False; torch_adam = False
    config = sys.argv[1] if len(sys.argv) > 1 else "c2"


pr = cProfile.Profile()
wall = {}
for modname in ("transformer", "heads", "models", "loss"):
    m = getattr(A, modname, None)
    for name, obj in list(vars(m).items()):
        if isinstance(obj, type) and issubclass(obj, torch.autograd.Function) and obj is not torch.autograd.Function:
            def mk(orig, key):
                def wrapped(ctx, *g):
                    t = time.perf_counter(); pr.enable()
                    try:
                        return orig(ctx, *g)
                    finally:
                        pr.disable(); wall[key] = wall.get(key, 0.0) + time.perf_counter() - t
                return wrapped
            obj.backward = staticmethod(mk(obj.backward, f"{modname}.{name}"))
dev = torch.device("cuda:0")
r = bench.Region(A, torch, None, Args.config, "bf16", Args, dev, 0, 1, False)
for _ in range(30):
    r.step()
torch.cuda.synchronize()
wall.clear(); pr = cProfile.Profile()
n = 20
mainp = cProfile.Profile()
t0 = time.perf_counter()
mainp.enable()
for _ in range(n):
    r.step()
mainp.disable()
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"{Args.config}: host issue {(t1 - t0) / n * 1e3:.3f} ms/step (profilers on), with sync {(time.perf_counter() - t0) / n * 1e3:.3f}")
for k, v in sorted(wall.items(), key=lambda kv: -kv[1]): print(f"  backward {k}: {v / n * 1e3:.3f} ms/step")
print("---- main thread"); pstats.Stats(mainp).sort_stats("tottime").print_stats(22)
print("---- autograd thread (our backward functions)"); pstats.Stats(pr).sort_stats("tottime").print_stats(16)
