"""Development aid: cProfile of the BACKWARD half of the real avformer heads step (the autograd engine runs our Python backward
functions on its own thread, invisible to a main-thread profile): every custom autograd.Function's backward is wrapped."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import avformer_amd as A
import importlib
pr = cProfile.Profile()
wall = {}
for modname in ("transformer", "heads", "models", "loss", "dp"):
    m = importlib.import_module("avformer_amd." + modname) if False else getattr(A, modname, None)
    if m is None:
        continue
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
B = 64
torch.manual_seed(0)
model = A.build_model("avformer", task="AU").cuda().train()
opt = A.optim.FusedAdam(model, lr=5e-4, weight_decay=5e-5)
x = {"clip": torch.randn(B, 512, device="cuda"), "audio_features": torch.randn(B, 512, device="cuda")}
y = (torch.rand(B, 12, device="cuda") > 0.5).float()
def step():
    opt.zero_grad(set_to_none=True)
    loss = model.get_au_loss(model(x), y)
    loss.backward()
    opt.step()
for _ in range(10): step()
torch.cuda.synchronize()
wall.clear(); pr = cProfile.Profile()
n = 20
t0 = time.perf_counter()
for _ in range(n): step()
torch.cuda.synchronize()
print(f"step {(time.perf_counter()-t0)/n*1e3:.3f} ms (with profiling overhead)")
for k, v in sorted(wall.items(), key=lambda kv: -kv[1]): print(f"  {k}: {v/n*1e3:.3f} ms/step")
pstats.Stats(pr).sort_stats("tottime").print_stats(30)
