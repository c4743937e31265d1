"""Development aid: cProfile of the host side of one training step of the real avformer heads (eager)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import avformer_amd as A
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
    return loss
for _ in range(10): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"host issue {(t1-t0)/50*1e3:.3f} ms/step; with sync {(time.perf_counter()-t0)/50*1e3:.3f}")
def part(name, fn, n=50):
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): r = fn()
    dt=(time.perf_counter()-t)/n*1e3; torch.cuda.synchronize(); print(f"  {name}: host {dt:.3f} ms"); return r
part("zero_grad", lambda: opt.zero_grad(set_to_none=True))
out = part("forward", lambda: model(x))
loss = part("loss", lambda: model.get_au_loss(model(x), y))
def fb():
    l = model.get_au_loss(model(x), y); l.backward(); return l
part("fwd+loss+bwd", fb)
part("adam", lambda: opt.step())
acc = [0.0] * 5
for _ in range(50):
    a = time.perf_counter(); opt.zero_grad(set_to_none=True)
    b = time.perf_counter(); out = model(x)
    c = time.perf_counter(); l = model.get_au_loss(out, y)
    d = time.perf_counter(); l.backward()
    e = time.perf_counter(); opt.step()
    f = time.perf_counter()
    for i, dt in enumerate((b - a, c - b, d - c, e - d, f - e)): acc[i] += dt
torch.cuda.synchronize()
print("in-step host ms: zero_grad %.3f forward %.3f loss %.3f backward %.3f adam %.3f" % tuple(v / 50 * 1e3 for v in acc))
pr = cProfile.Profile(); pr.enable()
for _ in range(20): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(35)
