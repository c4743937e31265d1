#!/usr/bin/env python3
"""Step time of the REAL avformer head shapes (reference avformer.py: two AU_former stacks d=128 L=2 over 12 tokens +
former_AU_head d=256 L=3 over 12 tokens; B=64 = opts.py default batch).  These shapes are launch-bound."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import avformer_amd as A
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
torch.manual_seed(0)
model = A.build_model("avformer", task="AU").cuda().train()
if os.environ.get("AVF_SINGLE_STREAM") == "1":
    model.concurrent_streams = False
use_torch = os.environ.get("AVF_TORCH_ADAM") == "1"
mk = (lambda cap: torch.optim.Adam(model.parameters(), lr=5e-4, weight_decay=5e-5, fused=True, capturable=cap)) if use_torch \
    else (lambda cap: A.optim.FusedAdam(model, lr=5e-4, weight_decay=5e-5))
opt = mk(False)
x = {"clip": torch.randn(B, 512, device="cuda"), "audio_features": torch.randn(B, 512, device="cuda")}
y = (torch.rand(B, 12, device="cuda") > 0.5).float()
def step():
    opt.zero_grad(set_to_none=True)
    loss = model.get_au_loss(model(x), y)
    loss.backward()
    opt.step()
    return loss
for _ in range(100): step()  # the first ~50 eager steps run at 3x the steady-state time (code objects load on first launch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 50
print(f"[{type(opt).__name__}] real avformer heads (train mode, dropout 0.2), B={B}: eager {dt*1e3:.3f} ms/step, {B/dt:.0f} clips/s")
opt2 = mk(True)
batch = dict(x, labels=y)
gs = A.graphs.GraphedTrainStep(model, opt2, lambda m, b: m.get_au_loss(m({"clip": b["clip"], "audio_features": b["audio_features"]}), b["labels"]), batch)
for _ in range(10): gs(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): gs(batch)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 50
print(f"[{type(opt2).__name__}] real avformer heads (train mode, dropout 0.2), B={B}: hipGraph replay {dt*1e3:.3f} ms/step, {B/dt:.0f} clips/s")
