import cProfile, os, pstats, sys, torch
sys.path.insert(0, "/root/repo")
import avformer_amd as A
torch.manual_seed(0)
B=64
model = A.build_model("avformer", task="AU").cuda().train()
opt = torch.optim.Adam(model.parameters(), lr=5e-4, weight_decay=5e-5, fused=True)
x = {"clip": torch.randn(B, 512, device="cuda"), "audio_features": torch.randn(B, 512, device="cuda")}
y = (torch.rand(B, 12, device="cuda") > 0.5).float()
def step():
    opt.zero_grad(set_to_none=True)
    loss = model.get_au_loss(model(x), y)
    loss.backward()
    opt.step()
for _ in range(10): step()
torch.cuda.synchronize()
import time
t0=time.perf_counter()
for _ in range(50): step()
torch.cuda.synchronize()
print("ms/step", (time.perf_counter()-t0)/50*1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(40): step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
