import os, sys, cProfile, pstats, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import avformer_amd as A
torch.manual_seed(0)
model = A.build_model("avformer", task="AU").cuda().train()
opt = torch.optim.Adam(model.parameters(), lr=5e-4, weight_decay=5e-5, fused=True)
B = 64
x = {"clip": torch.randn(B, 512, device="cuda"), "audio_features": torch.randn(B, 512, device="cuda")}
y = (torch.rand(B, 12, device="cuda") > 0.5).float()
def step():
    model.zero_grad(set_to_none=True)
    loss = model.get_au_loss(model(x), y)
    loss.backward()
    opt.step()
for _ in range(10): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(30): step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
