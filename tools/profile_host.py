#!/usr/bin/env python3
"""Where the HOST time of one C2-shaped step goes (cProfile over 40 steps at B=1, so the GPU never back-pressures)."""
import cProfile, os, pstats, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import avformer_amd as A

torch.manual_seed(0)
B, Tv, Ta, D = 1, 196, 128, 512
model = A.SyntheticAVFormer(D, 6, 8, 64, 1024, Tv, Ta, task="AU").cuda()
opt = A.optim.FusedAdam(model, lr=5e-4, weight_decay=5e-5)
batch = {"clip": torch.randn(B, Tv, D, device="cuda"), "audio_features": torch.randn(B, Ta, D, device="cuda")}
y = (torch.rand(B, 12, device="cuda") > 0.5).float()


def step():
    opt.zero_grad(set_to_none=True)
    loss = model.get_au_loss(model(batch), y)
    loss.backward()
    opt.step()


for _ in range(10): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(40): step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(32)
