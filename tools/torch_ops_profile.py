#!/usr/bin/env python3
"""Which torch operators (outside the HIP library) launch kernels in one C2 step: torch.profiler table."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import avformer_amd as A  # noqa: E402
from torch.profiler import profile, ProfilerActivity  # noqa: E402

B, Tv, Ta, D = 32, 196, 128, 512
torch.manual_seed(123)
model = A.SyntheticAVFormer(D, 6, 8, 64, 1024, Tv, Ta, task="AU", compute_dtype="bf16", residual_dtype="bf16").cuda()
batch = {"clip": torch.randn(B, Tv, D, device="cuda"), "audio_features": torch.randn(B, Ta, D, device="cuda")}
labels = (torch.rand(B, 12, device="cuda") > 0.5).float()
opt = A.optim.FusedAdam(model, lr=5e-4, weight_decay=5e-5)


def step():
    opt.zero_grad(set_to_none=True)
    loss = model.get_au_loss(model(batch), labels)
    loss.backward()
    opt.step()


for _ in range(5): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(3): step()
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=50))
