"""Development aid: does stepping a layer's parameters on a side stream as soon as its gradients exist (while the layers below
are still in backward) beat the optimizer launch after backward?   python tools/diag/adam_overlap_probe.py [c2|c3]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import avformer_amd as A
from avformer_amd import _lib
from avformer_amd.transformer import PARAMS_PER_LAYER


class Args:
    batch = 0; residual = "f32"; no_optimizer = False; torch_adam = False
    config = sys.argv[1] if len(sys.argv) > 1 else "c2"


dev = torch.device("cuda:0")
r = bench.Region(A, torch, None, Args.config, "bf16", Args, dev, 0, 1, False)
model, opt = r.model, r.opt
for _ in range(8):
    r.step()
torch.cuda.synchronize()


def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


base = timeit(r.step)
lib = _lib.load()
st = opt._stacks[0]
params = st.flat_parameters()
L = st.depth
hit = opt._flat[(0, "arrays")]
sizes = [p.numel() for p in params[:PARAMS_PER_LAYER]]
side = torch.cuda.Stream()
hp = opt.param_groups[0]
cfg = st._cfg(1, 1)
keep = []


def hook(l, flat):
    off, ptrs = 0, []
    for n in sizes:
        ptrs.append(flat.data_ptr() + 4 * off); off += n
    g = _lib.LayerPtrs(*ptrs)
    ev = torch.cuda.Event(); ev.record()
    side.wait_event(ev)
    lows = hit[4]
    _lib.check(lib.avf_layer_adam_step(C.byref(cfg), C.byref(hit[1][l]), C.byref(g), C.byref(hit[2][l]), C.byref(hit[3][l]),
                                       C.c_void_p(lows[l]), float(hp["lr"]), 0.9, 0.999, float(hp["eps"]),
                                       float(hp["weight_decay"]), C.c_void_p(opt._step_dev.data_ptr()),
                                       C.c_void_p(side.cuda_stream)), "layer_adam")
    keep.append(g)
    return None


def overlapped():
    keep.clear()
    opt.zero_grad(set_to_none=True)
    opt._step_dev.add_(1.0)
    out = model(r_batch)
    loss = model.get_au_loss(out, r_labels)
    loss.backward()
    torch.cuda.current_stream().wait_stream(side)
    stacks, opt._stacks = opt._stacks, []
    opt._step_dev.sub_(1.0)  # step() adds it again
    opt.step()
    opt._stacks = stacks
    st._lowp_ready = True
    st._lowp_ptrs = [p.data_ptr() for p in params]
    st._lowp_versions = [p._version for p in params]
    return loss


# the batch / labels live in the closure of r.step: rebuild the same shapes
c = r.c
g = torch.Generator().manual_seed(1)
r_batch = {"clip": torch.randn(c["batch"], c["t_video"], c["dim"], generator=g).to(dev),
           "audio_features": torch.randn(c["batch"], c["t_audio"], c["dim"], generator=g).to(dev)}
r_labels = (torch.rand(c["batch"], 12, generator=g) > 0.5).float().to(dev)
st.set_grad_hook(hook)
ov = timeit(overlapped)
st.set_grad_hook(None)
base2 = timeit(r.step)
print(f"{Args.config}: optimizer after backward {base:.4f} / {base2:.4f} ms/step; per-layer Adam on a side stream inside backward {ov:.4f} ms/step")
