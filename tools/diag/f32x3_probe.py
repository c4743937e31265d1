"""fp32 GEMMs of the parity mode in both arithmetic modes (bf16x3 on the bf16 matrix pipe / the f32-input MFMA): error against
fp64 and time per launch on the layer's shapes at C2 (NT forward, NN dX, TN dW)."""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import avformer_amd as A
from avformer_amd import _lib
from tools.bench_mx8 import timeit
ops = A.ops
R = 10368
shapes = [("NT", R, 1536, 512, False, True), ("NT", R, 512, 512, False, True), ("NT", R, 1024, 512, False, True),
          ("NT", R, 512, 1024, False, True), ("NN", R, 512, 1536, False, False), ("NN", R, 1024, 512, False, False),
          ("NN", R, 512, 1024, False, False), ("TN", 1536, 512, R, True, False), ("TN", 1024, 512, R, True, False),
          ("TN", 512, 1024, R, True, False), ("TN", 512, 512, R, True, False), ("NT", 4096, 4096, 4096, False, True)]
for (form, M, N, K, ta, tb) in shapes:
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn((K, M) if ta else (M, K), generator=g)
    b = torch.randn((N, K) if tb else (K, N), generator=g) / K ** 0.5
    ref = (a.double().t() if ta else a.double()) @ (b.double().t() if tb else b.double())
    ac, bc = a.cuda(), b.cuda()
    line = f"{form} {M}x{N}x{K}:"
    for mode in ("f32", "bf16x3"):
        _lib.set_f32_arithmetic(mode)
        c = ops.gemm(ac, bc, trans_a=ta, trans_b=tb)
        err = float((c.double().cpu() - ref).norm() / ref.norm())
        mx = float((c.double().cpu() - ref).abs().max() / ref.abs().max())
        t = timeit(lambda: ops.gemm(ac, bc, trans_a=ta, trans_b=tb))
        line += f"  {mode}: {t:7.1f} us {2.0 * M * N * K / t / 1e6:6.1f} TF/s relfro {err:.2e} maxrel {mx:.2e} |"
    print(line, flush=True)
_lib.set_f32_arithmetic("bf16x3")
