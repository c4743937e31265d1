"""one fp32 GEMM shape in the bf16x3 arithmetic, a few launches (for rocprofv3 counter passes): f32x3_one.py FORM M N K"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import avformer_amd as A
ops = A.ops
form, M, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
ta, tb = {"NT": (False, True), "NN": (False, False), "TN": (True, False)}[form]
a = torch.randn((K, M) if ta else (M, K), device="cuda")
b = torch.randn((N, K) if tb else (K, N), device="cuda")
for _ in range(6):
    ops.gemm(a, b, trans_a=ta, trans_b=tb)
torch.cuda.synchronize()
