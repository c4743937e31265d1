python - <<'PY' &
import sys, torch, time
sys.path.insert(0, ".")
import avformer_amd as A
ops = A.ops
a = torch.randn(4096, 4096, device="cuda"); b = torch.randn(4096, 4096, device="cuda")
ab = a.bfloat16(); bb = b.bfloat16()
t0 = time.time()
while time.time() - t0 < 6: 
    for _ in range(20): ops.gemm(a, b)
    torch.cuda.synchronize()
t0 = time.time()
while time.time() - t0 < 6:
    for _ in range(200): ops.gemm(ab, bb)
    torch.cuda.synchronize()
PY
sleep 3; echo "--- during fp32 MFMA GEMM"; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power|fclk" | head -6
sleep 6; echo "--- during bf16 MFMA GEMM"; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power|fclk" | head -6
wait
echo "--- idle"; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | head -4
