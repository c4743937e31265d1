set -o pipefail
O=gpurun_out/r2c; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "tn_group or gemm_bf16_tn" > $O/pytest_tn.log 2>&1; echo "pytest tn rc $?"; tail -3 $O/pytest_tn.log
for big in 0 1; do for st in 3 2; do
  if [ $big = 0 ] && [ $st = 2 ]; then continue; fi
  echo "== AVF_TN_BIG=$big AVF_TN_STAGES=$st"; AVF_TN_BIG=$big AVF_TN_STAGES=$st python tools/bench_ops.py gemm_tn_group --tokens 512 2>&1 | grep gemm_tn_group
  AVF_TN_BIG=$big AVF_TN_STAGES=$st python tools/bench_ops.py gemm_tn_group --tokens 324 2>&1 | grep gemm_tn_group
done; done
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python - <<'PY'
import json
j=json.load(open("gpurun_out/r2c/bench.json"))
print("C2", j["value"], j["ms_per_step"], j["kernel_classes"]["gemm_bf16_tn"])
print("C3", j["north_star_shape"]["ms_per_step"], j["north_star_shape"]["stack_frac_of_mfma_peak"], j["north_star_shape"]["kernel_classes"]["gemm_bf16_tn"])
print("f32", j.get("f32_parity"))
PY
