set -o pipefail
O=gpurun_out/r2k; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_configs.py -m gpu -q -x -k "tn_group or full_size or c2" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
for x in 0 1; do
  for cfg in c2 c3; do
  AVF_TN_XCD=$x timeout -k 10 300 python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/bench_${cfg}_$x.json 2> $O/bench_${cfg}_$x.err; echo "bench $cfg xcd=$x rc $?"
  python - <<PY
import json
j=json.load(open("$O/bench_${cfg}_$x.json"))
print("$cfg xcd=$x", j["value"], j["ms_per_step"], j["kernel_classes"]["gemm_bf16_tn"])
PY
  done
done
B=tools/diag/bin/nt_pp
for shape in "16384 512 512" "16384 512 1024" "10368 512 512" "10368 512 1024"; do for v in 0 3 4; do timeout -k 5 60 $B $shape $v 0 || echo "variant $v FAILED rc $?"; done; done
