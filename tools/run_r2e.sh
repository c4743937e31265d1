set -o pipefail
O=gpurun_out/r2e; mkdir -p $O
timeout -k 10 800 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log
for cfg in c2 c4; do for big in 0 1; do
AVF_TN_BIG=$big timeout -k 10 300 python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/bench_${cfg}_$big.json 2> $O/bench_${cfg}_$big.err; echo "bench $cfg big=$big rc $?"
python - <<PY
import json
j=json.load(open("$O/bench_${cfg}_$big.json"))
print("$cfg big=$big", j["value"], j["ms_per_step"], j["kernel_classes"]["gemm_bf16_tn"])
PY
done; done
