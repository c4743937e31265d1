set -o pipefail
O=gpurun_out/r2f; mkdir -p $O
AVF_RECORD_ERRORS=$O/errors.json timeout -k 10 600 python -m pytest tests/test_gpu_resid16.py tests/test_gpu_transformer.py tests/test_gpu_ops.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest.log
for res in f32 bf16; do
timeout -k 10 300 python bench.py --residual $res --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$res.json 2> $O/bench_$res.err; echo "bench $res rc $?"
python - <<PY
import json
j=json.load(open("$O/bench_$res.json"))
print("$res C2", j["value"], j["ms_per_step"], {k:v["ms_per_step"] for k,v in j["kernel_classes"].items()})
n=j["north_star_shape"]; print("$res C3", n["ms_per_step"], n["stack_frac_of_mfma_peak"], {k:v["ms_per_step"] for k,v in n["kernel_classes"].items()})
PY
done
