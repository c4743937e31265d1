set -o pipefail
O=gpurun_out/r2i; mkdir -p $O
AVF_RECORD_ERRORS=$O/errors.json timeout -k 10 800 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log
python tools/bench_real_model.py 64 2>&1 | grep "real avformer"
timeout -k 10 300 python bench.py --dtype f32 --steps 6 --warmup 2 --no-cpu-baseline --no-extra --no-kernel-events 2>/dev/null | cut -c1-200
