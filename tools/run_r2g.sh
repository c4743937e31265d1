set -o pipefail
O=gpurun_out/r2g; mkdir -p $O
AVF_RECORD_ERRORS=$O/errors.json timeout -k 10 800 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -25 $O/pytest.log
python tools/bench_real_model.py 64 2>&1 | grep "real avformer"
