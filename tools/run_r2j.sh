set -o pipefail
O=gpurun_out/r2j; mkdir -p $O
AVF_RECORD_ERRORS=$O/errors.json timeout -k 10 800 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -12 $O/pytest.log
