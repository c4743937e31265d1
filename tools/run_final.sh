set -o pipefail
O=gpurun_out/final_r2; mkdir -p $O
AVF_RECORD_ERRORS=$O/errors.json timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; cut -c1-600 $O/bench.json
