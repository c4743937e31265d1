set -o pipefail
O=gpurun_out/r2d; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "tn_group" > $O/pytest_tn.log 2>&1; echo "pytest tn rc $?"; tail -3 $O/pytest_tn.log
for big in 0 1; do
  echo "== AVF_TN_BIG=$big"; AVF_TN_BIG=$big python tools/bench_ops.py gemm_tn_group --tokens 512 2>&1 | grep gemm_tn_group
  AVF_TN_BIG=$big python tools/bench_ops.py gemm_tn_group --tokens 324 2>&1 | grep gemm_tn_group
done
