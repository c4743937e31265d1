# kernel stats of the one-rank data-parallel rehearsal (AVF_BENCH_FORCE_DP=1) beside the plain step, same box
set -e -o pipefail
export AVF_BENCH_SETTLE_S=0
R=$PWD; O=$R/gpurun_out/dpprof; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for MODE in plain dp; do
  if [ $MODE == dp ]; then export AVF_BENCH_FORCE_DP=1; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$MODE -o s -- python $R/bench.py --config c2 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events --no-extra > $O/$MODE.json 2> $O/$MODE.err
  cp $O/$MODE/s_kernel_stats.csv $O/${MODE}_kernel_stats.csv
  cp $O/$MODE/s_kernel_trace.csv $O/${MODE}_kernel_trace.csv
  rm -rf $O/$MODE
done
