# quickprof.sh for several environments: bash tools/diag/quickprof_env.sh TAG1 "ENV1=.. ENV2=.." TAG2 "..."  (per-shape tables under gpurun_out/quick/<TAG>_*)
set -e -o pipefail
export AVF_BENCH_SETTLE_S=0
R=$PWD; O=$R/gpurun_out/quick; mkdir -p $O
while [ $# -ge 2 ]; do
  TAG=$1; ENVS=$2; shift 2
  for CFG in c2 c3; do
    ( cd /tmp; export TMPDIR=/tmp; export $ENVS DUMMY_QP=1
      timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_${TAG}_$CFG -o s -- python $R/bench.py --config $CFG --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events --no-extra > $O/st_${TAG}_$CFG.json 2> $O/st_${TAG}_$CFG.err
      rm -f $O/sh_${TAG}_$CFG.csv
      AVF_SHAPE_LOG=$O/sh_${TAG}_$CFG.csv timeout -k 10 300 python $R/bench.py --config $CFG --steps 1 --warmup 0 --launch eager --no-cpu-baseline --no-kernel-events --no-extra > /dev/null 2> $O/sh_${TAG}_$CFG.err )
    python tools/shape_table.py $O/st_${TAG}_$CFG/s_kernel_trace.csv $O/sh_${TAG}_$CFG.csv $O/${TAG}_${CFG}_shapes.csv > /dev/null
    rm -rf $O/st_${TAG}_$CFG
  done
done
