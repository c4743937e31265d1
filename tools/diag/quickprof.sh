set -e -o pipefail
export AVF_BENCH_SETTLE_S=0
R=$PWD; O=$R/gpurun_out/quick; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for CFG in c2 c3; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$CFG -o s -- python $R/bench.py --config $CFG --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events --no-extra > $O/stats_$CFG.json 2> $O/stats_$CFG.err
  rm -f $O/shapes_$CFG.csv
  AVF_SHAPE_LOG=$O/shapes_$CFG.csv timeout -k 10 300 python $R/bench.py --config $CFG --steps 1 --warmup 0 --launch eager --no-cpu-baseline --no-kernel-events --no-extra > /dev/null 2> $O/shapes_$CFG.err
done
cd $R
for CFG in c2 c3; do python tools/shape_table.py $O/stats_$CFG/s_kernel_trace.csv $O/shapes_$CFG.csv $O/q_${CFG}_shapes.csv > /dev/null; cp $O/stats_$CFG/s_kernel_stats.csv $O/q_${CFG}_kernel_stats.csv; done
for CFG in c2 c3; do cp $O/stats_$CFG/s_kernel_trace.csv $O/q_${CFG}_kernel_trace.csv; done
rm -rf $O/stats_c2 $O/stats_c3
