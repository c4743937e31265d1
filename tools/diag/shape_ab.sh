set -e
R=$PWD; O=$R/gpurun_out/trial2; mkdir -p $O; cd /tmp; export TMPDIR=/tmp AVF_BENCH_SETTLE_S=0
for MODE in 1 0; do
  export AVF_LN_FUSE=$MODE
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/stats_$MODE -o s -- python $R/bench.py --config c3 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events --no-extra > $O/stats_$MODE.json 2> $O/stats_$MODE.err
  rm -f $O/shapes_$MODE.csv
  AVF_SHAPE_LOG=$O/shapes_$MODE.csv timeout -k 10 300 python $R/bench.py --config c3 --steps 1 --warmup 0 --launch eager --no-cpu-baseline --no-kernel-events --no-extra > /dev/null 2> $O/shapes_$MODE.err
done
cd $R
for MODE in 1 0; do echo "== ln_fuse=$MODE"; python tools/shape_table.py $O/stats_$MODE/s_kernel_trace.csv $O/shapes_$MODE.csv $O/shapes_table_$MODE.csv | cut -d, -f2-9 ; done
