set -e
R=$PWD; O=$R/gpurun_out/trial3; mkdir -p $O; cd /tmp; export TMPDIR=/tmp AVF_BENCH_SETTLE_S=0
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/stats -o s -- python $R/bench.py --config c3 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events --no-extra > $O/stats.json 2> $O/stats.err
rm -f $O/shapes.csv
AVF_SHAPE_LOG=$O/shapes.csv timeout -k 10 300 python $R/bench.py --config c3 --steps 1 --warmup 0 --launch eager --no-cpu-baseline --no-kernel-events --no-extra > /dev/null 2> $O/shapes.err
cd $R
python tools/shape_table.py $O/stats/s_kernel_trace.csv $O/shapes.csv $O/table.csv | grep "gemm_bf16_nt" | cut -d, -f2-16
