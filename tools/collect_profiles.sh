#!/bin/bash
# Runs the rocprofv3 passes profiles/README.md lists (on the GPU box) and condenses them into profiles/<tag>_*.
#   bash tools/collect_profiles.sh r03        (from the repo root; raw output under gpurun_out/final)
# Each pass is its own rocprofv3 process with the program directly after "--" (no wrappers); counters are collected
# without any trace domain beside the kernel trace.  C2 (the bench's `value` workload): kernel stats, FETCH / WRITE traffic,
# two SQ counter passes.  C3 (the north-star shape) and C4 (d=768, T=1024: streaming attention): kernel stats and traffic.  The reference's real head shapes: kernel stats.  C5 in the mx8 and in the bf16 mode: kernel stats, per-shape tables and (round 5) traffic.
set -e -o pipefail
export AVF_BENCH_SETTLE_S=0  # profiler passes: the trace should hold the requested steps, not the settling ones
TAG=${1:-r05}
R=$PWD
O=$R/gpurun_out/final
mkdir -p $O
# the plain bench line of THIS box first (no profiler attached), stored beside the tables: profiles/<tag>_bench_line_same_box.json,
# so that roofline.frac of the line and the fractions of the rocprof tables can be compared on one box (boxes of the pool differ
# by up to 12 % in what their power management allows)
( cd $R && AVF_BENCH_SETTLE_S=0.3 python bench.py --no-cpu-baseline > $O/bench_line_same_box.json 2> $O/bench_line_same_box.err && cp $O/bench_line_same_box.json profiles/${TAG}_bench_line_same_box.json ) || echo "bench line of this box: FAILED (see $O/bench_line_same_box.err)"
echo "bench line of this box done"
cd /tmp
export TMPDIR=/tmp
# every traced pass runs WITHOUT the per-dispatch event pass (--no-kernel-events): a third of the launches averaged in the
# round-2 summaries were the instrumented ones (8-13 % slower), which made rocprof and the bench line disagree
for CFG in c2 c3 c4; do
  B="python $R/bench.py --config $CFG --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-events --no-extra"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$CFG -o s -- python $R/bench.py --config $CFG --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events --no-extra > $O/stats_$CFG.json 2> $O/stats_$CFG.err
  echo "$CFG stats done"
  rm -f $O/shapes_$CFG.csv
  AVF_SHAPE_LOG=$O/shapes_$CFG.csv timeout -k 10 300 python $R/bench.py --config $CFG --steps 1 --warmup 0 --launch eager --no-cpu-baseline --no-kernel-events --no-extra > /dev/null 2> $O/shapes_$CFG.err
  echo "$CFG shape log done"
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_$CFG -o f -- $B > /dev/null 2> $O/fetch_$CFG.err
  echo "$CFG fetch done"
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_$CFG -o w -- $B > /dev/null 2> $O/write_$CFG.err
  echo "$CFG write done"
done
B="python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-events --no-extra"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq1 -o q -- $B > /dev/null 2> $O/sq1.err
echo "sq1 done"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $O/sq2 -o q -- $B > /dev/null 2> $O/sq2.err
echo "sq2 done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_real -o s -- python $R/tools/bench_real_model.py 64 > $O/stats_real.log 2> $O/stats_real.err
echo "real-model stats done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c5 -o s -- python $R/bench.py --config c5 --dtype mx8 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events --no-extra > $O/stats_c5.json 2> $O/stats_c5.err
rm -f $O/shapes_c5.csv
AVF_SHAPE_LOG=$O/shapes_c5.csv timeout -k 10 300 python $R/bench.py --config c5 --dtype mx8 --steps 1 --warmup 0 --launch eager --no-cpu-baseline --no-kernel-events --no-extra > /dev/null 2> $O/shapes_c5.err
B5="python $R/bench.py --config c5 --dtype mx8 --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-events --no-extra"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_c5 -o f -- $B5 > /dev/null 2> $O/fetch_c5.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_c5 -o w -- $B5 > /dev/null 2> $O/write_c5.err
echo "c5 (mx8) stats + shape log + traffic done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c5b -o s -- python $R/bench.py --config c5 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events --no-extra > $O/stats_c5b.json 2> $O/stats_c5b.err
rm -f $O/shapes_c5b.csv
AVF_SHAPE_LOG=$O/shapes_c5b.csv timeout -k 10 300 python $R/bench.py --config c5 --steps 1 --warmup 0 --launch eager --no-cpu-baseline --no-kernel-events --no-extra > /dev/null 2> $O/shapes_c5b.err
B5="python $R/bench.py --config c5 --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-events --no-extra"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_c5b -o f -- $B5 > /dev/null 2> $O/fetch_c5b.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_c5b -o w -- $B5 > /dev/null 2> $O/write_c5b.err
echo "c5 (bf16) stats + shape log + traffic done"
# round 6: the parity mode (compute_dtype f32) in its two arithmetics: three bf16 products per fp32 product (default) and the
# f32-input MFMA (AVF_F32_ARITH=f32 is read by bench.py only, it calls avf_set_f32_arith)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_f32x3 -o s -- python $R/bench.py --dtype f32 --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-events --no-extra > $O/stats_f32x3.json 2> $O/stats_f32x3.err
AVF_F32_ARITH=f32 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_f32m -o s -- python $R/bench.py --dtype f32 --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-events --no-extra > $O/stats_f32m.json 2> $O/stats_f32m.err
echo "f32 parity stats done"
cd $R
cp $O/stats_f32x3/s_kernel_stats.csv profiles/${TAG}_f32_bf16x3_kernel_stats.csv
cp $O/stats_f32m/s_kernel_stats.csv profiles/${TAG}_f32_mfma_kernel_stats.csv
python tools/pmc_traffic.py $O/stats_c2/s_kernel_stats.csv $O/fetch_c2/f_counter_collection.csv $O/write_c2/w_counter_collection.csv $TAG c2
python tools/pmc_traffic.py $O/stats_c3/s_kernel_stats.csv $O/fetch_c3/f_counter_collection.csv $O/write_c3/w_counter_collection.csv ${TAG}_c3 c3
python tools/pmc_traffic.py $O/stats_c4/s_kernel_stats.csv $O/fetch_c4/f_counter_collection.csv $O/write_c4/w_counter_collection.csv ${TAG}_c4 c4
for CFG in c2 c3 c4; do
  python tools/shape_table.py $O/stats_$CFG/s_kernel_trace.csv $O/shapes_$CFG.csv profiles/${TAG}_${CFG}_shapes.csv $O/fetch_$CFG/f_counter_collection.csv $O/write_$CFG/w_counter_collection.csv > /dev/null
done
python tools/shape_table.py $O/stats_c5/s_kernel_trace.csv $O/shapes_c5.csv profiles/${TAG}_c5_mx8_shapes.csv $O/fetch_c5/f_counter_collection.csv $O/write_c5/w_counter_collection.csv > /dev/null
python tools/shape_table.py $O/stats_c5b/s_kernel_trace.csv $O/shapes_c5b.csv profiles/${TAG}_c5_bf16_shapes.csv $O/fetch_c5b/f_counter_collection.csv $O/write_c5b/w_counter_collection.csv > /dev/null
python tools/sq_summary.py $O/sq1/q_counter_collection.csv $O/sq2/q_counter_collection.csv $TAG
cp $O/stats_real/s_kernel_stats.csv profiles/${TAG}_real_heads_kernel_stats.csv
cp $O/stats_c5/s_kernel_stats.csv profiles/${TAG}_c5_mx8_kernel_stats.csv
mkdir -p $O/profiles
cp profiles/${TAG}_* $O/profiles/
grep "real avformer" $O/stats_real.log || true
echo "condensed"
