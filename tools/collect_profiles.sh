#!/bin/bash
# Runs the rocprofv3 passes profiles/README.md lists (on the GPU box) and condenses them into profiles/<tag>_*.
#   bash tools/collect_profiles.sh r01        (from the repo root; raw output under gpurun_out/final)
# Each pass is its own rocprofv3 process with the program directly after "--" (no wrappers); counters are collected
# without any trace domain beside the kernel trace.
set -e -o pipefail
TAG=${1:-r01}
R=$PWD
O=$R/gpurun_out/final
mkdir -p $O
cd /tmp
export TMPDIR=/tmp
B="python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-events"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/stats.json 2> $O/stats.err
echo "stats done"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- $B > /dev/null 2> $O/fetch.err
echo "fetch done"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- $B > /dev/null 2> $O/write.err
echo "write done"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq1 -o q -- $B > /dev/null 2> $O/sq1.err
echo "sq1 done"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $O/sq2 -o q -- $B > /dev/null 2> $O/sq2.err
echo "sq2 done"
cd $R
mkdir -p $O/profiles
python tools/pmc_traffic.py $O/stats/s_kernel_stats.csv $O/fetch/f_counter_collection.csv $O/write/w_counter_collection.csv $TAG
python tools/sq_summary.py $O/sq1/q_counter_collection.csv $O/sq2/q_counter_collection.csv $TAG
cp profiles/${TAG}_kernel_stats.csv profiles/${TAG}_pmc_summary.csv profiles/${TAG}_traffic.json profiles/${TAG}_sq_counters.csv $O/profiles/
echo "condensed"
