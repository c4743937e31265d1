# quick per-kernel profile of one bench.py invocation:  bash tools/diag/qp.sh TAG [bench.py args...]
# -> gpurun_out/qp/TAG_kernel_stats.csv + TAG.txt (us per step per kernel, steps counted by au_loss_kernel launches)
set -e -o pipefail
export AVF_BENCH_SETTLE_S=0
TAG=$1; shift
R=$PWD; O=$R/gpurun_out/qp; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$TAG -o s -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events --no-extra "$@" > $O/$TAG.json 2> $O/$TAG.err
cd $R
cp $O/$TAG/s_kernel_stats.csv $O/${TAG}_kernel_stats.csv
rm -rf $O/$TAG
python tools/diag/qp_summary.py $O/${TAG}_kernel_stats.csv > $O/$TAG.txt
