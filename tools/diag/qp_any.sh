# per-kernel profile of any python script:  bash tools/diag/qp_any.sh TAG script.py [args...]  -> gpurun_out/qp/TAG_kernel_stats.csv
set -e -o pipefail
TAG=$1; shift
R=$PWD; O=$R/gpurun_out/qp; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
S=$1; shift
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$TAG -o s -- python $R/$S "$@" > $O/$TAG.log 2> $O/$TAG.err
cd $R
cp $O/$TAG/s_kernel_stats.csv $O/${TAG}_kernel_stats.csv
rm -rf $O/$TAG
