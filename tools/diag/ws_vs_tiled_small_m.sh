# bash tools/diag/ws_vs_tiled_small_m.sh -> gpurun_out/ws_vs_tiled_small_m.txt (see the .py)
set -e -o pipefail
R=$PWD; O=$R/gpurun_out/wsm; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/ws -o s -- python $R/tools/diag/ws_vs_tiled_small_m.py ws > $O/ws.log 2> $O/ws.err
AVF_TUNING=1 AVF_NT_WS=0 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/tl -o s -- python $R/tools/diag/ws_vs_tiled_small_m.py tiled > $O/tl.log 2> $O/tl.err
cd $R
(python tools/diag/ws_vs_tiled_small_m.py --summarise $(find $O/ws -name s_kernel_trace.csv | head -1); python tools/diag/ws_vs_tiled_small_m.py --summarise $(find $O/tl -name s_kernel_trace.csv | head -1)) | sort -k2,2n -k4,4n -s > gpurun_out/ws_vs_tiled_small_m.txt
rm -rf $O/ws $O/tl
cat gpurun_out/ws_vs_tiled_small_m.txt
