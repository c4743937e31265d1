# bash tools/diag/fresh_operand.sh [rows]  -> gpurun_out/fresh_operand_<rows>.txt  (see fresh_operand.py)
set -e -o pipefail
ROWS=${1:-10368}
R=$PWD; O=$R/gpurun_out/fo; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/t$ROWS -o s -- python $R/tools/diag/fresh_operand.py $ROWS > $O/run$ROWS.log 2> $O/run$ROWS.err
cd $R
python tools/diag/fresh_operand.py --summarise $(find $O/t$ROWS -name 's_kernel_trace.csv' | head -1) $ROWS > gpurun_out/fresh_operand_$ROWS.txt
rm -rf $O/t$ROWS
cat gpurun_out/fresh_operand_$ROWS.txt
