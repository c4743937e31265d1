# bash tools/diag/fresh_operand_epi.sh [rows]  -> gpurun_out/fresh_operand_epi_<rows>.txt  (see fresh_operand.py, run_epi)
set -e -o pipefail
ROWS=${1:-10368}
R=$PWD; O=$R/gpurun_out/fo; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/e$ROWS -o s -- python $R/tools/diag/fresh_operand.py --epi $ROWS > $O/epi$ROWS.log 2> $O/epi$ROWS.err
cd $R
python tools/diag/fresh_operand.py --summarise-epi $(find $O/e$ROWS -name 's_kernel_trace.csv' | head -1) $ROWS > gpurun_out/fresh_operand_epi_$ROWS.txt
rm -rf $O/e$ROWS
cat gpurun_out/fresh_operand_epi_$ROWS.txt
