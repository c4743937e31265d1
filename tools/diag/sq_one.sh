#!/bin/bash
# SQ counter passes of one command (two passes), condensed by tools/sq_summary.py into profiles/<tag>_sq_counters.csv
#   bash tools/diag/sq_one.sh <tag> python tools/diag/f32x3_one.py NT 10368 1536 512
set -e -o pipefail
TAG=$1; shift
R=$PWD; O=$R/gpurun_out/sq_$TAG; mkdir -p $O
CMD="$1 $R/$2 ${@:3}"
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq1 -o q -- $CMD > /dev/null 2> $O/sq1.err
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $O/sq2 -o q -- $CMD > /dev/null 2> $O/sq2.err
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM --output-format csv -d $O/sq3 -o q -- $CMD > /dev/null 2> $O/sq3.err || true
cd $R
python tools/sq_summary.py $O/sq1/q_counter_collection.csv $O/sq2/q_counter_collection.csv diag_$TAG
cat profiles/diag_${TAG}_sq_counters.csv
python - <<P
import csv,collections
try:
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
    for r in csv.DictReader(open("$O/sq3/q_counter_collection.csv")):
        k=r["Kernel_Name"][:60]; acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k,v in acc.items(): print(k, len(n[k]), {c: round(x/len(n[k])) for c,x in v.items()})
except Exception as e: print("sq3:", e)
P
