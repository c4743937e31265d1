# per-SHAPE kernel table of one bench.py configuration, without the PMC passes:
#   bash tools/diag/shapes_quick.sh TAG c2|c3|c4 [bench.py args...]  -> gpurun_out/sq/TAG_shapes.csv (+ TAG.txt: the hot rows)
set -e -o pipefail
export AVF_BENCH_SETTLE_S=0
TAG=$1; CFG=$2; shift; shift
R=$PWD; O=$R/gpurun_out/sq; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/$TAG -o s -- python $R/bench.py --config $CFG --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events --no-extra "$@" > $O/$TAG.json 2> $O/$TAG.err
rm -f $O/${TAG}_log.csv
AVF_SHAPE_LOG=$O/${TAG}_log.csv timeout -k 10 300 python $R/bench.py --config $CFG --steps 1 --warmup 0 --launch eager --no-cpu-baseline --no-kernel-events --no-extra "$@" > /dev/null 2> $O/${TAG}_log.err
cd $R
python tools/shape_table.py $(find $O/$TAG -name 's_kernel_trace.csv' | head -1) $O/${TAG}_log.csv $O/${TAG}_shapes.csv > /dev/null
rm -rf $O/$TAG
python - $O/${TAG}_shapes.csv > $O/$TAG.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    print(f"{r['kernel'][:58]:58s} wg={r['workgroups']:>5s} M={r['M (or B)']:>6s} N={r['N']:>5s} K={r['K (or H*dh)']:>5s} epi={r['epilogue']:>2s} n={r['launches']:>4s} us={float(r['mean_us']):7.2f} frac={float(r['frac_of_bf16_mfma_peak']):.3f}")
PY
