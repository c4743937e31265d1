set -o pipefail
O=$PWD/gpurun_out/r2h; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_heads.py tests/test_gpu_graph.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest.log
R=$PWD; cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python $R/tools/bench_real_model.py 64 > $O/real.log 2>&1; echo "prof rc $?"
grep "real avformer" $O/real.log
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/stats/s_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:32]:
    print(f'{r["Name"][:90]:90s} {r["Calls"]:>6s} {float(r["AverageNs"])/1e3:8.2f} us {float(r["TotalDurationNs"])/tot*100:5.1f}%')
PY
