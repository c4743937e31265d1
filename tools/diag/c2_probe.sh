for c in c2 c3; do
for a in "" "--graph" "" "--graph"; do
timeout -k 10 300 python bench.py --config $c --no-extra --no-cpu-baseline --no-kernel-events $a > gpurun_out/c4p.json 2>gpurun_out/c4p.err || { tail -3 gpurun_out/c4p.err; exit 1; }
python - "$c $a" <<PY
import json,sys
d=json.loads(open("gpurun_out/c4p.json").read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], d["ms_per_step"])
PY
done
done
