for c in c2 c4; do
for w in 5 50 200 5 200; do
timeout -k 10 300 python bench.py --config $c --no-extra --no-cpu-baseline --no-kernel-events --warmup $w --steps 20 > gpurun_out/wp.json 2>gpurun_out/wp.err || { tail -3 gpurun_out/wp.err; exit 1; }
python - "$c warmup=$w" <<PY
import json,sys
d=json.loads(open("gpurun_out/wp.json").read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], d["ms_per_step"])
PY
done
done
