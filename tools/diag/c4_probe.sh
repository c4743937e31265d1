for a in "" "--graph" "" "--graph"; do
timeout -k 10 300 python bench.py --config c4 --no-extra --no-cpu-baseline --no-kernel-events $a > gpurun_out/c4p.json 2>gpurun_out/c4p.err || { tail -3 gpurun_out/c4p.err; exit 1; }
python - "$a" <<PY
import json,sys
d=json.loads(open("gpurun_out/c4p.json").read().strip().splitlines()[-1])
print(sys.argv[1] or "eager", d["value"], d["ms_per_step"])
PY
done
