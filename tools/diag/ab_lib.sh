# in-box A/B of builds of the library: VARIANTS="old new" (files lib/<name>.so), bench.py --no-extra <args>, alternating
L=multi-modal-multi-label-facial-action-unit-detection-with-transformer_amd/lib
V=${VARIANTS:-old new}
for r in 1 2; do
  for v in $V; do
    cp $L/$v.so $L/libavformer_hip.so
    timeout -k 10 300 python bench.py --no-extra --no-cpu-baseline "$@" > gpurun_out/ab_$v$r.json 2>/dev/null || exit 1
    python - <<PY
import json
d=json.loads(open("gpurun_out/ab_$v$r.json").read().strip().splitlines()[-1])
k=d["kernel_classes"]
print("$v$r", d["value"], d["ms_per_step"], {c:k[c]["ms_per_step"] for c in k})
PY
  done
done
cp $L/new.so $L/libavformer_hip.so
