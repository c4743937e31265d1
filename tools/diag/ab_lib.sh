# in-box A/B of two builds of the library: lib/old.so vs lib/new.so (bench.py --no-extra, alternating)
L=multi-modal-multi-label-facial-action-unit-detection-with-transformer_amd/lib
for r in 1 2; do
  for v in old new; do
    cp $L/$v.so $L/libavformer_hip.so
    timeout -k 10 200 python bench.py --no-extra "$@" > gpurun_out/ab_$v$r.json 2>/dev/null || exit 1
    python - <<PY
import json
d=json.loads(open("gpurun_out/ab_$v$r.json").read().strip().splitlines()[-1])
k=d["kernel_classes"]
print("$v$r", d["value"], d["ms_per_step"], {c:k[c]["ms_per_step"] for c in k})
PY
  done
done
cp $L/new.so $L/libavformer_hip.so
