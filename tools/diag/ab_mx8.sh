# C5 step, same box: bf16 | mx8-fwd | mx8 (AVF_MX8_OUTPROJ=0/1)
run() {
  tag=$1; shift
  timeout -k 10 200 python bench.py --config c5 --no-extra --no-cpu-baseline "$@" > gpurun_out/c5_$tag.json 2>gpurun_out/c5_$tag.err || { tail -5 gpurun_out/c5_$tag.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/c5_$tag.json").read().strip().splitlines()[-1])
k=d["kernel_classes"]
print("$tag", d["value"], d["ms_per_step"], {c:(k[c]["ms_per_step"],k[c]["launches_per_step"], k[c]["frac_of_mfma_peak"]) for c in k})
PY
}
for r in 1 2; do
  run bf16 --dtype bf16
  AVF_MX8_OUTPROJ=0 run fwd_o0 --dtype mx8-fwd
  AVF_MX8_OUTPROJ=1 run fwd_o1 --dtype mx8-fwd
  AVF_MX8_OUTPROJ=0 run mx8_o0 --dtype mx8
  AVF_MX8_OUTPROJ=1 run mx8_o1 --dtype mx8
done
