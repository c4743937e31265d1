# what limits the shader clock while bench.py replays the step?  bash tools/diag/throttle_probe.sh c2|c3  (AVF_* from the environment)
CFG=${1:-c3}
python bench.py --config $CFG --steps 8000 --warmup 5 --no-cpu-baseline --no-extra --no-kernel-events > gpurun_out/thr_$CFG.json 2>/dev/null &
sleep 10
which amd-smi rocm-smi
for i in 1 2 3; do
  rocm-smi --showclocks --showpower --showtemp --showvoltage 2>/dev/null | grep -E "sclk|fclk|mclk|Power|Temperature|Voltage" | sed -E 's/GPU\[0\]\s*: //' | tr '\n' ';'; echo
  sleep 0.5
done
amd-smi metric -g 0 2>/dev/null | grep -i -E -A12 "throttle|clock|power|violation" | head -120
wait
python -c "import json; d=json.loads([l for l in open('gpurun_out/thr_$CFG.json') if l.startswith('{')][-1]); print('$CFG ms/step', d['ms_per_step'])"
