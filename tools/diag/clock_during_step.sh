# rocm-smi samples (shader clock, package power) while bench.py replays the training step:  bash tools/diag/clock_during_step.sh c2|c3
CFG=${1:-c2}
python bench.py --config $CFG --steps 6000 --warmup 5 --no-cpu-baseline --no-extra --no-kernel-events > gpurun_out/clk_$CFG.json 2>/dev/null &
sleep 9
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Package Power" | sed -E 's/.*\((.*Mhz)\)/\1/; s/.*\(W\): //' | tr '\n' ' '; echo; sleep 0.7; done
wait
python -c "import json; d=json.loads([l for l in open('gpurun_out/clk_$CFG.json') if l.startswith('{')][-1]); print('$CFG ms/step', d['ms_per_step'])"
