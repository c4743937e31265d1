export AVF_TUNING=1
for cfg in c2 c3 c4 c5; do
  for w in 0 1 0 1 0 1; do
    AVF_NT_WPF=$w python bench.py --config $cfg --steps 150 --warmup 5 --no-cpu-baseline --no-extra --no-kernel-events 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$cfg wpf=$w ms/step', d['ms_per_step'])"
  done
done
