run() {
  SV_CONV_FORCE_RANGE="$2" python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-34s %.2f frames/s (min %.2f max %.2f ms/step)' % ('$1', d['value'], d['repeats']['ms_per_step_min'], d['repeats']['ms_per_step_max']))
"
}
for round in 1 2; do
  run "default" ""
  run "level0 <64,4,2>" "80000:100000:384:64,4,2"
  run "level0+1 <64,4,2>" "20000:100000:384:64,4,2"
  run "level0 <32,4,3>" "80000:100000:384:32,4,3"
done
