#!/bin/bash
# A/B of the frame pipeline's stream count and the level-0 hand-over inside ONE lease (boxes differ by +-3 %): every
# configuration twice, interleaved.  usage: tools/ab_streams.sh > gpurun_out/ab_streams.txt
for round in 1 2; do
  for cfg in "2 0" "2 1" "3 0" "3 1" "4 0" "4 1" "5 1"; do
    set -- $cfg
    python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --streams $1 --stagger-level0 $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('streams $1 stagger $2: %.2f frames/s (median of %d; min %.2f max %.2f ms/step); dominant launch frac %.3f, end-to-end %.3f' % (d['value'], d['repeats']['n'], d['repeats']['ms_per_step_min'], d['repeats']['ms_per_step_max'], r['frac'], r['end_to_end']['frac']))
"
  done
done
