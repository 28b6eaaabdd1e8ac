#!/bin/bash
# with the offset-range passes in place (launches a third as long): re-check streams, tile heights, tail share, more rules
run() {
  env $2 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline'] or {}
print('%-50s %.2f frames/s (min %.2f max %.2f ms/step) e2e %s' % ('$1', d['value'], d['repeats']['ms_per_step_min'], d['repeats']['ms_per_step_max'], (r.get('end_to_end') or {}).get('frac')))
"
}
for round in 1 2; do
  run "default (levels 0,1: 9,18; 3 streams)" "X=1" ""
  run "4 streams" "X=1" "--streams 4"
  run "2 streams" "X=1" "--streams 2"
  run "3 streams, no hand-over" "X=1" "--stagger-level0 0"
  run "level 0 on <128,4,3>" "SV_CONV_FORCE_RANGE=80000:100000:384:128,4,3" ""
  run "tail share 0.3" "SV_CONV_TAIL=0.3" ""
  run "tail share 0.05" "SV_CONV_TAIL=0.05" ""
  run "rules 20000:9,18;5000:14" "MRCC_SPLIT_RULES=20000:9,18;5000:14" ""
  run "rules 20000:7,14,20" "MRCC_SPLIT_RULES=20000:7,14,20" ""
  run "rules 60000:7,14,20;20000:9,18" "MRCC_SPLIT_RULES=60000:7,14,20;20000:9,18" ""
done
