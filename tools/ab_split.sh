#!/bin/bash
# offset-range passes of the wide 3x3x3 layers inside the frame pipeline (MRCC_SPLIT_RULES = "min_rows:cuts;..."): which
# levels, how many passes.  Two interleaved rounds in one lease.
run() {
  MRCC_SPLIT_RULES="$2" python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('%-52s %.2f frames/s (min %.2f max %.2f ms/step) e2e %.3f isolated %.3f' % ('$1', d['value'], d['repeats']['ms_per_step_min'], d['repeats']['ms_per_step_max'], r['end_to_end']['frac'], r['isolated']['frac']))
"
}
for round in 1 2; do
  run "single pass everywhere" ""
  run "levels 0,1: two passes (14)" "60000:14;20000:14"
  run "level 0: three (9,18), level 1: two (14)" "60000:9,18;20000:14"
  run "levels 0,1: three passes (9,18)" "20000:9,18"
  run "levels 0,1,2: two passes" "5000:14"
  run "level 0 three, levels 1,2 two" "60000:9,18;5000:14"
  run "levels 0..3: two passes" "1500:14"
done
