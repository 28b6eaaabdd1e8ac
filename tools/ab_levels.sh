#!/bin/bash
# per-level instance choice INSIDE the frame pipeline (the isolated optimum is not the pipeline's): SV_CONV_FORCE_RANGE forces
# one tile shape on the 384-wide layers of one pyramid level; two interleaved rounds in one lease.
run() {
  SV_CONV_FORCE_RANGE="$2" python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-34s %.2f frames/s (min %.2f max %.2f ms/step) e2e %.3f' % ('$1', d['value'], d['repeats']['ms_per_step_min'], d['repeats']['ms_per_step_max'], d['roofline']['end_to_end']['frac']))
"
}
for round in 1 2; do
  run "default" ""
  run "level1 <64,4,2>" "20000:40000:384:64,4,2"
  run "level1 <32,4,3>" "20000:40000:384:32,4,3"
  run "level1 <16,4,3>" "20000:40000:384:16,4,3"
  run "level2 <16,4,3>" "5000:9000:384:16,4,3"
  run "level2 <32,2,3>" "5000:9000:384:32,2,3"
  run "level3 <32,2,3>" "1200:2500:384:32,2,3"
  run "level0 <64,4,2>" "80000:100000:384:64,4,2"
done
