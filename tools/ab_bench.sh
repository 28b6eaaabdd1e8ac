#!/bin/bash
# tools/ab_bench.sh "<variant names>" [reps]: alternate the in-tree library and exp/libsvhip_<name>.so through the bench's
# timed region (no extras, no CPU baseline) inside ONE GPU lease; prints frames/s, ms/step and the dominant kernel's
# isolated TFLOP/s per run.
reps=${2:-3}
one() { python bench.py --no-cpu-baseline --no-extras --steps 60 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['isolated']['achieved'])"; }
for rep in $(seq $reps); do
  echo "in-tree $(one)"
  for l in $1; do echo "$l $(SVHIP_LIB=$PWD/exp/libsvhip_$l.so one)"; done
done
