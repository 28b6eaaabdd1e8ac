#!/bin/bash
# frames/s against the number of compute streams and the dual launch's tail share (SV_CONV_TAIL), inside one GPU lease
run() { python bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 10 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline'].get('isolated',{}).get('achieved'))"; }
for s in 1 2 3; do echo "streams=$s $(run --streams $s)"; done
for t in 0 0.08 0.15 0.25 0.35; do echo "tail=$t $(SV_CONV_TAIL=$t run)"; done
echo "streams=2 again $(run --streams 2)"
