#!/bin/bash
# tools/ab_args.sh "<bench.py arguments>" ... : the bench's timed region under different command lines (quoted groups),
# alternating, three rounds, inside one GPU lease.
one() { python bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"; }
for rep in 1 2 3; do for g in "$@"; do echo "[$g] $(one $g)"; done; done
