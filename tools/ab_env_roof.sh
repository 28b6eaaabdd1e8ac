#!/bin/bash
# tools/ab_env_roof.sh "<VAR=value ... [-- bench args]>" ...: as tools/ab_env.sh, printing the timed-region and isolated
# fractions of the dominant layer as well; words after "--" inside a group go to bench.py
one() {
  local envs=() args=() seen=0
  for w in "$@"; do if [ "$w" = "--" ]; then seen=1; elif [ $seen = 1 ]; then args+=("$w"); else envs+=("$w"); fi; done
  env "${envs[@]}" python bench.py --no-cpu-baseline --no-extras --steps 60 --warmup 10 "${args[@]}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], 'timed', r['frac'], 'all', r['all_launches_of_kernel']['frac'], 'isolated', r['isolated']['frac'])"; }
for rep in 1 2 3; do for g in "$@"; do echo "[$g] $(one $g)"; done; done
