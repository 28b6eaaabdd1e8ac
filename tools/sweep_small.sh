#!/bin/bash
# sweep kernel instances on the small pyramid levels (latency-bound layers) of the Cfg-2 cloud
for spec in "2 384 384" "3 384 384" "3 128 128" "4 256 256" "2 64 64" "1 32 32"; do
  set -- $spec
  for cfg in 32,2,3 16,4,3 32,4,3 32,2,2 16,4,2 32,4,2 64,2,2 16,4,1 32,4,1 32,2,1 64,2,1 64,1,1; do
    r=$(SV_CONV_FORCE=$cfg python tools/conv_microbench.py --level $1 --cin $2 --cout $3 --iters 5 2>/dev/null | grep level)
    echo "cfg=$cfg $r"
  done
done
