#!/bin/bash
# sweep kernel instances per pyramid level (384 -> 384, k27) on the Cfg-2 cloud
for l in 0 1 2 3 4; do
  for cfg in 128,4,3 128,2,3 64,4,3 64,2,3 128,1,3 64,1,3 32,4,3 32,2,3 16,4,3; do
    r=$(SV_CONV_FORCE=$cfg python tools/conv_microbench.py --level $l --iters 5 2>/dev/null | grep level)
    echo "cfg=$cfg $r"
  done
done
