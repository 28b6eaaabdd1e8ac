#!/bin/bash
# instance sweep on the mid pyramid levels with the settled-clock protocol (200 launches each)
for spec in "1 384" "1 416" "2 384" "2 448"; do
  set -- $spec
  for cfg in 64,4,3 32,4,3 32,2,3 16,4,3 64,4,2 32,4,2 16,4,2 128,4,3; do
    r=$(SV_CONV_FORCE=$cfg python tools/conv_microbench.py --level $1 --cin $2 2>/dev/null | grep level | cut -c1-70)
    echo "cfg=$cfg $r"
  done
done
