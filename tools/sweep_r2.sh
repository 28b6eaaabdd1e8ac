#!/bin/bash
# instance sweep with the settled-clock protocol at HEAD (round 2): levels 0-3, the Cin values of MinkUNet18D's decoder
for spec in "0 384" "0 416" "1 384" "1 416" "2 384" "2 448" "3 384" "3 512"; do
  set -- $spec
  for cfg in 128,4,3 64,4,3 32,4,3 16,4,3 64,4,2 32,4,2 16,4,2 32,2,3; do
    [ "$1" = 0 ] && [ "$cfg" = "16,4,2" ] && continue
    r=$(SV_CONV_FORCE=$cfg python tools/conv_microbench.py --level $1 --cin $2 2>/dev/null | grep "level$1" | cut -c1-72)
    echo "cfg=$cfg $r"
  done
done
