#!/bin/bash
# tools/ab_conv.sh "<lib names>" ["<level cin [force]> ..."]: microbenchmark every exp/libsvhip_<name>.so on the given layers
libs=$1; shift
specs=("$@"); [ ${#specs[@]} -eq 0 ] && specs=("0 384" "0 416" "1 384" "2 384" "3 384")
for spec in "${specs[@]}"; do
  set -- $spec
  for l in $libs; do
    r=$(SVHIP_LIB=$PWD/exp/libsvhip_$l.so SV_CONV_FORCE=$3 python tools/conv_microbench.py --level $1 --cin $2 2>/dev/null | grep "level$1" | cut -c1-78)
    echo "$l [$3] $r"
  done
done
