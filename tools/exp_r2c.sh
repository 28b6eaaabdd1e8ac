#!/bin/bash
run() {  # lib level cin force
  r=$(SVHIP_LIB=$PWD/exp/libsvhip_$1.so SV_CONV_FORCE=$4 python tools/conv_microbench.py --level $2 --cin $3 2>/dev/null | grep "level$2" | cut -c1-72)
  echo "$1 force=[$4] $r"
}
for lvl in 0 1 2; do for l in gd1 gd2 ablg ablw ablgw; do run $l $lvl 384 ""; done; done
run gd1 0 384 128,4,3; run gd2 0 384 128,4,3; run ablg 0 384 128,4,3
run gd1 1 384 16,4,3; run gd2 1 384 16,4,3; run ablg 1 384 16,4,3; run ablgw 1 384 16,4,3
run gd1 1 384 32,4,3; run gd2 1 384 32,4,3
rm -f gpurun_out/r2_wg_ablg.bin
SVHIP_LIB=$PWD/exp/libsvhip_ablg.so SV_CONV_TRACE=gpurun_out/r2_wg_ablg.bin python tools/conv_microbench.py --level 0 --iters 1 > /dev/null 2>&1
python tools/wg_trace.py gpurun_out/r2_wg_ablg.bin | grep -E "kernel span|us per step|tail|resident"
