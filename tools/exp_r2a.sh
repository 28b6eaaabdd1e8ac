#!/bin/bash
# round-2 experiment batch A: wave priority by tile rank, 64-channel steps on 64-row tiles, per-workgroup traces
run() {  # lib level cin force prio
  r=$(SVHIP_LIB=$PWD/exp/libsvhip_$1.so SV_CONV_FORCE=$4 SV_CONV_PRIO=$5 python tools/conv_microbench.py --level $2 --cin $3 2>/dev/null | grep "level$2" | cut -c1-72)
  echo "$1 force=[$4] prio=$5 $r"
}
for lvl in 0 1 2; do
  run vecb $lvl 384 "" 0
  run prio $lvl 384 "" 1
  run kc64 $lvl 384 "" 0
  run kc64 $lvl 384 "" 1
done
for f in 64,4,3 64,4,2 32,4,3 32,4,2; do
  run vecb 1 384 $f 0
  run prio 1 384 $f 1
  run kc64 1 384 $f 0
  run kc64 1 384 $f 1
done
run prio 0 384 32,4,3 1
run kc64 0 384 64,4,3 1
run kc64 0 384 64,4,2 1
# per-workgroup traces (one launch each)
rm -f gpurun_out/r2_wg_l0.bin gpurun_out/r2_wg_l1.bin gpurun_out/r2_wg_l1p.bin
SVHIP_LIB=$PWD/exp/libsvhip_prio.so SV_CONV_TRACE=gpurun_out/r2_wg_l0.bin python tools/conv_microbench.py --level 0 --iters 1 > /dev/null 2>&1
SVHIP_LIB=$PWD/exp/libsvhip_prio.so SV_CONV_TRACE=gpurun_out/r2_wg_l1.bin python tools/conv_microbench.py --level 1 --iters 1 > /dev/null 2>&1
SVHIP_LIB=$PWD/exp/libsvhip_prio.so SV_CONV_PRIO=1 SV_CONV_TRACE=gpurun_out/r2_wg_l1p.bin python tools/conv_microbench.py --level 1 --iters 1 > /dev/null 2>&1
SVHIP_LIB=$PWD/exp/libsvhip_prio.so SV_CONV_PRIO=1 SV_CONV_TRACE=gpurun_out/r2_wg_l0p.bin python tools/conv_microbench.py --level 0 --iters 1 > /dev/null 2>&1
ls -la gpurun_out/r2_wg_*.bin
