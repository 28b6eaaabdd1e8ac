#!/bin/bash
# offset-range split points of the dominant layer, one launch series each (tools/conv_microbench.py --split)
for lv in 0 1; do
  for sp in 0 14 9,18 10,17 11,16 8,19 12,15 13,14 9,13 13,18 5,13 13,22 9,13,18 7,13,20 9,12,15,18; do
    echo -n "level $lv split $sp: "
    timeout -k 10 120 python tools/conv_microbench.py --level $lv --split $sp --iters 100 2>&1 | grep -E "slot-efficiency|TFLOP" | sed 's/V=.*slot-efficiency=/eff /; s/ tiles=.*//; s/k3 level.*: //; s/algorithmic (.*//' | tr '\n' ' '
    echo
  done
done
