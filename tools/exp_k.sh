#!/bin/bash
python -m pytest "tests/test_gpu_conv.py" -m gpu -x -q -k "32-32 or down_up" 2>&1 | tail -1
for spec in "1 32 32" "0 32 32" "2 32 32"; do
  set -- $spec
  r=$(SV_CONV_THIN_PLAIN=1 python tools/conv_microbench.py --level $1 --cin $2 --cout $3 2>/dev/null | grep "level$1" | cut -c1-120); echo "plain $r"
  r=$(python tools/conv_microbench.py --level $1 --cin $2 --cout $3 2>/dev/null | grep "level$1" | cut -c1-120); echo "xcd   $r"
  r=$(SV_CONV_THIN=22 python tools/conv_microbench.py --level $1 --cin $2 --cout $3 2>/dev/null | grep "level$1" | cut -c1-120); echo "xcd22 $r"
done
