#!/bin/bash
# round-2 experiment batch B: mid-step hand-over (in-tree build) vs the end-of-step hand-over (exp/libsvhip_vecb.so)
python -m pytest tests/test_gpu_conv.py tests/test_gpu_cfg.py tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -3
run() {  # lib level cin force
  lib=$PWD/exp/libsvhip_$1.so; [ "$1" = tree ] && lib=$PWD/markerless-robot-camera-calibration_amd/libsvhip.so
  r=$(SVHIP_LIB=$lib SV_CONV_FORCE=$4 python tools/conv_microbench.py --level $2 --cin $3 2>/dev/null | grep "level$2" | cut -c1-72)
  echo "$1 force=[$4] $r"
}
for lvl in 0 1 2 3 4; do
  cin=384; [ $lvl = 4 ] && cin=256
  run vecb $lvl $cin ""
  run tree $lvl $cin ""
done
for f in 64,4,3 32,4,3 64,4,2 128,4,3; do run vecb 0 416 $f; run tree 0 416 $f; done
for f in 64,4,3 32,4,3 32,4,2 16,4,3; do run vecb 1 384 $f; run tree 1 384 $f; done
for f in 32,4,3 64,4,2 16,4,2; do run vecb 2 384 $f; run tree 2 384 $f; done
run vecb 1 32 ""; run tree 1 32 ""
run vecb 2 64 ""; run tree 2 64 ""
python bench.py --steps 30 --warmup 8 --no-cpu-baseline > gpurun_out/r2_b4.json 2> gpurun_out/r2_b4.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r2_b4.json')); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['isolated'])
for k,v in d['kernels_warmup'].items(): print(k, v)
PY
SVHIP_LIB=$PWD/exp/libsvhip_vecb.so python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('vecb bench', d['value'], d['roofline']['achieved'], d['roofline']['isolated'])"
