#!/bin/bash
# dual-body launches: correctness under SV_CONV_TAIL, then the tail-fraction sweep
SV_CONV_TAIL=0.2 python -m pytest tests/test_gpu_conv.py tests/test_gpu_cfg.py tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -2
run() {  # tail level cin want_scale
  r=$(SV_CONV_TAIL=$1 SV_CONV_WANT_SCALE=$4 python tools/conv_microbench.py --level $2 --cin $3 2>/dev/null | grep "level$2" | cut -c1-72)
  echo "tail=$1 want_scale=$4 $r"
}
for t in 0 0.08 0.12 0.16 0.2 0.25 0.3 0.4 0.5; do run $t 0 384 1; done
for t in 0 0.12 0.2 0.3; do run $t 0 416 1; done
for t in 0 0.2 0.3 0.4 0.5; do run $t 1 384 0.3; done
for t in 0 0.2 0.3 0.4 0.5; do run $t 1 416 0.3; done
rm -f gpurun_out/r2_wg_dual.bin
SV_CONV_TAIL=0.2 SV_CONV_TRACE=gpurun_out/r2_wg_dual.bin python tools/conv_microbench.py --level 0 --iters 1 > /dev/null 2>&1
python tools/wg_trace.py gpurun_out/r2_wg_dual.bin | grep -E "kernel span|us per step|tail|sum of"
for t in 0 0.2 0.3; do SV_CONV_TAIL=$t python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench tail=$t', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['isolated']['achieved'])"; done
