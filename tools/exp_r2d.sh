#!/bin/bash
# round-2 batch D: full GPU suite at HEAD (hand-written sort, probe-per-thread kernel map, new dispatch, gather depth 2 on
# thin layers), thin-layer microbenchmarks against the depth-1 build, bench + kernel trace
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
run() {  # lib level cin cout
  lib=$PWD/exp/libsvhip_$1.so; [ "$1" = tree ] && lib=$PWD/markerless-robot-camera-calibration_amd/libsvhip.so
  r=$(SVHIP_LIB=$lib python tools/conv_microbench.py --level $2 --cin $3 --cout $4 2>/dev/null | grep "level$2" | cut -c1-120)
  echo "$1 $r"
}
for spec in "1 32 32" "2 32 64" "2 64 64" "3 64 128" "0 32 32" "0 3 32" "1 384 384" "3 384 384"; do
  set -- $spec
  run vecb $1 $2 $3; run tree $1 $2 $3
done
python bench.py --steps 30 --warmup 8 > gpurun_out/r2_b5.json 2> gpurun_out/r2_b5.err; tail -3 gpurun_out/r2_b5.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r2_b5.json')); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['isolated'], d['accuracy']['labels_equal'], d['accuracy']['logits_bit_exact'])
for k,v in d['kernels_warmup'].items(): print(k, v)
PY
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2_trace2 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 8 --warmup 3 > $GRAFT_REPO_ROOT/gpurun_out/r2_trace2.json 2> $GRAFT_REPO_ROOT/gpurun_out/r2_trace2.err; cd $GRAFT_REPO_ROOT; python tools/timeline.py gpurun_out/r2_trace2/*/*_kernel_trace.csv --last-ms 160 | head -45
