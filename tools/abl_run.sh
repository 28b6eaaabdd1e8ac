#!/bin/bash
# tools/abl_run.sh "<variant names>": E1 (equal full tiles) and E6 (room-like masks) of tools/tile_shape_experiment.py
# for the in-tree library and every exp/libsvhip_<name>.so (timing-only ablation builds, tools/build_variant.sh)
export SV_CONV_TAIL=0 TILE_EXPS="E1 E6"
echo "== in-tree"; python tools/tile_shape_experiment.py 64 8 2>/dev/null | grep "^E"
for l in $1; do
  echo "== $l"; SVHIP_LIB=$PWD/exp/libsvhip_$l.so python tools/tile_shape_experiment.py 64 8 2>/dev/null | grep "^E"
done
