#!/bin/bash
# A/B experiments on the conv kernel: build exp/libsvhip_<name>.so from csrc/sv_conv.hip (or the file given as
# SRC=...) with extra compiler flags; the other objects come from the in-tree build.  Select with SVHIP_LIB=<path>.
#   tools/build_variant.sh full64 -DSV_EXP_FULL64
set -e
name=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/markerless-robot-camera-calibration_amd/csrc
mkdir -p $ROOT/exp
SRC=${SRC:-$C/sv_conv.hip}
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$ROOT/include -I$C -Wno-unused-result -ffp-contract=off \
  -mllvm -amdgpu-mfma-vgpr-form=1 "$@" -c $SRC -o $ROOT/exp/sv_conv_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/exp/libsvhip_$name.so $ROOT/exp/sv_conv_$name.o \
  $C/sv_coords.o $C/sv_sort.o $C/sv_frame.o $C/sv_post.o $C/sv_dense.o $C/sv_points.o $C/sv_icp.o $C/sv_cluster.o
echo built exp/libsvhip_$name.so
