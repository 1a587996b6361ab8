#!/bin/bash
# Build a probe variant of libyolo_hip.so WITHOUT touching the in-tree library: tools/probe/ab/build_variant.sh <name> [extra hipcc flags, e.g. -DOUT_STORE_AUX=0]
# -> tools/probe/ab/lib_<name>.bin (selected at run time through YOLO_HIP_LIB, see ab_multi.sh)
set -e
NAME=$1; shift
HERE=$(cd "$(dirname "$0")/../../.." && pwd)
SRC=$HERE/yolo_tensorflow_amd/csrc
OBJ=/tmp/abbuild/$NAME; mkdir -p "$OBJ"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-unused-value $*"
pids=()
for f in conv_igemm conv_halo13 conv_stem conv_block conv_block64 conv_f32; do /opt/rocm/bin/hipcc $FLAGS -c $SRC/$f.hip -o $OBJ/$f.o & pids+=($!); done
for f in ew_ops post_ops; do /opt/rocm/bin/hipcc $FLAGS -ffp-contract=off -c $SRC/$f.hip -o $OBJ/$f.o & pids+=($!); done
for f in yolo_api yolo_plan yolo_pack yolo_run yolo_ops yolo_dist; do /opt/rocm/bin/hipcc $FLAGS -x hip -c $SRC/$f.cpp -o $OBJ/$f.o & pids+=($!); done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $HERE/tools/probe/ab/lib_$NAME.bin $OBJ/*.o -ldl
ls -la $HERE/tools/probe/ab/lib_$NAME.bin
