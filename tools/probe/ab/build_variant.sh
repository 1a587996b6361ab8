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
# every source of the library, as buildinfo.source_hash and the Makefile see them (a hand-kept list went stale: ADVICE r05)
for src in $SRC/*.hip; do f=$(basename $src .hip)
  case $f in ew_ops|post_ops) X="-ffp-contract=off";; *) X="";; esac
  /opt/rocm/bin/hipcc $FLAGS $X -c $src -o $OBJ/$f.o & pids+=($!); done
for src in $SRC/yolo_*.cpp; do f=$(basename $src .cpp); /opt/rocm/bin/hipcc $FLAGS -x hip -c $src -o $OBJ/$f.o & pids+=($!); done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $HERE/tools/probe/ab/lib_$NAME.bin $OBJ/*.o -ldl
ls -la $HERE/tools/probe/ab/lib_$NAME.bin
