#!/bin/bash
# plain (write-back) output stores + image windows: does the Infinity Cache keep a producer's tensor for its consumer when the stores are not write-through?
cd "$(dirname "$0")/../.."
run() { python bench.py --dtype $1 --no-cpu-baseline --no-latency --tolerance none --parity-images 0 --no-calibration 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$1 $3 windows=$2: %.0f img/s, step %.3f ms, conv %.3f ms' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_forward']))"; }
for lib in sc1 plain; do
  if [ $lib = plain ]; then export YOLO_HIP_LIB=$PWD/tools/probe/ab/lib_plainst.bin; else unset YOLO_HIP_LIB; fi
  for spec in off 11,4 11,8 3,8; do
    if [ $spec = off ]; then export YOLO_NO_WINDOWS=1; unset YOLO_WINDOWS; else unset YOLO_NO_WINDOWS; export YOLO_WINDOWS=$spec; fi
    run fp16x2 $spec $lib
  done
  for spec in off 5,8 11,8; do
    if [ $spec = off ]; then export YOLO_NO_WINDOWS=1; unset YOLO_WINDOWS; else unset YOLO_NO_WINDOWS; export YOLO_WINDOWS=$spec; fi
    run bf16 $spec $lib
  done
done
