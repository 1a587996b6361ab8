#!/bin/bash
cd "$(dirname "$0")/../.."
run() { python bench.py --dtype $1 --no-cpu-baseline --no-latency --tolerance none --parity-images 0 --no-calibration 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$1 windows=$2: %.0f img/s, step %.3f ms, conv %.3f ms' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_forward']))"; }
for spec in off 1,8 1,4 3,8 5,8 11,16 11,8; do
  if [ $spec = off ]; then export YOLO_NO_WINDOWS=1; unset YOLO_WINDOWS; else unset YOLO_NO_WINDOWS; export YOLO_WINDOWS=$spec; fi
  run fp16x2 $spec
done
for spec in off 2,16 5,16 11,16 2,8; do
  if [ $spec = off ]; then export YOLO_NO_WINDOWS=1; unset YOLO_WINDOWS; else unset YOLO_NO_WINDOWS; export YOLO_WINDOWS=$spec; fi
  run bf16 $spec
done
