#!/bin/bash
# A/B of the image-window schedule (yolo_run.cpp window_plan) on one box: bench lines with and without it, interleaved
cd "$(dirname "$0")/../.."
for round in 1 2; do
for dt in bf16 fp16x2; do
  for w in on off; do
    if [ $w = off ]; then export YOLO_NO_WINDOWS=1; else unset YOLO_NO_WINDOWS; fi
    python bench.py --dtype $dt --no-cpu-baseline --no-latency --tolerance none --parity-images 1 --no-calibration $EXTRA 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$dt windows $w: %.0f img/s, step %.3f ms, conv %.3f ms, parity %s' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_forward'], d['parity']['min_iou']))"
  done
done
done
