#!/bin/bash
# same-box A/B of the fused split-fp16 stem (conv_stem_pair.hip) against the two launches it replaces
cd "$(dirname "$0")/../.."
run() { python bench.py --dtype fp16x2 --no-cpu-baseline --no-latency --tolerance none --parity-images 0 --no-calibration 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('fp16x2 stem $1: %.0f img/s, step %.3f ms, conv %.3f ms' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_forward']))"; }
for r in 1 2 3; do
  unset YOLO_NO_PAIR_STEM; run fused
  export YOLO_NO_PAIR_STEM=1; run separate
done
