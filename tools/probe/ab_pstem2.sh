#!/bin/bash
# same-box A/B: producer / consumer pair stem (default) vs the phase-serial kernel (YOLO_PAIR_STEM_V1=1) vs the separate launches
cd "$(dirname "$0")/../.."
run() { python bench.py --dtype fp16x2 --no-cpu-baseline --no-latency --tolerance none --parity-images 0 --no-calibration 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('fp16x2 stem $1: %.0f img/s, step %.3f ms, conv %.3f ms' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_forward']))"; }
for r in 1 2 3; do
  unset YOLO_PAIR_STEM_V1 YOLO_NO_PAIR_STEM; run producer_consumer
  export YOLO_PAIR_STEM_V1=1; run phase_serial; unset YOLO_PAIR_STEM_V1
done
python tools/layer_times.py fp16x2 2>/dev/null | grep -E "^  1 "
YOLO_PAIR_STEM_V1=1 python tools/layer_times.py fp16x2 2>/dev/null | grep -E "^  1 "
