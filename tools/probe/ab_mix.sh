#!/bin/bash
# same-box A/B: pair join / split on the mixed-precision FMA (in-tree) against the conversions + add / subtract of rounds 4-5 (variant -DPAIR_NO_MIX)
cd "$(dirname "$0")/../.."
run() { python bench.py --dtype fp16x2 --no-cpu-baseline --no-latency --tolerance none --parity-images 0 --no-calibration 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('fp16x2 $1: %.0f img/s, step %.3f ms, conv %.3f ms' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_forward']))"; }
for r in 1 2 3; do
  unset YOLO_HIP_LIB; run fma_mix
  export YOLO_HIP_LIB=$PWD/tools/probe/ab/lib_nomix.bin; run no_mix
done
