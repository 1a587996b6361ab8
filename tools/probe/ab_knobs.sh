#!/bin/bash
# same-box A/B of the pair K loop's knobs (variants built by tools/probe/ab/build_variant.sh <name> -DPAIR_PD=.. / -DPAIR_SLEEP=..)
cd "$(dirname "$0")/../.."
run() { python bench.py --dtype fp16x2 --no-cpu-baseline --no-latency --tolerance none --parity-images 0 --no-calibration 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('fp16x2 $1: %.0f img/s, step %.3f ms, conv %.3f ms' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_forward']))"; }
for r in 1 2; do
  unset YOLO_HIP_LIB; run in-tree
  for v in ${VARIANTS:-pd6 pd2 sl32 sl64}; do export YOLO_HIP_LIB=$PWD/tools/probe/ab/lib_$v.bin; run $v; done
done
