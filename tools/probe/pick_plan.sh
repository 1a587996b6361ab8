#!/bin/bash
# Same-box plan selection: tune several candidate tile plans (the in-situ tuner's near-ties fall differently from run to
# run), measure each one twice with the plan fixed, keep the fastest.   usage: tools/probe/pick_plan.sh [bf16|fp8] [candidates]
DT=${1:-bf16}; N=${2:-3}
PLAN=yolo_tensorflow_amd/tuned/yolov3_416_b32_$DT.json
cp $PLAN gpurun_out/plan_0.json
for k in $(seq 1 $N); do
  BENCH_TUNE_ITERS=10 python bench.py --dtype $DT --no-cpu-baseline --retune > /dev/null 2>&1
  cp gpurun_out/yolov3_416_b32_$DT.json gpurun_out/plan_$k.json
done
best=0; bestv=0
for k in $(seq 0 $N); do
  cp gpurun_out/plan_$k.json $PLAN
  v=0
  for r in 1 2; do
    x=$(python bench.py --dtype $DT --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; print(json.loads(sys.stdin.readline())["value"])')
    v=$(python -c "print($v + $x)")
  done
  echo "plan $k: mean $(python -c "print($v / 2)") img/s"
  if python -c "import sys; sys.exit(0 if $v > $bestv else 1)"; then best=$k; bestv=$v; fi
done
echo "best plan: $best"
cp gpurun_out/plan_$best.json $PLAN
cp gpurun_out/plan_$best.json gpurun_out/yolov3_416_b32_${DT}_picked.json
