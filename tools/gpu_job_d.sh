#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out; mkdir -p $O
python -m pytest tests/test_gpu_ops.py tests/test_gpu_fp16.py -x -q -m gpu -k "tile or cfg or ragged or halo" > $O/d_ops.log 2>&1; tail -3 $O/d_ops.log
python bench.py --steps 30 --warmup 10 --no-cpu-baseline --parity-images 0 > $O/d_416_old.json 2>$O/d_416_old.err; cut -c1-120 $O/d_416_old.json; grep -o '"frac": [0-9.]*' $O/d_416_old.json
BENCH_TUNE_ITERS=10 YOLO_TUNE_VERBOSE=1 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --parity-images 0 --retune > $O/d_416_new.json 2>$O/d_416_tune.err; grep -o '"value": [0-9.]*\|"frac": [0-9.]*' $O/d_416_new.json
cp $O/yolov3_416_b32_bf16.json $O/d_plan_416_b32.json
grep "cfg 52 \|cfg 40 " $O/d_416_tune.err
