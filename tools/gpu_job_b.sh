#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out; mkdir -p $O
python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "tile or cfg or ragged" > $O/b_ops.log 2>&1; tail -3 $O/b_ops.log
python bench.py --steps 30 --warmup 10 --no-cpu-baseline --parity-images 0 > $O/b_416_old.json 2>$O/b_416_old.err; cat $O/b_416_old.json
BENCH_TUNE_ITERS=10 YOLO_TUNE_VERBOSE=1 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --parity-images 0 --retune > $O/b_416_new.json 2>$O/b_416_tune.err; cat $O/b_416_new.json
cp $O/yolov3_416_b32_bf16.json $O/b_plan_416_b32.json
BENCH_TUNE_ITERS=10 YOLO_TUNE_VERBOSE=1 python bench.py --size 608 --batch 8 --steps 30 --warmup 10 --no-cpu-baseline --parity-images 0 --retune > $O/b_608_new.json 2>$O/b_608_tune.err; cat $O/b_608_new.json
