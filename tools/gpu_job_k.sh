#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out; mkdir -p $O
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --parity-images 0 > $O/k_old.json 2>/dev/null; grep -o '"value": [0-9.]*\|"frac": [0-9.]*' $O/k_old.json | tr '\n' ' '; echo " <- 416 b32 committed plan"
BENCH_TUNE_ITERS=15 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --parity-images 0 --retune > $O/k_new.json 2>$O/k_tune.err; grep -o '"value": [0-9.]*\|"frac": [0-9.]*' $O/k_new.json | tr '\n' ' '; echo " <- retuned"
cp $O/yolov3_416_b32_bf16.json $O/k_plan.json
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --parity-images 0 > $O/k_old2.json 2>/dev/null; grep -o '"value": [0-9.]*\|"frac": [0-9.]*' $O/k_old2.json | tr '\n' ' '; echo " <- committed plan again"
BENCH_PLAN=$O/k_plan.json python bench.py --steps 40 --warmup 10 --no-cpu-baseline --parity-images 0 > $O/k_new2.json 2>/dev/null; grep -o '"value": [0-9.]*\|"frac": [0-9.]*' $O/k_new2.json | tr '\n' ' '; echo " <- retuned plan again"
