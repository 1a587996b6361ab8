#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out; mkdir -p $O
python bench.py --size 608 --batch 8 --steps 30 --warmup 10 --no-cpu-baseline --parity-images 0 > $O/j_old.json 2>/dev/null; grep -o '"value": [0-9.]*\|"frac": [0-9.]*' $O/j_old.json | tr '\n' ' '; echo " <- 608 b8 committed plan"
BENCH_TUNE_ITERS=10 YOLO_TUNE_VERBOSE=1 python bench.py --size 608 --batch 8 --steps 30 --warmup 10 --no-cpu-baseline --parity-images 0 --retune > $O/j_new.json 2>$O/j_tune.err; grep -o '"value": [0-9.]*\|"frac": [0-9.]*' $O/j_new.json | tr '\n' ' '; echo " <- retuned"
grep "19_19_512_1024_3_1_00_1 cfg \(49\|52\|53\|23\|48\) " $O/j_tune.err
python -m pytest tests/test_gpu_natural.py -x -q -m gpu -s -k "32_images" 2>&1 | grep "log-statistics"
