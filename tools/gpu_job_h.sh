#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out; mkdir -p $O
for a in "--dtype fp8" "--dtype fp8 --size 608 --batch 8"; do
  python bench.py $a --steps 30 --warmup 10 --no-cpu-baseline --parity-images 0 > $O/h_old.json 2>/dev/null; grep -o '"value": [0-9.]*\|"frac": [0-9.]*' $O/h_old.json | tr '\n' ' '; echo " <- old plan: $a"
  BENCH_TUNE_ITERS=10 python bench.py $a --steps 30 --warmup 10 --no-cpu-baseline --parity-images 0 --retune > $O/h_new.json 2>$O/h_new.err; grep -o '"value": [0-9.]*\|"frac": [0-9.]*' $O/h_new.json | tr '\n' ' '; echo " <- retuned: $a"; tail -2 $O/h_new.err
done
ls $O/yolov3_*fp8.json
