#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; tail -4 $O/gpu_tests.log
for b in 4 8 16; do
  BENCH_TUNE_ITERS=10 python bench.py --batch $b --steps 30 --warmup 10 --no-cpu-baseline --parity-images 0 --retune > $O/e_416_b$b.json 2>$O/e_416_b$b.err; grep -o '"value": [0-9.]*\|"frac": [0-9.]*' $O/e_416_b$b.json | tr '\n' ' '; echo
done
