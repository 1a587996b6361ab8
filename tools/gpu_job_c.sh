#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fp16x2.py -x -q -m gpu -s > $O/c_x2.log 2>&1; tail -25 $O/c_x2.log
timeout 300 python bench.py --dtype fp16x2 --steps 10 --warmup 3 --no-cpu-baseline --retune > $O/c_bench_x2.json 2>$O/c_bench_x2.err; cat $O/c_bench_x2.json; tail -3 $O/c_bench_x2.err
