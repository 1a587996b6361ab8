#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fp16x2.py -x -q -m gpu -s > $O/g_x2.log 2>&1; tail -12 $O/g_x2.log
timeout 300 python bench.py --dtype fp16x2 --steps 10 --warmup 3 --no-cpu-baseline --retune > $O/g_bench_x2.json 2>$O/g_bench_x2.err; cut -c1-200 $O/g_bench_x2.json; grep -o '"kernel_ms_per_forward": [0-9.]*\|"min_iou": [0-9.]*' $O/g_bench_x2.json; tail -3 $O/g_bench_x2.err
