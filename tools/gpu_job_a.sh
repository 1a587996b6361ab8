#!/bin/bash
# round-4 GPU job A: new tests, K-loop clock, 608 retune, strong-scaling plans
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out; mkdir -p $O
python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "ragged" > $O/a_ragged.log 2>&1; tail -3 $O/a_ragged.log
python tools/kloop_clock.py r04 > $O/a_kloop.log 2>&1; tail -8 $O/a_kloop.log; cp profiles/r04_kloop_clock.json $O/ 2>/dev/null
# config 4's per-GPU share with the old plan, then retuned with the ragged halo forms available
python bench.py --size 608 --batch 8 --steps 30 --warmup 10 --no-cpu-baseline --parity-images 0 > $O/a_608_old.json 2>$O/a_608_old.err; cat $O/a_608_old.json
YOLO_TUNE_VERBOSE=1 python bench.py --size 608 --batch 8 --steps 30 --warmup 10 --no-cpu-baseline --parity-images 0 --retune > $O/a_608_new.json 2>$O/a_608_tune.err; cat $O/a_608_new.json
for b in 4 8 16; do
  python bench.py --batch $b --steps 30 --warmup 10 --no-cpu-baseline --parity-images 0 --retune > $O/a_416_b$b.json 2>$O/a_416_b$b.err; cat $O/a_416_b$b.json
done
ls $O/*.json | head -30
