#!/bin/bash
# round-4 evidence: the headline profile set, config 4's per-GPU share, kernel-level stats of the other dtypes
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
bash tools/profile_round.sh r04 > gpurun_out/f_r04.log 2>&1
SIZE=608 BATCH=8 bash tools/profile_round.sh r04_608_b8 > gpurun_out/f_r04_608.log 2>&1
for dt in fp8 mixed fp16x2; do
  D=gpurun_out/prof_r04_$dt; rm -rf $D; mkdir -p $D
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -o bench -- python3 bench.py --dtype $dt --no-cpu-baseline --parity-images 0 --steps 20 --warmup 5 > $D/bench_under_rocprof.json 2> $D/stats.err
  find $D -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r04_${dt}_kernel_stats.csv
  cp $D/bench_under_rocprof.json gpurun_out/r04_${dt}_bench_under_rocprof.json
  find $D -name "*.csv" -size +4M -delete
done
cat gpurun_out/prof_r04/summary/r04_bench.json | head -c 1500; echo
cat gpurun_out/prof_r04_608_b8/summary/r04_608_b8_bench.json | head -c 900; echo
ls gpurun_out/prof_r04/summary gpurun_out/prof_r04_608_b8/summary
