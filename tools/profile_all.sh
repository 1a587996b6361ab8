#!/bin/bash
# round-4 evidence in one gpurun call: the headline profile set, config 4 per-GPU share, the in-kernel K-loop clock
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
bash tools/profile_round.sh r04 > gpurun_out/prof_r04.log 2>&1
SIZE=608 BATCH=8 bash tools/profile_round.sh r04_608_b8 > gpurun_out/prof_r04_608.log 2>&1
python tools/kloop_clock.py r04 > gpurun_out/kloop.log 2>&1; cp profiles/r04_kloop_clock.json gpurun_out/
grep -o '"value": [0-9.]*\|"frac": [0-9.]*\|"traffic": [0-9a-z]*' gpurun_out/prof_r04/summary/r04_bench.json | tr '\n' ' '; echo
grep -o '"value": [0-9.]*\|"frac": [0-9.]*' gpurun_out/prof_r04_608_b8/summary/r04_608_b8_bench.json | tr '\n' ' '; echo
ls gpurun_out/prof_r04/summary | wc -l; ls gpurun_out/prof_r04_608_b8/summary | wc -l
