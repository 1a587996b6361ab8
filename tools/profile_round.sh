#!/bin/bash
# Runs on the GPU box (via gpurun) from the repo root: bench line, rocprofv3 kernel trace of the same command, and
# the PMC passes (HBM traffic, MFMA busy) on tools/prof_forward.py.  Everything lands in gpurun_out/prof_$1/;
# tools/summarize_profile.py condenses that into $OUT/summary/, which is copied to profiles/ by hand afterwards.
set -u
# SIZE / BATCH (default 416 / 32 = the headline workload) select another workload, e.g. SIZE=608 BATCH=8 bash tools/profile_round.sh r04_608_b8
# (BASELINE config 4's per-GPU share): the bf16 line, its kernel trace and the PMC passes only.
R=${1:-r06}
export SIZE=${SIZE:-416} B=${BATCH:-32}
W="--size $SIZE --batch $B"
OUT=gpurun_out/prof_$R
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
if [ "$SIZE" = 416 ] && [ "$B" = 32 ]; then
python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
python3 bench.py --dtype fp8 --no-cpu-baseline --parity-images 2 > "$OUT/bench_fp8.json" 2> "$OUT/bench_fp8.err"
python3 bench.py --dtype mixed --no-cpu-baseline --parity-images 2 > "$OUT/bench_mixed.json" 2> "$OUT/bench_mixed.err"
python3 bench.py --dtype fp16 --no-cpu-baseline --parity-images 2 > "$OUT/bench_fp16.json" 2> "$OUT/bench_fp16.err"
python3 bench.py --dtype fp32 --steps 10 --warmup 3 --no-cpu-baseline --parity-images 2 > "$OUT/bench_fp32.json" 2> "$OUT/bench_fp32.err"
python3 bench.py --dtype fp16x2 --steps 20 --warmup 5 --no-cpu-baseline --parity-images 2 > "$OUT/bench_fp16x2.json" 2> "$OUT/bench_fp16x2.err"
python3 bench.py --dtype mixed16 --steps 20 --warmup 5 --no-cpu-baseline --parity-images 2 > "$OUT/bench_mixed16.json" 2> "$OUT/bench_mixed16.err"
# kernel-level traces of the other storage types (per-kernel calls / average duration: the rooflines of those lines can be recomputed from them)
for DT in fp8 mixed fp16x2 mixed16; do
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$DT" -o bench -- python3 bench.py --dtype $DT --no-cpu-baseline --no-latency --parity-images 0 --steps 20 --warmup 5 > "$OUT/bench_${DT}_under_rocprof.json" 2> "$OUT/stats_$DT.err"
done
else
python3 bench.py $W --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err"
fi
# (--tolerance none: the kernel statistics and the timeline of the HEADLINE configuration only; the tolerance line has its own trace, stats_fp16x2)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- python3 bench.py $W --no-cpu-baseline --no-latency --parity-images 0 --tolerance none > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err"
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
    D="$OUT/pmc_$(echo $C | cut -d' ' -f1)"
    ITERS=2 timeout 600 rocprofv3 --pmc $C --output-format csv -d "$D" -o p -- python3 tools/prof_forward.py > "$D.log" 2>&1
done
rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -o t -- python3 bench.py $W --no-cpu-baseline --no-latency --parity-images 0 --tolerance none --steps 10 --warmup 3 > /dev/null 2> "$OUT/trace.err"
python3 tools/step_timeline.py "$OUT/trace" > "$OUT/step_timeline.txt" 2>&1
if [ "$SIZE" = 416 ] && [ "$B" = 32 ]; then      # the tolerance line's own step, launch by launch
rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace_fp16x2" -o t -- python3 bench.py --dtype fp16x2 --no-cpu-baseline --no-latency --parity-images 0 --steps 10 --warmup 3 > /dev/null 2> "$OUT/trace_fp16x2.err"
python3 tools/step_timeline.py "$OUT/trace_fp16x2" > "$OUT/fp16x2_step_timeline.txt" 2>&1
fi
python3 tools/summarize_profile.py "$OUT" "$R" "$OUT/summary"
cp "$OUT/step_timeline.txt" "$OUT/summary/${R}_step_timeline.txt"
[ -f "$OUT/fp16x2_step_timeline.txt" ] && cp "$OUT/fp16x2_step_timeline.txt" "$OUT/summary/${R}_fp16x2_step_timeline.txt"
find "$OUT" -name "*.csv" -size +8M -delete       # gpurun_out/ is merged back only up to 64 MiB
