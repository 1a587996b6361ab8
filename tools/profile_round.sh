#!/bin/bash
# Runs on the GPU box (via gpurun) from the repo root: bench line, rocprofv3 kernel trace of the same command, and
# the PMC passes (HBM traffic, MFMA busy) on tools/prof_forward.py.  Everything lands in gpurun_out/prof_$1/;
# tools/summarize_profile.py condenses that into $OUT/summary/, which is copied to profiles/ by hand afterwards.
set -u
R=${1:-r02}
OUT=gpurun_out/prof_$R
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
python3 bench.py --dtype fp8 --no-cpu-baseline > "$OUT/bench_fp8.json" 2> "$OUT/bench_fp8.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- python3 bench.py --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err"
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
    D="$OUT/pmc_$(echo $C | cut -d' ' -f1)"
    ITERS=2 timeout 600 rocprofv3 --pmc $C --output-format csv -d "$D" -o p -- python3 tools/prof_forward.py > "$D.log" 2>&1
done
python3 tools/summarize_profile.py "$OUT" "$R" "$OUT/summary"
find "$OUT" -name "*.csv" -size +8M -delete       # gpurun_out/ is merged back only up to 64 MiB
