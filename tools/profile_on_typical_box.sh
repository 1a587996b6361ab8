# One gpurun call: probe the box with a short bench run and, if it is at least as fast as a typical box (MINV img/s, default 12300), run the
# profile rounds of round r06 (headline set + BASELINE config 4 per-GPU share) on it.  Boxes differ by up to 9 % on one binary (docs/NOTEBOOK.md).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
python3 bench.py --no-cpu-baseline --parity-images 0 --steps 30 > gpurun_out/m_probe.json 2>/dev/null
V=$(grep -o '"value": [0-9.]*' gpurun_out/m_probe.json | head -1 | cut -d' ' -f2)
echo "probe $V"
if python3 -c "import sys; sys.exit(0 if float('$V') >= ${MINV:-12300} else 1)"; then
  bash tools/profile_round.sh r06 > gpurun_out/q_r06.log 2>&1
  SIZE=608 BATCH=8 bash tools/profile_round.sh r06_608_b8 > gpurun_out/q_r06_608.log 2>&1
  grep -o '"value": [0-9.]*\|"frac": [0-9.]*' gpurun_out/prof_r06/summary/r06_bench.json | tr '\n' ' '; echo
  grep -o '"value": [0-9.]*\|"frac": [0-9.]*' gpurun_out/prof_r06_608_b8/summary/r06_608_b8_bench.json | tr '\n' ' '; echo
fi
