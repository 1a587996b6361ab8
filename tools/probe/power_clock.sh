#!/bin/bash
# Samples socket power and shader clock while a workload runs: is the conv stack clock / power limited?
# usage: tools/probe/power_clock.sh <label> <python args...>
label=$1; shift
out=gpurun_out/power_${label}.log
python3 "$@" > gpurun_out/power_${label}_job.log 2>&1 &
job=$!
sleep ${WARM:-25}
for i in $(seq 1 ${N:-12}); do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Power" | tr -s ' \t' ' ' | tr '\n' ';' >> $out
  echo >> $out
  if [ $i = 3 ]; then amd-smi metric -g 0 --clock --power > gpurun_out/power_${label}_amdsmi.log 2>&1; fi
  sleep 0.3
done
wait $job
tail -3 gpurun_out/power_${label}_job.log
