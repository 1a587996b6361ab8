# same-box A/B with the committed tile plan (no retune): isolates a code change from tuner noise (variants selected through YOLO_HIP_LIB)
set -e
for rep in 1 2 3; do
for v in A B; do
  echo "$v: $(YOLO_HIP_LIB=$PWD/tools/probe/ab/lib_$v.bin python bench.py --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; j=json.loads(sys.stdin.readline()); print(j["value"], j["ms_per_step"], j["roofline"]["kernel_ms_per_forward"])')"
done; done
