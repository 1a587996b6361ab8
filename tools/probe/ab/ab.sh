# same-box A/B: each library variant retunes and runs the bench twice
for rep in 1 2; do
for v in A B; do
  cp tools/probe/ab/lib_$v.bin yolo_tensorflow_amd/libyolo_hip.so
  echo "$v: $(python bench.py --no-cpu-baseline --retune 2>/dev/null | python -c 'import sys,json; j=json.loads(sys.stdin.readline()); print(j["value"], j["ms_per_step"], j["roofline"]["kernel_ms_per_forward"])')"
done; done
