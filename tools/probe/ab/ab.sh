# same-box A/B: each library variant (selected through YOLO_HIP_LIB, never copied over the in-tree library) retunes and runs the bench twice
set -e
for rep in 1 2; do
for v in A B; do
  echo "$v: $(YOLO_HIP_LIB=$PWD/tools/probe/ab/lib_$v.bin python bench.py --no-cpu-baseline --retune 2>/dev/null | python -c 'import sys,json; j=json.loads(sys.stdin.readline()); print(j["value"], j["ms_per_step"], j["roofline"]["kernel_ms_per_forward"])')"
done; done
