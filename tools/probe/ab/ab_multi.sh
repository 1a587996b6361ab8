# same-box A/B/... of several library builds (tools/probe/ab/lib_<v>.bin) with the committed tile plan: interleaved rounds in one call.
# The variant is SELECTED through YOLO_HIP_LIB (yolo_tensorflow_amd/hip.py); the in-tree libyolo_hip.so is never overwritten.
set -e
for rep in 1 2 3 4; do
for v in "$@"; do
  echo "$v: $(YOLO_HIP_LIB=$PWD/tools/probe/ab/lib_$v.bin python bench.py --no-cpu-baseline --parity-images 0 --steps 40 2>/dev/null | python -c 'import sys,json; j=json.loads(sys.stdin.readline()); print(j["value"], j["ms_per_step"], j["roofline"]["kernel_ms_per_forward"])')"
done; done
