# same-box A/B/... of several library builds (tools/probe/ab/lib_<v>.bin) with the committed tile plan: interleaved rounds in one call
cp yolo_tensorflow_amd/libyolo_hip.so /tmp/lib_keep.so
for rep in 1 2 3 4; do
for v in "$@"; do
  cp tools/probe/ab/lib_$v.bin yolo_tensorflow_amd/libyolo_hip.so
  echo "$v: $(python bench.py --no-cpu-baseline --parity-images 0 --steps 40 2>/dev/null | python -c 'import sys,json; j=json.loads(sys.stdin.readline()); print(j["value"], j["ms_per_step"], j["roofline"]["kernel_ms_per_forward"])')"
done; done
cp /tmp/lib_keep.so yolo_tensorflow_amd/libyolo_hip.so
