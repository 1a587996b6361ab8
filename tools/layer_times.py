"""Per-layer device time of a configuration at its committed tile plan: python tools/layer_times.py [dtype=bf16] [batch=32] [size=416]
Prints layer, shape, tile configuration, microseconds, GFLOP and algorithmic TFLOP/s; sums per class of layer."""
import os, sys, json
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from yolo_tensorflow_amd import hip, darknet_io as IO
dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
size = int(sys.argv[3]) if len(sys.argv) > 3 else 416
txt = IO.with_input_size(IO.cfg_text("yolov3"), size); secs = IO.parse_cfg(txt); sh = IO.layer_shapes(secs)
DT = {"bf16": hip.BF16, "fp16": hip.FP16, "fp16x2": hip.FP16X2, "fp8": hip.FP8, "fp32": hip.FP32}[dtype]
eng = hip.Engine(txt, max_batch=B, dtype=DT); eng.set_weights(IO.synth_weights(secs, 0))
img = np.random.default_rng(0).integers(0, 256, (B, size, size, 3), dtype=np.uint8)
eng.forward(img, want_detections=False)
pf = os.environ.get("PLAN") or os.path.join(ROOT, "yolo_tensorflow_amd", "tuned", "yolov3_%d_b%d_%s.json" % (size, B, "bf16" if dtype == "fp16" else dtype))
plan = [-1] * eng.num_layers
if os.path.exists(pf):
    plan = json.load(open(pf))["cfgs"]; eng.set_tile_configs(plan)
ms = np.median([eng.time_layers(B, 10) for _ in range(3)], axis=0)
L = secs[1:]; tot = 0.0; cls = {}
for i, t in enumerate(ms):
    if t <= 0.0015:
        continue
    s = L[i]; fl = 0.0; key = s["type"]
    if s["type"] == "convolutional":
        k = int(s["size"]); fl = 2.0 * k * k * sh[i][4] * int(s["filters"]) * sh[i][1] * sh[i][2] * B
        key = "%dx%d/%s @%d" % (k, k, s.get("stride", "1"), sh[i][1])
    tot += t; a = cls.setdefault(key, [0.0, 0.0, 0]); a[0] += t; a[1] += fl; a[2] += 1
    print("%3d %-14s %-26s cfg %6d %8.1f us %8.1f GFLOP %7.1f TFLOP/s" % (i, s["type"], str(sh[i][1:]), plan[i], t * 1e3, fl / 1e9, fl / (t * 1e-3) / 1e12 if fl else 0))
print("-- by class")
for k, (t, fl, n) in sorted(cls.items(), key=lambda kv: -kv[1][0]):
    print("%-22s x%-3d %8.1f us %8.1f GFLOP %7.1f TFLOP/s" % (k, n, t * 1e3, fl / 1e9, fl / (t * 1e-3) / 1e12 if fl else 0))
print("sum of per-layer events %.1f us; time_forward (total, conv) %s" % (tot * 1e3, eng.time_forward(B, 20)))
