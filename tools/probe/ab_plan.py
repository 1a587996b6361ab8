"""Same-box A/B of tile plans (boxes differ by several per cent, so plans are only ever compared inside one process):
interleaved rounds of yolo_time_forward for each plan, median conv / forward ms per plan.

  python tools/probe/ab_plan.py tools/probe/plan_r01_416_b32_bf16.json yolo_tensorflow_amd/tuned/yolov3_416_b32_bf16.json
"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from yolo_tensorflow_amd import hip, darknet_io as IO

B = int(os.environ.get("B", "32")); size = int(os.environ.get("SIZE", "416")); DT = os.environ.get("DTYPE", "bf16"); R = int(os.environ.get("ROUNDS", "7"))
txt = IO.with_input_size(IO.cfg_text("yolov3"), size); secs = IO.parse_cfg(txt)
eng = hip.Engine(txt, max_batch=B, dtype={"bf16": hip.BF16, "fp8": hip.FP8}[DT]); eng.set_weights(IO.synth_weights(secs, 0))
img = np.random.default_rng(0).integers(0, 256, (B, size, size, 3), dtype=np.uint8)
eng.forward(img, want_detections=False)
plans = [(os.path.basename(p), json.load(open(p))["cfgs"]) for p in sys.argv[1:]]
res = {n: [] for n, _ in plans}
for r in range(R):
    for n, cfgs in plans:
        eng.set_tile_configs(cfgs)
        eng.time_forward(B, 3)                      # settle
        res[n].append(eng.time_forward(B, 10))
for n, _ in plans:
    t = np.array(res[n])
    print("%-40s forward %.3f ms (min %.3f)  conv %.3f ms (min %.3f)  -> %.0f TFLOP/s" % (
        n, np.median(t[:, 0]), t[:, 0].min(), np.median(t[:, 1]), t[:, 1].min(), eng.conv_flops() * B / np.median(t[:, 1]) / 1e9))
