"""Small driver for rocprofv3 runs: plan YOLOv3-416 batch 32, autotune (or force one tile config), run a few forwards."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from yolo_tensorflow_amd import hip, darknet_io as IO
B = int(os.environ.get("B", "32")); size = int(os.environ.get("SIZE", "416")); iters = int(os.environ.get("ITERS", "3"))
txt = IO.with_input_size(IO.cfg_text("yolov3"), size); secs = IO.parse_cfg(txt)
eng = hip.Engine(txt, max_batch=B); eng.set_weights(IO.synth_weights(secs, 0))
img = np.random.default_rng(0).integers(0, 256, (B, size, size, 3), dtype=np.uint8)
eng.forward(img, want_detections=False)
import json
plan = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "yolo_tensorflow_amd", "tuned", "yolov3_%d_b%d_bf16.json" % (size, B))
if os.environ.get("TUNE", "0") == "1" or not os.path.exists(plan):
    eng.autotune(B, 3)
else:
    eng.set_tile_configs(json.load(open(plan))["cfgs"])
for _ in range(iters):
    eng.forward(img, want_detections=False)
eng.postprocess(B)
eng.synchronize()
