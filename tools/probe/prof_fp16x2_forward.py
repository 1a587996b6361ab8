"""rocprofv3 driver: a few YOLOv3-416 batch-32 forwards of the split-fp16 configuration at its committed plan (for --pmc passes per kernel)."""
import sys, os, json
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from yolo_tensorflow_amd import hip, darknet_io as IO
B = 32; txt = IO.cfg_text("yolov3"); secs = IO.parse_cfg(txt)
eng = hip.Engine(txt, max_batch=B, dtype=hip.FP16X2); eng.set_weights(IO.synth_weights(secs, 0))
img = np.random.default_rng(0).integers(0, 256, (B, 416, 416, 3), dtype=np.uint8)
eng.forward(img, want_detections=False)
eng.set_tile_configs(json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "yolo_tensorflow_amd", "tuned", "yolov3_416_b32_fp16x2.json")))["cfgs"])
for _ in range(2):
    eng.forward(img, want_detections=False)
eng.synchronize()
