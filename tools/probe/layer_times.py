"""Device time of the first LAYERS conv launches (416 b32 bf16, committed plan) for one build of the library: LIB=path python tools/probe/layer_times.py"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from yolo_tensorflow_amd import hip, darknet_io as IO
if os.environ.get("LIB"):
    hip.LIB_PATH = os.path.abspath(os.environ["LIB"])
B = 32
txt = IO.cfg_text("yolov3"); secs = IO.parse_cfg(txt)
eng = hip.Engine(txt, max_batch=B); eng.set_weights(IO.synth_weights(secs, 0))
img = np.random.default_rng(0).integers(0, 256, (B, 416, 416, 3), dtype=np.uint8)
eng.forward(img, want_detections=False)
plan = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "yolo_tensorflow_amd", "tuned", "yolov3_416_b32_bf16.json")))["cfgs"]
eng.set_tile_configs(plan)
ms = np.median([eng.time_layers(B, 20) for _ in range(5)], axis=0)
lo, hi = [int(x) for x in os.environ.get("LAYERS", "0:12").split(":")]
print(os.environ.get("LIB", "default"), " ".join("%d:%.1f" % (i, ms[i] * 1e3) for i in range(lo, hi)))
