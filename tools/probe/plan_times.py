"""Conv time per forward (416 b32 bf16) under plan files: python tools/probe/plan_times.py planA.json planB.json ... ; prints per-layer times for LAYERS=lo:hi"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from yolo_tensorflow_amd import hip, darknet_io as IO
B = 32
txt = IO.cfg_text("yolov3"); secs = IO.parse_cfg(txt)
eng = hip.Engine(txt, max_batch=B); eng.set_weights(IO.synth_weights(secs, 0))
img = np.random.default_rng(0).integers(0, 256, (B, 416, 416, 3), dtype=np.uint8)
eng.forward(img, want_detections=False)
lo, hi = [int(x) for x in os.environ.get("LAYERS", "0:12").split(":")]
for rep in range(2):
    for path in sys.argv[1:]:
        eng.set_tile_configs(json.load(open(path))["cfgs"])
        eng.forward(img, want_detections=False)
        ms = np.median([eng.time_layers(B, 20) for _ in range(3)], axis=0)
        f = eng.time_forward(B, 30)
        print(os.path.basename(path), "forward %.3f conv %.3f |" % tuple(f), " ".join("%d:%.1f" % (i, ms[i] * 1e3) for i in range(lo, hi)))
