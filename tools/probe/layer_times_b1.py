"""Per-layer device time at batch 1 (416, bf16, committed b1 plan): where the 0.97 ms of `latency_b1_ms` goes."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from yolo_tensorflow_amd import hip, darknet_io as IO
B = int(os.environ.get("B", "1"))
txt = IO.cfg_text("yolov3"); secs = IO.parse_cfg(txt); sh = IO.layer_shapes(secs)
eng = hip.Engine(txt, max_batch=B); eng.set_weights(IO.synth_weights(secs, 0))
img = np.random.default_rng(0).integers(0, 256, (B, 416, 416, 3), dtype=np.uint8)
eng.forward(img, want_detections=False)
plan = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "yolo_tensorflow_amd", "tuned", "yolov3_416_b%d_bf16.json" % B)))["cfgs"]
eng.set_tile_configs(plan)
ms = np.median([eng.time_layers(B, 30) for _ in range(5)], axis=0)
tot = 0.0
for i, t in enumerate(ms):
    if t > 0.0015:
        tot += t
        print("%3d %-14s %-24s cfg %6d  %7.1f us" % (i, sh[i][0], str(sh[i][1:]), plan[i], t * 1e3))
print("sum of per-layer events %.1f us; time_forward %s" % (tot * 1e3, eng.time_forward(B, 50)))
