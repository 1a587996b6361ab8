import os, sys, json
import numpy as np
sys.path.insert(0, os.getcwd())
from yolo_tensorflow_amd import hip, darknet_io as IO
B = 32
txt = IO.cfg_text("yolov3"); secs = IO.parse_cfg(txt)
eng = hip.Engine(txt, max_batch=B); eng.set_weights(IO.synth_weights(secs, 0))
img = np.random.default_rng(0).integers(0, 256, (B, 416, 416, 3), dtype=np.uint8)
eng.forward(img, want_detections=False)
plan = json.load(open("yolo_tensorflow_amd/tuned/yolov3_416_b32_bf16.json"))["cfgs"]
res = {}
for rep in range(3):
    for v in (40, 10040):
        p = list(plan); p[104] = v; eng.set_tile_configs(p)
        ms = np.median([eng.time_layers(B, 20) for _ in range(5)], axis=0)
        res.setdefault(v, []).append(ms)
a = np.median(res[40], axis=0) * 1e3; b = np.median(res[10040], axis=0) * 1e3
d = b - a
print("cfg104 per-layer us, separate vs head tail: total %.1f vs %.1f" % (a.sum(), b.sum()))
for i in np.argsort(-np.abs(d))[:16]:
    print("  layer %3d: %.1f -> %.1f (%+.1f)" % (i, a[i], b[i], d[i]))
