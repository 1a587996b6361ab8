"""Per-layer device time under the committed 416 b32 bf16 plan with some layers' tile configs overridden: LAYERS=7:10041,10:41"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from yolo_tensorflow_amd import hip, darknet_io as IO
B = 32
txt = IO.cfg_text("yolov3"); secs = IO.parse_cfg(txt)
eng = hip.Engine(txt, max_batch=B); eng.set_weights(IO.synth_weights(secs, 0))
img = np.random.default_rng(0).integers(0, 256, (B, 416, 416, 3), dtype=np.uint8)
eng.forward(img, want_detections=False)
plan = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "yolo_tensorflow_amd", "tuned", "yolov3_416_b32_bf16.json")))["cfgs"]
show = set()
for variant in ("base", "override"):
    p = list(plan)
    if variant == "override":
        for kv in os.environ.get("LAYERS", "").split(","):
            if kv:
                i, c = kv.split(":"); p[int(i)] = int(c); show.update(range(int(i) - 1, int(i) + 4))
    eng.set_tile_configs(p)
    ms = np.median([eng.time_layers(B, 20) for _ in range(3)], axis=0)
    f = eng.time_forward(B, 20)
    print(variant, "forward %.3f conv %.3f |" % tuple(f), " ".join("%d:%.1f" % (i, ms[i] * 1e3) for i in sorted(show or range(4, 12))))
