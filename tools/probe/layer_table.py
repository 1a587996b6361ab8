"""Per-layer device time of the tuned YOLOv3-416 batch-32 plan: layer, kind, shape, tile cfg, ms, TFLOP/s."""
import sys, os, json
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from yolo_tensorflow_amd import hip, darknet_io as IO
B = int(os.environ.get("B", "32")); size = int(os.environ.get("SIZE", "416")); DT = os.environ.get("DTYPE", "bf16")
txt = IO.with_input_size(IO.cfg_text("yolov3"), size); secs = IO.parse_cfg(txt)
eng = hip.Engine(txt, max_batch=B, dtype={"bf16": hip.BF16, "fp8": hip.FP8}[DT]); eng.set_weights(IO.synth_weights(secs, 0))
img = np.random.default_rng(0).integers(0, 256, (B, size, size, 3), dtype=np.uint8)
eng.forward(img, want_detections=False)
plan = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "yolo_tensorflow_amd", "tuned", "yolov3_%d_b%d_%s.json" % (size, B, DT))
if os.path.exists(plan) and os.environ.get("TUNE", "0") != "1":
    eng.set_tile_configs(json.load(open(plan))["cfgs"])
else:
    eng.autotune(B, int(os.environ.get("TUNE_ITERS", "3")))
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "gpurun_out"); os.makedirs(out, exist_ok=True)
    json.dump({"num_cfgs": hip.op_conv_num_cfgs(), "cfgs": [int(v) for v in eng.get_tile_configs()]}, open(os.path.join(out, "yolov3_%d_b%d_%s.json" % (size, B, DT)), "w"))
cfgs = eng.get_tile_configs()
ms = eng.time_layers(B, 20)
shapes = IO.layer_shapes(secs)                  # (type, H, W, C_out, C_in)
specs = {sp["index"]: sp for sp in IO.conv_specs(secs)}
print("layer kind            out(h,w,c)      k   cin   cfg      ms   TFLOP/s")
tot = 0.0; bykind = {}
for i, ((kind, h, w, c, cin), t) in enumerate(zip(shapes, ms)):
    tot += t
    if i in specs:
        sp = specs[i]; fl = 2.0 * B * h * w * c * sp["size"] ** 2 * sp["cin"]
        key = "conv%d" % sp["size"]
        print("%4d %-14s %4d %4d %5d  %d %5d  %4d %7.4f %8.1f" % (i, kind, h, w, c, sp["size"], sp["cin"], cfgs[i], t, fl / t / 1e9 if t > 0 else 0))
    else:
        key = kind
        if t > 0: print("%4d %-14s %4d %4d %5d                  %7.4f" % (i, kind, h, w, c, t))
    bykind[key] = bykind.get(key, 0.0) + t
print("by kind:", {k: round(v, 3) for k, v in bykind.items()})
print("sum of layers %.3f ms;  forward %.3f ms (conv %.3f ms)" % ((tot,) + tuple(eng.time_forward(B, 20))))
