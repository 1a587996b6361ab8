import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from yolo_tensorflow_amd import hip, darknet_io as IO
txt = IO.with_input_size(IO.cfg_text("yolov3"), 160); secs = IO.parse_cfg(txt); flat = IO.synth_weights(secs, seed=21)
img = np.random.default_rng(22).integers(0, 256, (2, 160, 160, 3), dtype=np.uint8)
eng = hip.Engine(txt, max_batch=2, dtype=hip.FP16X2, keep_layers=True); eng.set_weights(flat)
d1 = eng.forward(img); a = {i: eng.layer_output(i, 2) for i in (84, 85, 86, 87, 96, 97, 98)}
d1b = eng.forward(img); print("repeatable", np.array_equal(d1, d1b))
os.environ["YOLO_PAIR_UPSAMPLE_VIA_F32"] = "1"
d2 = eng.forward(img); b = {i: eng.layer_output(i, 2) for i in a}
for i in a:
    print(i, a[i].shape, np.array_equal(a[i], b[i]), float(np.abs(a[i] - b[i]).max()), np.argwhere(a[i] != b[i])[:4].tolist())
