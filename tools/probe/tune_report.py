import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from yolo_tensorflow_amd import hip, darknet_io as IO
os.environ["YOLO_TUNE_VERBOSE"] = "1"
B = int(os.environ.get("B", "32")); size = int(os.environ.get("SIZE", "416"))
txt = IO.with_input_size(IO.cfg_text("yolov3"), size); secs = IO.parse_cfg(txt)
DT = {"bf16": hip.BF16, "fp8": hip.FP8}[os.environ.get("DTYPE", "bf16")]
eng = hip.Engine(txt, max_batch=B, dtype=DT); eng.set_weights(IO.synth_weights(secs, 0))
img = np.random.default_rng(0).integers(0, 256, (B, size, size, 3), dtype=np.uint8)
eng.forward(img, want_detections=False)
eng.autotune(B, 10)
t, c = eng.time_forward(B, 20)
print("forward %.3f ms conv %.3f ms -> %.0f img/s, %.1f TFLOP/s" % (t, c, B / t * 1e3, eng.conv_flops() * B / c / 1e9))
