"""Developer tool: per-phase cycle breakdown of the stamped conv builds on the 3x3 layer shapes of YOLOv3-416 batch 32
(YOLO_CONV_DIAG=1: tiled p176c128_s2; YOLO_CONV_DIAG=free: free-running halo form f176c256)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from yolo_tensorflow_amd import hip
rng = np.random.default_rng(0)
for mode in os.environ.get("MODES", "1,free").split(","):
    os.environ["YOLO_CONV_DIAG"] = mode
    shapes = ((32, 26, 256, 512, 3), (32, 52, 128, 256, 3), (32, 13, 512, 1024, 3))
    if os.environ.get("ONE_BY_ONE") and mode == "1":
        shapes = ((32, 26, 512, 256, 1), (32, 13, 1024, 512, 1), (32, 104, 128, 64, 1), (32, 52, 256, 128, 1))
    for (n, h, cin, cout, k) in shapes:
        x = rng.standard_normal((n, h, h, cin)).astype(np.float32)
        w = (rng.standard_normal((k, k, cin, cout)) * 0.05).astype(np.float32)
        print("mode", mode, "shape", (n, h, cin, cout, k), file=sys.stderr, flush=True)
        hip.op_conv2d(x, w, None, act=1, residual=rng.standard_normal((n, h, h, cout)).astype(np.float32), dtype={"bf16": hip.BF16, "fp8": hip.FP8}[os.environ.get("DTYPE", "bf16")])
