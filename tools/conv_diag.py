"""Developer tool: per-phase cycle breakdown of the stamped conv build on one layer shape (YOLO_CONV_DIAG=1)."""
import os, sys
import numpy as np
os.environ["YOLO_CONV_DIAG"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from yolo_tensorflow_amd import hip
rng = np.random.default_rng(0)
for (n, h, cin, cout, k) in ((32, 26, 256, 512, 3), (32, 52, 128, 256, 3), (32, 13, 512, 1024, 3), (32, 26, 512, 256, 1)):
    x = rng.standard_normal((n, h, h, cin)).astype(np.float32)
    w = (rng.standard_normal((k, k, cin, cout)) * 0.05).astype(np.float32)
    print("shape", (n, h, cin, cout, k), file=sys.stderr)
    hip.op_conv2d(x, w, None, act=1, dtype={"bf16": hip.BF16, "fp8": hip.FP8}[os.environ.get("DTYPE", "bf16")])
