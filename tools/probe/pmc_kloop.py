"""One long-K 3x3 conv under a chosen tile config (the K loop dominates: 288 K-steps): for LDS-conflict counters of the K loop alone."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from yolo_tensorflow_amd import hip
rng = np.random.default_rng(0)
cfg = int(os.environ.get("CFG", "40")); cin = int(os.environ.get("CIN", "2048")); cout = int(os.environ.get("COUT", "512")); h = int(os.environ.get("H", "26"))
x = rng.standard_normal((8, h, h, cin)).astype(np.float32)
w = (rng.standard_normal((3, 3, cin, cout)) * 0.02).astype(np.float32)
hip.op_conv2d(x, w, None, act=1, tile_cfg=cfg)
