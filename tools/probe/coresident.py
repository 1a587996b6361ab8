"""Do kernels of two streams share a CU?  A chain of 26x26 (1x1 256 -> 3x3 512) pairs under a tile configuration that leaves room
for a second workgroup per CU (cfg 16: 4 waves, 78 KB of LDS):
  one : one engine at batch 32 (two workgroups of the SAME launch per CU)
  seq : two engines at batch 16 on ONE stream (one workgroup per CU, nothing to overlap with)
  two : two engines at batch 16 on TWO streams (one workgroup of each per CU if the dispatcher co-schedules them)"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from yolo_tensorflow_amd import hip, darknet_io as IO

CFG3 = int(os.environ.get("CFG3", "16")); PAIRS = 10


def conv(f, k, s=1, act="leaky", bn=1):
    return "[convolutional]\n%sfilters=%d\nsize=%d\nstride=%d\npad=1\nactivation=%s\n\n" % ("batch_normalize=1\n" if bn else "", f, k, s, act)


t = "[net]\nwidth=416\nheight=416\nchannels=3\n\n" + conv(32, 3) + "[maxpool]\nsize=2\nstride=2\n\n" * 4
idx = []; n = 5
for _ in range(PAIRS):
    t += conv(256, 1); n += 1
    t += conv(512, 3); idx.append(n); n += 1
t += conv(255, 1, act="linear", bn=0) + "[yolo]\nmask=0,1,2\nanchors=10,13, 16,30, 33,23\nclasses=80\nnum=3\n\n"
secs = IO.parse_cfg(t); flat = IO.synth_weights(secs, 0)
dev = torch.device("cuda", 0)
img = torch.from_numpy(np.random.default_rng(1).integers(0, 256, (32, 416, 416, 3), dtype=np.uint8)).to(dev)


def mk(B, stream):
    e = hip.Engine(t, max_batch=B, stream=stream.cuda_stream); e.set_weights(flat)
    e.forward(img[:B], want_detections=False)
    plan = np.full(e.num_layers, -1, np.int32)
    for i in idx:
        plan[i] = CFG3
    e.set_tile_configs(plan)
    boxes = torch.zeros((B, 120), dtype=torch.int32, device=dev); counts = torch.zeros((B,), dtype=torch.int32, device=dev)
    return e, boxes, counts


s0, s1, s2 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
one = mk(32, s0); a = mk(16, s1); b = mk(16, s2); c = mk(16, s1)


def run_one(k):
    for _ in range(k): one[0].detect_graph(img, one[1], one[2])
def run_seq(k):
    for _ in range(k): a[0].detect_graph(img[:16], a[1], a[2]); c[0].detect_graph(img[16:], c[1], c[2])
def run_two(k):
    for _ in range(k): a[0].detect_graph(img[:16], a[1], a[2]); b[0].detect_graph(img[16:], b[1], b[2])


fns = (("one", run_one), ("seq", run_seq), ("two", run_two))
for _, fn in fns:
    fn(5); torch.cuda.synchronize()
res = {k: [] for k, _ in fns}
for r in range(5):
    for name, fn in fns:
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(20); torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t0) / 20 * 1e3)
print("cfg %d: " % CFG3 + "  ".join("%s %.3f ms" % (k, np.median(v)) for k, v in res.items()))
