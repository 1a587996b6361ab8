"""Experiment: one engine at batch 32 against two engines at batch 16 on two streams (the fixed costs of one stream's kernels --
prologue / epilogue HBM bursts, launch gaps -- under the other stream's K loops).  Same box, interleaved rounds."""
import os, sys, time, json
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from yolo_tensorflow_amd import hip, darknet_io as IO

txt = IO.cfg_text("yolov3"); secs = IO.parse_cfg(txt); flat = IO.synth_weights(secs, 0)
dev = torch.device("cuda", 0)
plan = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "yolo_tensorflow_amd", "tuned", "yolov3_416_b32_bf16.json")))["cfgs"]
img = torch.from_numpy(np.random.default_rng(1).integers(0, 256, (32, 416, 416, 3), dtype=np.uint8)).to(dev)


def mk(B, stream):
    e = hip.Engine(txt, max_batch=B, stream=stream.cuda_stream); e.set_weights(flat)
    e.forward(img[:B], want_detections=False); e.set_tile_configs(plan)
    boxes = torch.zeros((B, 120), dtype=torch.int32, device=dev); counts = torch.zeros((B,), dtype=torch.int32, device=dev)
    return e, boxes, counts


s0 = torch.cuda.Stream(); s1 = torch.cuda.Stream(); s2 = torch.cuda.Stream()
one = mk(32, s0); a = mk(16, s1); b = mk(16, s2)
if os.environ.get("TUNE16") == "1":
    a[0].autotune(16, 3); cfg16 = a[0].get_tile_configs(); b[0].set_tile_configs(cfg16)


def run_one(n):
    for _ in range(n):
        one[0].detect_graph(img, one[1], one[2])


OFFSET_US = float(os.environ.get("OFFSET_US", "0"))


def run_two(n):
    if OFFSET_US > 0:                       # stream B starts this much later: its layers run out of phase with stream A's
        with torch.cuda.stream(s2):
            torch.cuda._sleep(int(OFFSET_US * 2100))
    for _ in range(n):
        a[0].detect_graph(img[:16], a[1], a[2]); b[0].detect_graph(img[16:], b[1], b[2])


for fn in (run_one, run_two):
    fn(5); torch.cuda.synchronize()
res = {"one": [], "two": []}
for r in range(5):
    for name, fn in (("one", run_one), ("two", run_two)):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(20); torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t) / 20 * 1e3)
for k, v in res.items():
    print("%s: median %.3f ms per 32 images (min %.3f) -> %.0f img/s" % (k, np.median(v), min(v), 32e3 / np.median(v)))
