"""Experiment: one engine at batch 32 against S engines at batch 32/S on S streams (the fixed costs of one stream's kernels --
prologue / epilogue HBM bursts, launch gaps -- under the other streams' K loops, and kernels that each cover 1/S of the CUs running
out of phase).  Same box, interleaved rounds.  env: S (2), PLAN (tile plan json), TUNE=1 (autotune the sub-batch engines)."""
import os, sys, time, json
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from yolo_tensorflow_amd import hip, darknet_io as IO

S = int(os.environ.get("S", "2")); BS = 32 // S
txt = IO.cfg_text("yolov3"); secs = IO.parse_cfg(txt); flat = IO.synth_weights(secs, 0)
dev = torch.device("cuda", 0)
plan = json.load(open(os.environ.get("PLAN") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "yolo_tensorflow_amd", "tuned", "yolov3_416_b32_bf16.json")))["cfgs"]
img = torch.from_numpy(np.random.default_rng(1).integers(0, 256, (32, 416, 416, 3), dtype=np.uint8)).to(dev)


def mk(B, stream):
    e = hip.Engine(txt, max_batch=B, stream=stream.cuda_stream); e.set_weights(flat)
    e.forward(img[:B], want_detections=False); e.set_tile_configs(plan)
    boxes = torch.zeros((B, 120), dtype=torch.int32, device=dev); counts = torch.zeros((B,), dtype=torch.int32, device=dev)
    return e, boxes, counts


one = mk(32, torch.cuda.Stream())
subs = [mk(BS, torch.cuda.Stream()) for _ in range(S)]
if os.environ.get("TUNE") == "1":
    subs[0][0].autotune(BS, 3); cfgs = subs[0][0].get_tile_configs()
    for e in subs[1:]:
        e[0].set_tile_configs(cfgs)
    print("sub-batch plan uses", sorted(set(int(c) for c in cfgs)))


def run_one(n):
    for _ in range(n):
        one[0].detect_graph(img, one[1], one[2])


def run_multi(n):
    for _ in range(n):
        for i, e in enumerate(subs):
            e[0].detect_graph(img[i * BS:(i + 1) * BS], e[1], e[2])


for fn in (run_one, run_multi):
    fn(5); torch.cuda.synchronize()
res = {"one": [], "multi": []}
for r in range(5):
    for name, fn in (("one", run_one), ("multi", run_multi)):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(20); torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t) / 20 * 1e3)
print("S=%d: " % S + "  ".join("%s: median %.3f ms per 32 images (min %.3f) -> %.0f img/s" % (k, np.median(v), min(v), 32e3 / np.median(v)) for k, v in res.items()))
