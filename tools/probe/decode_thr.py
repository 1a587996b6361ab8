"""Developer probe: whole-step time of the graph-replayed detect path at several score thresholds (what the objectness pre-filter of the
lean decode lets through), with the one-launch multi-head decode (default) and with the per-head launches (YOLO_NO_LEAN_MULTI=1)."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from yolo_tensorflow_amd import hip, darknet_io as IO
B = 32
txt = IO.cfg_text("yolov3"); secs = IO.parse_cfg(txt)
eng = hip.Engine(txt, max_batch=B, stream=torch.cuda.current_stream().cuda_stream); eng.set_weights(IO.synth_weights(secs, 0))
plan = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "yolo_tensorflow_amd", "tuned", "yolov3_416_b32_bf16.json")))
eng.set_tile_configs(plan["cfgs"])
img = torch.from_numpy(np.random.default_rng(1).integers(0, 256, (B, 416, 416, 3), dtype=np.uint8)).cuda()
boxes = torch.zeros((B, 20 * 6), dtype=torch.int32, device="cuda"); counts = torch.zeros((B,), dtype=torch.int32, device="cuda")
det = eng.forward(img.cpu().numpy())
obj = det[..., 4]
print("objectness > 0.5: %.3f of the boxes; > 0.9: %.3f; max score > 0.5: %.4f" % ((obj > 0.5).mean(), (obj > 0.9).mean(), ((det[..., 4:5] * det[..., 5:]).max(-1) > 0.5).mean()))
for thr in (0.5, 0.9, 0.999):
    for _ in range(5):
        eng.detect_graph(img, boxes, counts, score_thr=thr, iou_thr=0.5, max_out=20)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(30):
        eng.detect_graph(img, boxes, counts, score_thr=thr, iou_thr=0.5, max_out=20)
    e1.record(); torch.cuda.synchronize()
    print("thr %.3f: %.4f ms per step, kept %d" % (thr, e0.elapsed_time(e1) / 30, int(counts.sum())))
