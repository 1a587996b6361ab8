"""Race hunt: the tuned production plan (fused stem, halo conv, 1x1 tails, alias-scoped LDS stages) must give bit-identical
detections on every one of many repeated forwards, eager and graph-replayed, bf16, fp8 and split fp16 (whose epilogue makes two barrier-separated
passes through one LDS tile)."""
import sys, os, json
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from yolo_tensorflow_amd import hip, darknet_io as IO
N = int(os.environ.get("REPS", "300")); B = 32
txt = IO.cfg_text("yolov3"); secs = IO.parse_cfg(txt); flat = IO.synth_weights(secs, 0)
img = np.random.default_rng(0).integers(0, 256, (B, 416, 416, 3), dtype=np.uint8)
dimg = torch.from_numpy(img).cuda()
for name, dt in (("bf16", hip.BF16), ("fp8", hip.FP8), ("fp16x2", hip.FP16X2)):
    eng = hip.Engine(txt, max_batch=B, dtype=dt)
    eng.set_weights(flat)
    plan = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "yolo_tensorflow_amd", "tuned", "yolov3_416_b32_%s.json" % name)
    eng.set_tile_configs(json.load(open(plan))["cfgs"])
    ref = eng.forward(dimg).copy()
    bad = 0
    for i in range(N):
        out = eng.forward(dimg)
        if not np.array_equal(out, ref):
            bad += 1
    boxes = torch.zeros((B, 120), dtype=torch.int32, device="cuda"); counts = torch.zeros((B,), dtype=torch.int32, device="cuda")
    eng.detect_graph(dimg, boxes, counts); eng.detect_graph(dimg, boxes, counts); eng.synchronize()
    b0, c0 = boxes.cpu().numpy().copy(), counts.cpu().numpy().copy()
    gbad = 0
    for i in range(N):
        eng.detect_graph(dimg, boxes, counts)
        if i % 10 == 9:
            eng.synchronize()
            if not (np.array_equal(boxes.cpu().numpy(), b0) and np.array_equal(counts.cpu().numpy(), c0)):
                gbad += 1
    print("%s: %d eager forwards, %d mismatches; %d graph replays, %d mismatching checks" % (name, N, bad, N, gbad))
    assert bad == 0 and gbad == 0
    eng.close()
print("deterministic")
