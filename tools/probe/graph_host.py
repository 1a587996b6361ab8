"""Developer probe: host time of one yolo_detect_graph replay (hipGraphLaunch of the ~70-node captured step) next to the device time per
step -- is the step loop host-bound?"""
import os, sys, json, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from yolo_tensorflow_amd import hip, darknet_io as IO
B = 32
txt = IO.cfg_text("yolov3"); secs = IO.parse_cfg(txt)
if os.environ.get("SIDE_STREAM"):
    torch.cuda.set_stream(torch.cuda.Stream())
eng = hip.Engine(txt, max_batch=B, stream=torch.cuda.current_stream().cuda_stream); eng.set_weights(IO.synth_weights(secs, 0))
print("stream handle", torch.cuda.current_stream().cuda_stream)
eng.set_tile_configs(json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "yolo_tensorflow_amd", "tuned", "yolov3_416_b32_bf16.json")))["cfgs"])
img = torch.from_numpy(np.random.default_rng(1).integers(0, 256, (B, 416, 416, 3), dtype=np.uint8)).cuda()
boxes = torch.zeros((B, 20 * 6), dtype=torch.int32, device="cuda"); counts = torch.zeros((B,), dtype=torch.int32, device="cuda")
for _ in range(5):
    eng.detect_graph(img, boxes, counts)
torch.cuda.synchronize()
N = 40
host = []
t0 = time.perf_counter()
for _ in range(N):
    t = time.perf_counter(); eng.detect_graph(img, boxes, counts); host.append(time.perf_counter() - t)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("host per replay: median %.1f us, max %.1f us; all %d enqueued after %.2f ms; device done after %.2f ms (%.3f ms per step)" % (
    np.median(host) * 1e6, max(host) * 1e6, N, t_enq * 1e3, t_all * 1e3, t_all / N * 1e3))
# the same step launched eagerly (yolo_detect with device buffers: ~70 plain launches, the lean decode path)
import ctypes as C
def eager():
    rc = eng.lib.yolo_detect(eng.ctx, C.c_void_p(img.data_ptr()), B, hip.IMG_U8, hip.DEVICE, 1.0 / 255.0, 0.5, 0.5, 20, hip.NMS_TF, hip.SELECT_GT,
                             C.c_void_p(boxes.data_ptr()), C.c_void_p(counts.data_ptr()), hip.DEVICE)
    assert rc == 0
for _ in range(5):
    eager()
torch.cuda.synchronize()
host = []
t0 = time.perf_counter()
for _ in range(N):
    t = time.perf_counter(); eager(); host.append(time.perf_counter() - t)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("eager: host per step: median %.1f us, max %.1f us; all %d enqueued after %.2f ms; device done after %.2f ms (%.3f ms per step)" % (
    np.median(host) * 1e6, max(host) * 1e6, N, t_enq * 1e3, t_all * 1e3, t_all / N * 1e3))
