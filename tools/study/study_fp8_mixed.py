"""CPU study (oracle emulation of the device's fp8 scheme, no device): which layers of YOLOv3 have to leave e4m3 for the boxes to come
within IoU 0.97 of the fp32 reference?  Candidate plans = sets of conv layers stored in bf16 (cfg key yolo_store, closed under the
one-type-per-residual-stream rule, darknet_io.store_closure).  Prints per plan: the convs stored in bf16, the share of the conv FLOPs
that then runs on the bf16 MFMA (half the e4m3 rate), min / mean IoU and max |dscore| over the fp32 oracle's candidates."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
from oracle import yolo_ref as R
from yolo_tensorflow_amd import darknet_io as IO

SIZE = int(os.environ.get("SIZE", "416")); NIMG = int(os.environ.get("N", "2"))
txt0 = IO.cfg_text("yolov3") if SIZE == 416 else IO.with_input_size(IO.cfg_text("yolov3"), SIZE)
secs0 = IO.parse_cfg(txt0)
flat = IO.synth_weights(secs0, seed=0)
img = np.random.default_rng(1).integers(0, 256, (NIMG, SIZE, SIZE, 3), dtype=np.uint8)
x01 = img.astype(np.float32) / np.float32(255)
osecs0 = R.parse_cfg(txt0); params = R.unflatten_weights(flat, osecs0)
ref = R.yolo_v3_detections(R.forward(osecs0, params, x01)[0], SIZE, ratio=True)
shapes = IO.layer_shapes(secs0)
convs = [i for i, s in enumerate(secs0[1:]) if s["type"] == "convolutional"]
flops = {i: 2.0 * int(secs0[1:][i]["size"]) ** 2 * shapes[i][4] * shapes[i][3] * shapes[i][1] * shapes[i][2] for i in convs}
tot = sum(flops.values())


def iou(a, b):
    ix = np.maximum(0, np.minimum(a[:, 2], b[:, 2]) - np.maximum(a[:, 0], b[:, 0])); iy = np.maximum(0, np.minimum(a[:, 3], b[:, 3]) - np.maximum(a[:, 1], b[:, 1]))
    inter = ix * iy
    return inter / ((a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]) + (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]) - inter + 1e-12)


def evaluate(name, want):
    S = IO.store_closure(secs0, want)
    txt = IO.with_layer_store(txt0, S)
    osecs = R.parse_cfg(txt)
    heads, _ = R.fp8_scheme_forward(osecs, params, x01)
    det = R.yolo_v3_detections(heads, SIZE, ratio=True)
    ious, ds = [], []
    for b in range(NIMG):
        rb, rs, rc, ridx = R.select_threshold(ref[b], 0.5)
        d = det[b][ridx]
        bx = np.stack([d[:, 0] - d[:, 2] / 2, d[:, 1] - d[:, 3] / 2, d[:, 0] + d[:, 2] / 2, d[:, 1] + d[:, 3] / 2], -1)
        ious.append(iou(rb, bx)); ds.append(np.abs(rs - (d[:, 4:5] * d[:, 5:]).max(-1)))
    ious = np.concatenate(ious); ds = np.concatenate(ds)
    # a conv runs on the bf16 MFMA when its INPUT is stored in bf16: the consumers of the layers in S (and layer 0)
    L = secs0[1:]
    def is16(i):
        t = L[i]["type"]
        if t == "convolutional": return i in S
        if t in ("yolo", "region"): return False
        if t == "shortcut":
            return is16(i - 1)
        if t == "route":
            ls = [int(v) if int(v) >= 0 else i + int(v) for v in L[i]["layers"].split(",")]
            return is16(ls[0])
        return is16(i - 1)
    b16 = sum(flops[i] for i in convs if i == 0 or is16(i - 1))
    print("%-34s %3d convs stored bf16, %5.1f %% of FLOPs on the bf16 MFMA | %4d candidates: min IoU %.4f  mean %.4f  max |dscore| %.4f" % (
        name, len(S), 100.0 * b16 / tot, len(ious), ious.min(), ious.mean(), ds.max()), flush=True)
    return S


heads_in = [i for i in convs if secs0[1:][i + 1]["type"] in ("yolo",)]          # the head convs themselves (fp32 out)
plans = [("all e4m3", []),
         ("convs feeding the heads", [i - 1 for i in heads_in]),
         ("last 3 convs before each head", [j for i in heads_in for j in (i - 1, i - 2, i - 3)]),
         ("FPN blocks (75-80, 84-92, 96-104)", [i for i in convs if i >= 75 and i not in heads_in]),
         ("13x13 stage + FPN (62-104)", [i for i in convs if i >= 62 and i not in heads_in]),
         ("26x26 stage on + FPN (37-104)", [i for i in convs if i >= 37 and i not in heads_in]),
         ("residual streams only (3x3 outputs)", [i for i in convs if secs0[1:][i + 1]["type"] == "shortcut"]),
         ("everything but 104/208 stages", [i for i in convs if i >= 12 and i not in heads_in]),
         ("all bf16 storage", [i for i in convs if i not in heads_in])]
sel = os.environ.get("PLANS")
for k, (name, want) in enumerate(plans):
    if sel and str(k) not in sel.split(","):
        continue
    evaluate(name, want)
