"""CPU study (oracle only, no device): how far does bf16 activation storage move YOLOv3 boxes from the fp32 oracle, and
does a single-rounding shortcut (fp32 add, one bf16 rounding) do better than the double rounding the device uses
(conv output rounded, sum rounded again -- DESIGN.md section 2)?  Prints min IoU / max |dscore| over all candidates that
clear the threshold by more than the margin."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import yolo_ref as R
from yolo_tensorflow_amd import darknet_io as IO

SIZE = int(os.environ.get("SIZE", "160")); NIMG = int(os.environ.get("N", "4"))


def forward_single_round(secs, params, x):
    """R.forward(emulate_bf16=True) with the conv->shortcut pair keeping the conv output in fp32 until after the add."""
    q = R.to_bf16
    x = q(np.asarray(x, np.float32)); layers = secs[1:]; outs = []; heads = []; ci = 0; pre = {}
    for i, s in enumerate(layers):
        t = s["type"]
        if t == "convolutional":
            p = params[ci]; ci += 1
            is_head = i + 1 < len(layers) and layers[i + 1]["type"] in ("yolo", "region")
            w, b = R.fold_bn(p)
            y = R.conv2d_nhwc(x, R.to_bf16(w), int(s.get("stride", 1))) + b
            if s.get("activation", "logistic") == "leaky":
                y = R.leaky_relu(y)
            y = y.astype(np.float32); pre[i] = y
            x = y if is_head else q(y)
        elif t == "shortcut":
            f = int(s["from"]); f = f if f >= 0 else i + f
            x = q(pre[i - 1] + outs[f])             # fp32 conv output + stored bf16 residual, rounded once
        elif t == "route":
            ls = [int(v) for v in s["layers"].split(",")]; ls = [l if l >= 0 else i + l for l in ls]
            x = np.concatenate([outs[l] for l in ls], -1) if len(ls) > 1 else outs[ls[0]]
        elif t == "upsample":
            x = q(R.upsample_tf(x))
        elif t in ("yolo", "region"):
            heads.append((s, outs[i - 1])); outs.append(None); continue
        else:
            raise ValueError(t)
        outs.append(x)
        pre = {k: v for k, v in pre.items() if k >= i}      # only the directly preceding conv is ever needed
    return heads


def iou(a, b):
    ix = np.maximum(0, np.minimum(a[:, 2], b[:, 2]) - np.maximum(a[:, 0], b[:, 0])); iy = np.maximum(0, np.minimum(a[:, 3], b[:, 3]) - np.maximum(a[:, 1], b[:, 1]))
    inter = ix * iy
    return inter / ((a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]) + (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]) - inter + 1e-12)


txt = IO.with_input_size(IO.cfg_text("yolov3"), SIZE)
secs = R.parse_cfg(txt); flat = IO.synth_weights(IO.parse_cfg(txt), seed=0); params = R.unflatten_weights(flat, secs)
img = np.random.default_rng(1).integers(0, 256, (NIMG, SIZE, SIZE, 3), dtype=np.uint8)
x01 = img.astype(np.float32) / np.float32(255)
ref = R.yolo_v3_detections(R.forward(secs, params, x01)[0], SIZE, ratio=True)
dbl = R.yolo_v3_detections(R.forward(secs, params, R.to_bf16(x01), emulate_bf16=True)[0], SIZE, ratio=True)
sgl = R.yolo_v3_detections(forward_single_round(secs, params, R.to_bf16(x01)), SIZE, ratio=True)
for name, det in (("double rounding (device)", dbl), ("single rounding", sgl)):
    miou, mds, cnt = 1.0, 0.0, 0
    for b in range(NIMG):
        rb, rs, rc, ridx = R.select_threshold(ref[b], 0.5)
        sc = (det[b][:, 4:5] * det[b][:, 5:]).max(-1)[ridx]
        bx = R.detections_boxes(det[b][None])[0][ridx, :4]
        ok = rs >= 0.5 + 3e-2
        if ok.any():
            miou = min(miou, float(iou(rb[ok], bx[ok]).min())); mds = max(mds, float(np.abs(rs[ok] - sc[ok]).max())); cnt += int(ok.sum())
    print("%-26s size %d, %d images, %d candidates: min IoU %.4f  max |dscore| %.4f   rel max err of decoded tensor %.3e" % (
        name, SIZE, NIMG, cnt, miou, mds, np.abs(det - ref).max() / np.abs(ref).max()))
