"""Study (CPU, oracle only): where does the fp8 configuration lose its box accuracy?  The shipped scheme re-quantises the residual
stream to e4m3 after every shortcut add; variant B keeps that stream in bf16 (the conv after it still reads an e4m3 copy).
YOLOv3 at a small size, synthetic weights, both against the fp32 oracle: min IoU / max |dscore| over the oracle's candidates."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import yolo_ref as R
from yolo_tensorflow_amd import darknet_io as IO

SIZE = int(os.environ.get("SIZE", "160")); N = int(os.environ.get("N", "2"))


def fp8_forward(secs, params, x, hi_stream):
    f32 = np.float32
    layers = secs[1:]; NL = len(layers)
    codes = [None] * NL; hi = [None] * NL; outs = []; heads = []
    x = R.to_bf16(np.asarray(x, np.float32)); ci = 0
    for i, s in enumerate(layers):
        t = s["type"]
        if t == "convolutional":
            p = params[ci]; ci += 1
            st = int(s.get("stride", 1))
            is_head = i + 1 < NL and layers[i + 1]["type"] in ("yolo", "region")
            w, b = R.fold_bn(p, mode="tf")
            if i == 0:
                y = R.conv2d_nhwc(x, R.to_bf16(w), st) + b
            else:
                amax = np.abs(w).max(axis=(0, 1, 2)); osc = np.where(amax > 0, amax / f32(448.0), f32(1.0)).astype(f32)
                wq = R.to_fp8_e4m3(w / osc[None, None, None, :])
                y = (R.conv2d_nhwc(codes[i - 1], wq, st).astype(np.float64) * osc + b).astype(f32)
            if s.get("activation", "logistic") == "leaky":
                y = R.leaky_relu(y)
            y = y.astype(f32)
            if is_head:
                outs.append(y); continue
            hi[i] = R.to_bf16(y); codes[i] = R.to_fp8_e4m3(hi[i])
        elif t == "shortcut":
            f = int(s["from"]); f = f if f >= 0 else i + f
            if hi_stream:
                hi[i] = R.to_bf16(hi[i - 1] + hi[f]); codes[i] = R.to_fp8_e4m3(hi[i])
            else:
                codes[i] = R.to_fp8_e4m3(codes[i - 1] + codes[f]); hi[i] = codes[i]
        elif t == "route":
            ls = [int(v) for v in s["layers"].split(",")]; ls = [l if l >= 0 else i + l for l in ls]
            codes[i] = np.concatenate([codes[l] for l in ls], -1) if len(ls) > 1 else codes[ls[0]]; hi[i] = codes[i]
        elif t == "upsample":
            codes[i] = R.to_fp8_e4m3(R.upsample_tf(codes[i - 1])); hi[i] = codes[i]
        elif t == "yolo":
            heads.append((s, outs[i - 1])); outs.append(None); continue
        else:
            raise ValueError(t)
        outs.append(codes[i])
    return heads


def iou(a, b):
    ax0, ay0, ax1, ay1 = a[:, 0] - a[:, 2] / 2, a[:, 1] - a[:, 3] / 2, a[:, 0] + a[:, 2] / 2, a[:, 1] + a[:, 3] / 2
    bx0, by0, bx1, by1 = b[:, 0] - b[:, 2] / 2, b[:, 1] - b[:, 3] / 2, b[:, 0] + b[:, 2] / 2, b[:, 1] + b[:, 3] / 2
    iw = np.maximum(0, np.minimum(ax1, bx1) - np.maximum(ax0, bx0)); ih = np.maximum(0, np.minimum(ay1, by1) - np.maximum(ay0, by0))
    inter = iw * ih
    return inter / (a[:, 2] * a[:, 3] + b[:, 2] * b[:, 3] - inter)


txt = IO.with_input_size(IO.cfg_text("yolov3"), SIZE); secs = R.parse_cfg(txt)
params = R.unflatten_weights(IO.synth_weights(IO.parse_cfg(txt), 0), secs)
img = np.random.default_rng(3).integers(0, 256, (N, SIZE, SIZE, 3), dtype=np.uint8).astype(np.float32) / np.float32(255)
t0 = time.time()
ref_heads, _ = R.forward(secs, params, img)
ref = R.yolo_v3_detections(ref_heads, SIZE, ratio=True)
print("fp32 oracle %.1f s" % (time.time() - t0))
sc_ref = ref[..., 4:5] * ref[..., 5:]; cand = sc_ref.max(-1) > 0.5
print("candidates:", int(cand.sum()))
for name, hs in (("shipped scheme (e4m3 residual stream)", False), ("bf16 residual stream", True)):
    det = R.yolo_v3_detections(fp8_forward(secs, params, img, hs), SIZE, ratio=True)
    sc = det[..., 4:5] * det[..., 5:]
    i = iou(ref[cand][:, :4], det[cand][:, :4]); ds = np.abs(sc.max(-1)[cand] - sc_ref.max(-1)[cand])
    print("%-40s min IoU %.4f  mean IoU %.4f  max |dscore| %.4f  mean |dscore| %.4f" % (name, i.min(), i.mean(), ds.max(), ds.mean()))
