"""CPU study (oracle only): which layers of YOLOv3 must be stored as split-fp16 PAIRS (22 significant bits, 3 MFMA products per
algorithmic one in the convs that read them) for the decoded boxes to stay within IoU >= 0.999 of the fp32 oracle, the rest being plain fp16
(11 bits, 1 product)?  Emulation: layer i's stored tensor is rounded to fp16 (plain) or to 22 bits (pair); a conv's folded filters are
rounded to fp16 when its input tensor is plain and to 22 bits when it is a pair (W_hi x_hi + W_hi x_lo + W_lo x_hi); a shortcut / route /
upsample output takes the type the plan gives it (the closure rule -- one type per residual stream and per concatenation -- is applied by
darknet_io.pair_closure before the emulation).  Prints, per plan, the share of the conv FLOPs that runs at 3 products and the min IoU / max
|dscore| per image.  Usage: study_mixed16.py real|log|benign [image indices]   -> profiles/r05_mixed16_study.txt"""
import glob, os, sys
import numpy as np
_H = os.path.dirname(os.path.abspath(__file__)); sys.path[:0] = [os.path.join(_H, "..", ".."), os.path.join(_H, "..", "..", "tests"), _H]
from oracle import yolo_ref as R
from yolo_tensorflow_amd import darknet_io as IO
from test_gpu_tuned import box_deviation
from PIL import Image


def q22(x):
    x = np.asarray(x, np.float32); m, e = np.frexp(x)
    return np.ldexp(np.round(m * (1 << 22)) / (1 << 22), e).astype(np.float32)


def forward_mixed(secs, params, x01, pair):
    """pair[i] True: layer i's tensor is a split-fp16 pair.  The network input is a pair iff pair[-1] (key -1)."""
    layers = secs[1:]; outs = []; heads = []; ci = 0
    qa = lambda v, i: q22(v) if pair.get(i, False) else R.to_f16(v)
    x = qa(np.asarray(x01, np.float32), -1); src = -1
    for i, s in enumerate(layers):
        t = s["type"]
        if t == "convolutional":
            p = params[ci]; ci += 1
            is_head = i + 1 < len(layers) and layers[i + 1]["type"] in ("yolo", "region")
            w, b = R.fold_bn(p); st = int(s.get("stride", 1))
            wq = q22(w) if pair.get(src_of[i], False) else R.to_f16(w)
            z = R.conv2d_nhwc(x, wq, st) + b
            if s.get("activation", "logistic") == "leaky":
                z = R.leaky_relu(z)
            z = z.astype(np.float32)
            x = z if is_head else qa(z, i)
        elif t == "shortcut":
            f = int(s["from"]); f = f if f >= 0 else i + f
            x = qa(outs[i - 1] + outs[f], i)
        elif t == "route":
            ls = [int(v) for v in s["layers"].split(",")]; ls = [l if l >= 0 else i + l for l in ls]
            x = np.concatenate([outs[l] for l in ls], -1) if len(ls) > 1 else outs[ls[0]]
        elif t == "upsample":
            x = qa(R.upsample_tf(x), i)
        elif t in ("yolo", "region"):
            heads.append((s, outs[i - 1])); outs.append(None); continue
        outs.append(x)
    return heads


stats = sys.argv[1] if len(sys.argv) > 1 else "real"
sel = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [2, 5]
txt = IO.cfg_text("yolov3"); secs = R.parse_cfg(txt); isecs = IO.parse_cfg(txt)
paths = sorted(glob.glob(os.path.join(_H, "..", "..", "tests", "golden", "images", "*.jpg")))
imgs = [np.asarray(Image.open(p).convert("RGB")) for p in paths]
x_all = np.concatenate([R.input_process(im, 416) for im in imgs])
noise = np.random.default_rng(5).random((2, 416, 416, 3), dtype=np.float32)
flat = IO.synth_weights(isecs, seed=3, stats=stats, obj_bias=-2.5 if stats in ("log", "real") else -0.75)
params = R.unflatten_weights(flat, secs)
if stats in ("log", "real"):
    R.calibrate_bn_statistics(secs, params, np.concatenate([x_all, noise]), seed=3, keep_var=stats == "real")
layers = secs[1:]; NL = len(layers)
# the tensor a conv reads (its producer layer index, -1 = the image)
src_of = {}
for i, s in enumerate(layers):
    if s["type"] == "convolutional":
        src_of[i] = i - 1
shapes = IO.layer_shapes(isecs)
flops = {i: 2.0 * int(s["size"]) ** 2 * shapes[i][4] * int(s["filters"]) * shapes[i][1] * shapes[i][2] for i, s in enumerate(layers) if s["type"] == "convolutional"}
tot = sum(flops.values())
x = x_all[sel]
ref = R.yolo_v3_detections(R.forward(secs, params, x)[0], 416, ratio=True)
print("== study_mixed16 stats=%s images=%s" % (stats, [os.path.basename(paths[k]) for k in sel]), flush=True)
plans = [("all plain fp16", set()), ("all pairs", set(range(-1, NL)))]
if os.environ.get("PLANS", "prefix") == "wide":
    for lo in (75, 62, 37, 12, 5):
        plans.append(("pairs from layer %d on" % lo, set(range(lo, NL))))
    for hi in (11, 36, 61):
        plans.append(("pairs up to layer %d" % hi, set(range(-1, hi + 1))))
    plans.append(("pairs on layers 37..86 (26 x 26 stage, 13 x 13 stage, first FPN block)", set(range(37, 87))))
else:       # what the wide sweep found: the EARLY layers are where pairs pay (noise injected early is what the stack multiplies)
    for hi in (1, 4, 8, 11, 15, 24, 36):
        plans.append(("pairs up to layer %d" % hi, set(range(-1, hi + 1))))
    plans.append(("pairs on layers 0..11 but not the image", set(range(0, 12))))
for name, want in plans:
    pair = IO.pair_closure(isecs, want)
    share = sum(f for i, f in flops.items() if pair.get(src_of[i], False)) / tot
    det = R.yolo_v3_detections(forward_mixed(secs, params, x, pair), 416, ratio=True)
    per = [box_deviation(ref[k:k + 1], det[k:k + 1], 1e-3, thr=0.4) for k in range(len(sel))]
    print("%-72s 3-product FLOP share %.2f (cost %.2f x fp16) | " % (name, share, 1 + 2 * share) + " | ".join("min IoU %.5f max|ds| %.5f lost %d" % (m[0], m[1], m[3]) for m in per), flush=True)
