"""CPU study (oracle only, no device): where does reduced-precision STORAGE move YOLOv3 boxes away from the fp32 reference, on natural
images (the reference's own jpgs, tests/golden/images) and for both synthetic weight flavours -- benign batch-norm statistics and the
trained-file ranges of D2T/log.txt calibrated on those images (oracle.calibrate_bn_statistics)?

Schemes (all with fp32 accumulation and bf16-rounded folded filters, like the device):
  dev      the device's scheme: every stored activation rounded to bf16, the conv output rounded before the shortcut add, the sum rounded
  trunk32  a true fp32 residual trunk: the shortcut stream is kept in fp32 (conv output added unrounded, sum not rounded); convs that
           READ the stream see its bf16 rounding
  in0      the first conv computed exactly (fp32 pixels x / 255 and fp32 filters); everything else as `dev`
  both     trunk32 + in0
  f16      `dev` with IEEE fp16 storage (11-bit significand) instead of bf16 (8-bit) for activations and filters
Prints per scheme: relative rms error of the three raw head tensors, min IoU / max |dscore| over the oracle's candidates (score > 0.4 by
more than 1e-2), candidates lost below the threshold."""
import glob, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
from oracle import yolo_ref as R
from yolo_tensorflow_amd import darknet_io as IO
from test_gpu_tuned import box_deviation

NIMG = int(os.environ.get("N", "3"))


def to_f16(x):
    return np.asarray(x, np.float32).astype(np.float16).astype(np.float32)


def forward_scheme(secs, params, x01, q, trunk32=False, in0=False):
    layers = secs[1:]; outs = []; heads = []; ci = 0
    x = np.asarray(x01, np.float32) if in0 else q(np.asarray(x01, np.float32))
    exact = {}                                   # fp32 value of a stream tensor whose stored copy is rounded (trunk32)
    for i, s in enumerate(layers):
        t = s["type"]
        if t == "convolutional":
            p = params[ci]; ci += 1
            is_head = i + 1 < len(layers) and layers[i + 1]["type"] in ("yolo", "region")
            w, b = R.fold_bn(p)
            y = R.conv2d_nhwc(x, w if (in0 and i == 0) else q(w), int(s.get("stride", 1))) + b
            if s.get("activation", "logistic") == "leaky":
                y = R.leaky_relu(y)
            y = y.astype(np.float32)
            if trunk32:
                exact[i] = y
            x = y if is_head else q(y)
        elif t == "shortcut":
            f = int(s["from"]); f = f if f >= 0 else i + f
            if trunk32:
                e = exact[i - 1] + exact.get(f, outs[f]); exact[i] = e; x = q(e)
            else:
                x = q(outs[i - 1] + outs[f])
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
        exact = {k: v for k, v in exact.items() if k >= i - 3}
    return heads


txt = IO.cfg_text("yolov3"); secs = R.parse_cfg(txt)
paths = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden", "images", "*.jpg")))
from PIL import Image
imgs = [np.asarray(Image.open(p).convert("RGB")) for p in paths]
x_all = np.concatenate([R.input_process(im, 416) for im in imgs])
for stats in ("benign", "log"):
    flat = IO.synth_weights(IO.parse_cfg(txt), seed=3, stats=stats, obj_bias=-2.5 if stats == "log" else -0.75)
    params = R.unflatten_weights(flat, secs)
    if stats == "log":
        R.calibrate_bn_statistics(secs, params, np.concatenate([x_all, np.random.default_rng(5).random((2, 416, 416, 3), dtype=np.float32)]), seed=3)
    x = x_all[:NIMG]
    h32 = R.forward(secs, params, x)[0]
    ref = R.yolo_v3_detections(h32, 416, ratio=True)
    print("== %s weights, %d natural images" % (stats, NIMG))
    for name, kw in (("dev", dict(q=R.to_bf16)), ("trunk32", dict(q=R.to_bf16, trunk32=True)), ("in0", dict(q=R.to_bf16, in0=True)),
                     ("both", dict(q=R.to_bf16, trunk32=True, in0=True)), ("f16", dict(q=to_f16))):
        hs = forward_scheme(secs, params, x, **kw)
        det = R.yolo_v3_detections(hs, 416, ratio=True)
        rel = [float(np.sqrt(((a[1] - b[1]) ** 2).mean()) / np.sqrt((b[1] ** 2).mean())) for a, b in zip(hs, h32)]
        miou, mds, cnt, lost = box_deviation(ref, det, 1e-2, thr=0.4)
        print("  %-8s head rel rms err %.4f %.4f %.4f | %4d candidates: min IoU %.4f  max |dscore| %.4f  lost %d" % (name, rel[0], rel[1], rel[2], cnt, miou, mds, lost))
