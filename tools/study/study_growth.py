"""CPU study (oracle only): per-layer error (rms over elements of the per-channel-normalised difference to the fp32 oracle) of the
YOLOv3 stack when only the activations, only the filters, or both are rounded to bf16 (centred / plain), and the fate of a 1e-3
perturbation INJECTED at cfg layer 1 into an otherwise exact network.  Usage: study_growth.py log,benign.
Result (profiles/r04_precision_study.txt): the network with trained-file statistics amplifies any perturbation ~10x between layer 1 and
the heads (the benign one damps it 10-100x), so its sensitivity to storage rounding is a property of that random network, not of offsets."""
import glob, os, sys
import numpy as np
_H = os.path.dirname(os.path.abspath(__file__)); sys.path[:0] = [os.path.join(_H, "..", ".."), os.path.join(_H, "..", "..", "tests"), _H]
from oracle import yolo_ref as R
from yolo_tensorflow_amd import darknet_io as IO
import study_centred as S
from PIL import Image
txt = IO.cfg_text("yolov3"); secs = R.parse_cfg(txt)
paths = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden", "images", "*.jpg")))
imgs = [np.asarray(Image.open(p).convert("RGB")) for p in paths]
x_all = np.concatenate([R.input_process(im, 416) for im in imgs])
noise = np.random.default_rng(5).random((2, 416, 416, 3), dtype=np.float32)

def fwd(secs, params, x01, qa, qw, offs, inject=None):
    """general: qa = activation rounding, qw = weight rounding; inject=(layer, rel): add gaussian noise of rel*centred-rms at that layer only"""
    layers = secs[1:]; outs = []; ms = []; heads = []; ci = 0
    x = qa(np.asarray(x01, np.float32)); m = np.zeros(x.shape[-1], np.float32)
    rng = np.random.default_rng(0)
    for i, s in enumerate(layers):
        t = s["type"]
        if t == "convolutional":
            p = params[ci]; ci += 1
            is_head = i + 1 < len(layers) and layers[i + 1]["type"] in ("yolo", "region")
            w, b = R.fold_bn(p); st = int(s.get("stride", 1))
            z = R.conv2d_nhwc(x, qw(w), st) + b
            if np.any(m != 0):
                M = np.broadcast_to(m, (1,) + x.shape[1:]).astype(np.float32)
                z = z + R.conv2d_nhwc(M, w, st)
            if s.get("activation", "logistic") == "leaky":
                z = R.leaky_relu(z)
            z = z.astype(np.float32)
            if inject and inject[0] == i:
                sd = z.std((0, 1, 2), keepdims=True)
                z = z + (rng.standard_normal(z.shape).astype(np.float32) * sd * inject[1])
            mo = np.zeros(z.shape[-1], np.float32) if (is_head or offs is None) else offs[i]
            x = z if is_head else qa(z - mo); m = mo
        elif t == "shortcut":
            f = int(s["from"]); f = f if f >= 0 else i + f
            x = qa(outs[i - 1] + outs[f]); m = ms[i - 1] + ms[f]
        elif t == "route":
            ls = [int(v) for v in s["layers"].split(",")]; ls = [l if l >= 0 else i + l for l in ls]
            x = np.concatenate([outs[l] for l in ls], -1) if len(ls) > 1 else outs[ls[0]]
            m = np.concatenate([ms[l] for l in ls])
        elif t == "upsample":
            x = qa(R.upsample_tf(x))
        elif t in ("yolo", "region"):
            heads.append((s, outs[i - 1])); outs.append(None); ms.append(None); continue
        outs.append(x); ms.append(m)
    return heads, outs, ms

ident = lambda a: a
for stats in sys.argv[1].split(","):
    flat = IO.synth_weights(IO.parse_cfg(txt), seed=3, stats=stats, obj_bias=-2.5 if stats in ("log", "real") else -0.75)
    params = R.unflatten_weights(flat, secs)
    if stats in ("log", "real"):
        R.calibrate_bn_statistics(secs, params, np.concatenate([x_all, noise]), seed=3, keep_var=stats == "real")
    x = x_all[:1]
    oc = S.calib_offsets(secs, params, np.concatenate([x_all[3:5], noise[:1]]))
    h0, o0, m0 = fwd(secs, params, x, ident, ident, None)
    def relerr(outs, ms):
        r = []
        for i, (a, b) in enumerate(zip(outs, o0)):
            if b is None: r.append(None); continue
            aa = a + (ms[i] if ms[i] is not None else 0)
            r.append(float(np.sqrt(((aa - b) ** 2).mean()) / b.std((0,1,2)).mean()))   # rough
        return r
    def relerr2(outs, ms):
        r = []
        for i, (a, b) in enumerate(zip(outs, o0)):
            if b is None: r.append(None); continue
            aa = a + (ms[i] if ms[i] is not None else 0)
            sd = b.std((0, 1, 2)) + 1e-12
            r.append(float(np.sqrt((((aa - b) / sd) ** 2).mean())))    # per-channel-normalised rms error
        return r
    print("==", stats)
    rows = {}
    for name, kw in (("act bf16 only", dict(qa=R.to_bf16, qw=ident, offs=oc)), ("w bf16 only", dict(qa=ident, qw=R.to_bf16, offs=oc)),
                     ("both cen", dict(qa=R.to_bf16, qw=R.to_bf16, offs=oc)), ("both dev", dict(qa=R.to_bf16, qw=R.to_bf16, offs=None)),
                     ("inject L1 1e-3", dict(qa=ident, qw=ident, offs=None, inject=(1, 1e-3))),
                     ("inject L40 1e-3", dict(qa=ident, qw=ident, offs=None, inject=(40, 1e-3)))):
        h, o, ms = fwd(secs, params, x, **kw)
        rows[name] = relerr2(o, ms)
    idx = [0, 1, 2, 3, 4, 5, 8, 11, 12, 20, 28, 36, 37, 45, 53, 61, 62, 70, 74, 79, 80, 84, 91, 92, 96, 103, 104]
    print("layer " + " ".join("%6d" % i for i in idx))
    for k, v in rows.items():
        print("%-16s" % k + " ".join("%6.4f" % v[i] if v[i] is not None else "   -  " for i in idx))
