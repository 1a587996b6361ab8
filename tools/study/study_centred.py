"""CPU study (oracle only, no device): does MEAN-CENTRED 16-bit storage remove the cancellation that costs bf16 / fp16 storage its box
accuracy on weights with a trained file's batch-norm statistics (tools/study/study_precision.py)?

Scheme `cen`: every stored tensor is x' = round(x - m_c) with a per-channel offset m_c that never touches the 16-bit type:
  conv        z = sum W16 * x'  (zero padding of x', fp32 accumulate)  +  [ b + sum_{taps inside the image} W32 * m_in ]   (exact fold, per
              border case);  y = leaky(z);  stored y' = round(y - m_out)
  shortcut    out' = round(f' + x'),  m_out = m_f + m_x
  route / upsample / maxpool carry the offsets of their inputs; heads stay fp32 with m = 0.
In exact arithmetic this is the same network for ANY m; only the rounding differs.  Offsets: `calib` = per-channel means of the fp32
oracle on calibration images (disjoint from the evaluated ones), `analytic` = E[leaky(N(beta, gamma^2))] from the file's own batch-norm
parameters (no data), shortcut sums of those.
Prints per scheme: relative rms error of the three raw head tensors, min IoU / max |dscore| over the oracle's candidates, candidates lost."""
import glob, math, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
from oracle import yolo_ref as R
from yolo_tensorflow_amd import darknet_io as IO
from test_gpu_tuned import box_deviation

NIMG = int(os.environ.get("N", "3"))
STATS = os.environ.get("STATS", "log,benign").split(",")


def to_f16(x):
    return np.asarray(x, np.float32).astype(np.float16).astype(np.float32)


def leaky_mean(beta, std, slope=0.1):
    """E[leaky(z)], z ~ N(beta, std^2)."""
    beta = np.asarray(beta, np.float64); std = np.maximum(np.asarray(std, np.float64), 1e-30)
    t = beta / std
    Phi = 0.5 * (1.0 + np.vectorize(math.erf)(t / math.sqrt(2.0))); phi = np.exp(-0.5 * t * t) / math.sqrt(2 * math.pi)
    return beta * (Phi + slope * (1 - Phi)) + (1 - slope) * std * phi


def analytic_offsets(secs, params):
    layers = secs[1:]; m = []; ci = 0
    for i, s in enumerate(layers):
        t = s["type"]
        if t == "convolutional":
            p = params[ci]; ci += 1
            n = p["w_hwio"].shape[-1]
            is_head = i + 1 < len(layers) and layers[i + 1]["type"] in ("yolo", "region")
            if is_head or "bias" in p:
                m.append(np.zeros(n, np.float32))
            else:
                std = np.abs(p["gamma"])          # rolling variance == what the conv produces -> unit variance in front of gamma
                m.append(leaky_mean(p["beta"], std, 0.1 if s.get("activation") == "leaky" else 1.0).astype(np.float32))
        elif t == "shortcut":
            f = int(s["from"]); f = f if f >= 0 else i + f
            m.append(m[i - 1] + m[f])
        elif t == "route":
            ls = [int(v) for v in s["layers"].split(",")]; ls = [l if l >= 0 else i + l for l in ls]
            m.append(np.concatenate([m[l] for l in ls]))
        elif t == "upsample":
            m.append(m[i - 1])
        else:
            m.append(None)
    return m


def calib_offsets(secs, params, x):
    _, outs = R.forward(secs, params, x, collect=True)
    return [None if o is None else o.mean((0, 1, 2), dtype=np.float64).astype(np.float32) for o in outs]


def forward_centred(secs, params, x01, q, offs, first=0):
    """offs[i]: per-channel offset of layer i's stored output (zeros for heads); layers < `first` are stored uncentred."""
    layers = secs[1:]; outs = []; ms = []; heads = []; ci = 0
    x = q(np.asarray(x01, np.float32)); m = np.zeros(x.shape[-1], np.float32)
    for i, s in enumerate(layers):
        t = s["type"]
        if t == "convolutional":
            p = params[ci]; ci += 1
            is_head = i + 1 < len(layers) and layers[i + 1]["type"] in ("yolo", "region")
            w, b = R.fold_bn(p); st = int(s.get("stride", 1))
            z = R.conv2d_nhwc(x, q(w), st) + b
            if np.any(m != 0):
                M = np.broadcast_to(m, (1,) + x.shape[1:]).astype(np.float32)
                z = z + R.conv2d_nhwc(M, w, st)                       # exact fp32 fold, border cases included
            if s.get("activation", "logistic") == "leaky":
                z = R.leaky_relu(z)
            z = z.astype(np.float32)
            mo = np.zeros(z.shape[-1], np.float32) if (is_head or i < first) else offs[i]
            x = z if is_head else q(z - mo); m = mo
        elif t == "shortcut":
            f = int(s["from"]); f = f if f >= 0 else i + f
            x = q(outs[i - 1] + outs[f]); m = ms[i - 1] + ms[f]
        elif t == "route":
            ls = [int(v) for v in s["layers"].split(",")]; ls = [l if l >= 0 else i + l for l in ls]
            x = np.concatenate([outs[l] for l in ls], -1) if len(ls) > 1 else outs[ls[0]]
            m = np.concatenate([ms[l] for l in ls])
        elif t == "upsample":
            x = q(R.upsample_tf(x))
        elif t in ("yolo", "region"):
            heads.append((s, outs[i - 1])); outs.append(None); ms.append(None); continue
        else:
            raise ValueError(t)
        outs.append(x); ms.append(m)
    return heads


if __name__ == "__main__":
    txt = IO.cfg_text("yolov3"); secs = R.parse_cfg(txt)
    paths = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden", "images", "*.jpg")))
    from PIL import Image
    imgs = [np.asarray(Image.open(p).convert("RGB")) for p in paths]
    x_all = np.concatenate([R.input_process(im, 416) for im in imgs])
    noise = np.random.default_rng(5).random((2, 416, 416, 3), dtype=np.float32)
    for stats in STATS:
        flat = IO.synth_weights(IO.parse_cfg(txt), seed=3, stats=stats, obj_bias=-2.5 if stats == "log" else -0.75)
        params = R.unflatten_weights(flat, secs)
        if stats == "log":
            R.calibrate_bn_statistics(secs, params, np.concatenate([x_all, noise]), seed=3)
        x = x_all[:NIMG]
        h32 = R.forward(secs, params, x)[0]
        ref = R.yolo_v3_detections(h32, 416, ratio=True)
        oc = calib_offsets(secs, params, np.concatenate([x_all[NIMG:NIMG + 2], noise[:1]]))     # calibration images != evaluated images
        oa = analytic_offsets(secs, params)
        zero = [None if o is None else np.zeros_like(o) for o in oc]
        print("== %s weights, %d natural images" % (stats, NIMG), flush=True)
        for name, kw in (("bf16 dev", dict(q=R.to_bf16, offs=zero)), ("bf16 cen/calib", dict(q=R.to_bf16, offs=oc)),
                         ("bf16 cen/analytic", dict(q=R.to_bf16, offs=oa)), ("bf16 cen/calib 12+", dict(q=R.to_bf16, offs=oc, first=12)),
                         ("f16 dev", dict(q=to_f16, offs=zero)), ("f16 cen/calib", dict(q=to_f16, offs=oc)),
                         ("f16 cen/analytic", dict(q=to_f16, offs=oa)), ("f16 cen/calib 12+", dict(q=to_f16, offs=oc, first=12))):
            hs = forward_centred(secs, params, x, **kw)
            det = R.yolo_v3_detections(hs, 416, ratio=True)
            rel = [float(np.sqrt(((a[1] - b[1]) ** 2).mean()) / np.sqrt((b[1] ** 2).mean())) for a, b in zip(hs, h32)]
            miou, mds, cnt, lost = box_deviation(ref, det, 1e-2, thr=0.4)
            print("  %-20s head rel rms err %.4f %.4f %.4f | %4d candidates: min IoU %.4f  max |dscore| %.4f  lost %d"
                  % (name, rel[0], rel[1], rel[2], cnt, miou, mds, lost), flush=True)
