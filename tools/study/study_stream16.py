"""CPU study (oracle only): cheaper-than-3-products plans for the tolerance-meeting 16-bit-MFMA configuration.

Emulation as in study_mixed16.py, generalised: every stored tensor has a significand width (11 = plain fp16, 22 = split-fp16 pair) and every
conv's folded filters have one of their own (11 / 22), so that the MFMA products per algorithmic product of a conv are
    x22 * w22 -> 3   (W_hi x_hi + W_hi x_lo + W_lo x_hi)
    x22 * w11 -> 2   (W_hi x_hi + W_hi x_lo)            "operand-asymmetric", activations wide
    x11 * w22 -> 2   (W_hi x_hi + W_lo x_hi)            "operand-asymmetric", filters wide
    x11 * w11 -> 1
Plans:
  * stream: the RESIDUAL STREAM only as pairs -- the image, the stem, the stride-2 convs, every 3x3 conv whose output enters a shortcut, the
    shortcut sums; the block-internal 1x1 outputs plain.  (The 1x1 conv then reads a pair: 3 products on 10 % of a block's FLOPs; the 3x3 reads a
    plain tensor: 1 product.)  Variants for the FPN / detection blocks, which have no shortcuts: all plain, all pairs, or 1x1-out plain / 3x3-out pair.
  * asym: activations 22 x filters 11 everywhere, and the reverse.
  * combinations with a full-pairs prefix (layers 0..11, what study_mixed16 found to matter on the real vectors).
Prints per plan: MFMA products per algorithmic product (FLOP-weighted) and min IoU / max |dscore| / lost per image against the fp32 oracle.
Usage: study_stream16.py real|log|benign [image indices, default all six]      -> profiles/r06_stream16_study.txt"""
import glob, os, sys
import numpy as np
_H = os.path.dirname(os.path.abspath(__file__)); sys.path[:0] = [os.path.join(_H, "..", ".."), os.path.join(_H, "..", "..", "tests"), _H]
from oracle import yolo_ref as R
from yolo_tensorflow_amd import darknet_io as IO
from test_gpu_tuned import box_deviation
from PIL import Image


def qbits(x, bits):
    if bits >= 24:
        return np.asarray(x, np.float32)
    if bits == 11:
        return R.to_f16(x)
    x = np.asarray(x, np.float32); m, e = np.frexp(x)
    return np.ldexp(np.round(m * (1 << bits)) / (1 << bits), e).astype(np.float32)


def forward_plan(secs, params, x01, abits, wbits):
    """abits[i]: significand bits of layer i's stored tensor (-1 = the image); wbits[i]: of conv i's folded filters."""
    layers = secs[1:]; outs = []; heads = []; ci = 0
    x = qbits(np.asarray(x01, np.float32), abits[-1])
    for i, s in enumerate(layers):
        t = s["type"]
        if t == "convolutional":
            p = params[ci]; ci += 1
            is_head = i + 1 < len(layers) and layers[i + 1]["type"] in ("yolo", "region")
            w, b = R.fold_bn(p); st = int(s.get("stride", 1))
            z = R.conv2d_nhwc(x, qbits(w, wbits[i]), st) + b
            if s.get("activation", "logistic") == "leaky":
                z = R.leaky_relu(z)
            z = z.astype(np.float32)
            x = z if is_head else qbits(z, abits[i])
        elif t == "shortcut":
            f = int(s["from"]); f = f if f >= 0 else i + f
            x = qbits(outs[i - 1] + outs[f], abits[i])
        elif t == "route":
            ls = [int(v) for v in s["layers"].split(",")]; ls = [l if l >= 0 else i + l for l in ls]
            x = np.concatenate([outs[l] for l in ls], -1) if len(ls) > 1 else outs[ls[0]]
        elif t == "upsample":
            x = qbits(R.upsample_tf(x), abits[i])
        elif t in ("yolo", "region"):
            heads.append((s, outs[i - 1])); outs.append(None); continue
        outs.append(x)
    return heads


def main():
    stats = sys.argv[1] if len(sys.argv) > 1 else "real"
    txt = IO.cfg_text("yolov3"); secs = R.parse_cfg(txt); isecs = IO.parse_cfg(txt)
    paths = sorted(glob.glob(os.path.join(_H, "..", "..", "tests", "golden", "images", "*.jpg")))
    sel = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else list(range(len(paths)))
    imgs = [np.asarray(Image.open(p).convert("RGB")) for p in paths]
    x_all = np.concatenate([R.input_process(im, 416) for im in imgs])
    noise = np.random.default_rng(5).random((2, 416, 416, 3), dtype=np.float32)
    flat = IO.synth_weights(isecs, seed=3, stats=stats, obj_bias=-2.5 if stats in ("log", "real") else -0.75)
    params = R.unflatten_weights(flat, secs)
    if stats in ("log", "real"):
        R.calibrate_bn_statistics(secs, params, np.concatenate([x_all, noise]), seed=3, keep_var=stats == "real")
    layers = secs[1:]; NL = len(layers)
    shapes = IO.layer_shapes(isecs)
    convs = [i for i, s in enumerate(layers) if s["type"] == "convolutional"]
    flops = {i: 2.0 * int(layers[i]["size"]) ** 2 * shapes[i][4] * int(layers[i]["filters"]) * shapes[i][1] * shapes[i][2] for i in convs}
    tot = sum(flops.values())

    def inputs(i):
        s = layers[i]; t = s["type"]
        if t == "shortcut":
            f = int(s["from"]); return [i - 1, f if f >= 0 else i + f]
        if t == "route":
            return [int(v) if int(v) >= 0 else i + int(v) for v in s["layers"].split(",")]
        return [i - 1]

    # the block-internal 1x1 convs of the backbone: a 1x1 conv whose only reader is the next 3x3 conv whose output enters a shortcut
    inner = set(i for i in convs if int(layers[i]["size"]) == 1 and i + 2 < NL and layers[i + 1]["type"] == "convolutional"
                and int(layers[i + 1]["size"]) == 3 and layers[i + 2]["type"] == "shortcut")
    first_fpn = 75            # cfg layers 75.. : the detection blocks (1x1 / 3x3 alternating, no shortcuts), heads, routes, upsamples
    fpn_1x1 = set(i for i in convs if i >= first_fpn and int(layers[i]["size"]) == 1)
    fpn_3x3 = set(i for i in convs if i >= first_fpn and int(layers[i]["size"]) == 3)

    def close(ab):
        """movers inherit; both operands of a shortcut / all inputs of a route share the WIDER form (an 11-bit producer feeding a 22-bit
        concatenation is promoted, as darknet_io.pair_closure does)."""
        changed = True
        while changed:
            changed = False
            for i, s in enumerate(layers):
                t = s["type"]
                if t in ("shortcut", "route"):
                    want = max([ab[j] for j in inputs(i)] + ([ab[i]] if t == "shortcut" else []))
                    for j in inputs(i) + [i]:
                        k = j
                        while k >= 0 and layers[k]["type"] in ("route", "upsample", "maxpool") and len(inputs(k)) == 1:
                            k = inputs(k)[0]
                        for m in (j, k):
                            if m >= -1 and ab.get(m, 0) != want and (m < 0 or layers[m]["type"] not in ("yolo",)):
                                ab[m] = want; changed = True
                elif t in ("upsample", "maxpool"):
                    if ab[i] != ab[i - 1]:
                        ab[i] = ab[i - 1]; changed = True
        return ab

    def plan(act22, w_rule):
        """act22: set of tensor indices stored as pairs; w_rule(i, xbits) -> filter bits of conv i."""
        ab = {i: (22 if i in act22 else 11) for i in range(-1, NL)}
        ab = close(ab)
        wb = {i: w_rule(i, ab[i - 1]) for i in convs}
        return ab, wb

    everything = set(range(-1, NL))
    follow = lambda i, xb: xb                     # filters as wide as the input (22 x 22 = 3 products, 11 x 11 = 1)
    plans = [
        ("all plain fp16 (x11 w11)", plan(set(), follow)),
        ("all pairs (x22 w22)", plan(everything, follow)),
        ("asym: x22 w11 everywhere", plan(everything, lambda i, xb: 11)),
        ("asym: x11 w22 everywhere", plan(set(), lambda i, xb: 22)),
        ("stream pairs, inner 1x1 plain; FPN plain", plan(everything - inner - set(range(first_fpn, NL)), follow)),
        ("stream pairs, inner 1x1 plain; FPN pairs", plan(everything - inner, follow)),
        ("stream pairs, inner 1x1 plain; FPN 1x1-out plain", plan(everything - inner - fpn_1x1, follow)),
        ("stream pairs, inner 1x1 plain; FPN 3x3-out plain", plan(everything - inner - fpn_3x3, follow)),
        ("stream pairs + FPN 1x1-out plain, 1x1 filters 11 bits (x22 w11 on the 1x1)", plan(everything - inner - fpn_1x1, lambda i, xb: 11 if int(layers[i]["size"]) == 1 else xb)),
        ("stream pairs + FPN 1x1-out plain, 3x3 filters 22 bits (x11 w22 on the 3x3)", plan(everything - inner - fpn_1x1, lambda i, xb: 22)),
        ("pairs 0..11 full, then stream pairs + FPN 1x1-out plain", plan(everything - set(i for i in inner if i > 11) - fpn_1x1, follow)),
        ("pairs 0..11 full, then x22 w11", plan(everything, lambda i, xb: 22 if i <= 11 else 11)),
    ]
    x = x_all[sel]
    ref = R.yolo_v3_detections(R.forward(secs, params, x)[0], 416, ratio=True)
    print("== study_stream16 stats=%s images=%s" % (stats, [os.path.basename(paths[k]) for k in sel]), flush=True)
    for name, (ab, wb) in plans:
        prod = {i: {(22, 22): 3, (22, 11): 2, (11, 22): 2, (11, 11): 1}[(ab[i - 1], wb[i])] for i in convs}
        cost = sum(flops[i] * prod[i] for i in convs) / tot
        det = R.yolo_v3_detections(forward_plan(secs, params, x, ab, wb), 416, ratio=True)
        per = [box_deviation(ref[k:k + 1], det[k:k + 1], 1e-3, thr=0.4) for k in range(len(sel))]
        print("%-84s products %.2f | min over images %.5f | " % (name, cost, min(m[0] for m in per))
              + " | ".join("%.5f %.5f %d" % (m[0], m[1], m[3]) for m in per), flush=True)


if __name__ == "__main__":
    main()
