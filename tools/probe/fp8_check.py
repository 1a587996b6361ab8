"""Developer check of the fp8 configuration on the GPU: conv operator and a small network against the oracle's emulation."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from yolo_tensorflow_amd import hip, darknet_io as IO
from oracle import yolo_ref as R

rng = np.random.default_rng(0)


def conv_ref(x, w, b, stride, act, residual):
    xq = R.to_fp8_e4m3(x)
    amax = np.abs(w).max(axis=(0, 1, 2)); osc = np.where(amax > 0, amax / np.float32(448), np.float32(1)).astype(np.float32)
    wq = R.to_fp8_e4m3(w / osc[None, None, None, :])
    acc = R.conv2d_nhwc(xq, wq, stride)
    v = (acc.astype(np.float64) * osc + b.astype(np.float64)).astype(np.float32)
    if act:
        v = R.leaky_relu(v)
    v = R.to_bf16(v)
    q = R.to_fp8_e4m3(v)
    if residual is not None:
        q = R.to_fp8_e4m3(q + R.to_fp8_e4m3(residual))
    return q


for (n, h, cin, cout, k, st, res) in [(2, 13, 64, 128, 3, 1, False), (2, 16, 128, 64, 1, 1, True), (1, 26, 32, 64, 3, 2, False), (2, 13, 256, 512, 3, 1, True), (1, 13, 48, 255, 1, 1, False)]:
    x = rng.normal(0, 1, (n, h, h, cin)).astype(np.float32)
    w = (rng.normal(0, 1, (k, k, cin, cout)) / np.sqrt(k * k * cin)).astype(np.float32)
    b = rng.normal(0, 0.1, cout).astype(np.float32)
    ho = (h + 2 * (k // 2) - k) // st + 1
    r = rng.normal(0, 1, (n, ho, ho, cout)).astype(np.float32) if res else None
    want = conv_ref(x, w, b, st, 1, r)
    for cfg in (-1, 0, 16, 17, 23, 32):
        got = hip.op_conv2d(x, w, b, stride=st, act=1, residual=r, dtype=hip.FP8, tile_cfg=cfg)
        same = float(np.mean(got == want)); md = float(np.abs(got - want).max())
        print("conv n%d h%d cin%d cout%d k%d s%d res%d cfg %3d: identical %.5f max|diff| %.4f" % (n, h, cin, cout, k, st, res, cfg, same, md))

if len(sys.argv) > 1 and sys.argv[1] == "net":
    cfg, size, sem = "yolov3", 96, "tf"
    txt = IO.with_input_size(IO.cfg_text(cfg), size)
    secs = IO.parse_cfg(txt); flat = IO.synth_weights(secs, seed=0)
    img = np.random.default_rng(2).integers(0, 256, (2, size, size, 3), dtype=np.uint8)
    x01 = img.astype(np.float32) / np.float32(255)
    osecs = R.parse_cfg(txt); params = R.unflatten_weights(flat, osecs)
    _, outs32 = R.forward(osecs, params, x01, semantics=sem, collect=True)
    sc = R.fp8_calibrate_scales(osecs, outs32)
    eng = hip.Engine(txt, max_batch=2, dtype=hip.FP8, keep_layers=True)
    eng.set_act_scales(sc); eng.set_weights(flat)
    eng.forward(img)
    dev = []
    for i, s in enumerate(osecs[1:]):
        multi = s["type"] == "route" and "," in s["layers"]
        dev.append(None if s["type"] in ("yolo", "region") or multi else eng.layer_output(i, 2))
    heads, outs = R.fp8_scheme_forward(osecs, params, x01, scales=sc, semantics=sem, teacher=dev)
    for i, s in enumerate(osecs[1:]):
        if dev[i] is None: continue
        d = np.abs(dev[i] - outs[i]); m = np.maximum(np.abs(dev[i]), np.abs(outs[i]))
        bad = d > 0.126 * m
        print(i, s["type"], "scale", sc[i], "same %.5f" % np.mean(dev[i] == outs[i]), "beyond-1-step", int(bad.sum()),
              "max|x|/scale %.1f" % (np.abs(outs[i]).max() / sc[i]), "ex", (dev[i][bad][:3], outs[i][bad][:3]) if bad.any() else "")
        if i > 12: break
    # exact float64 evaluation of layer 1 from the device's layer-0 output
    i = 1
    p = params[1]; w, b = R.fold_bn(p)
    sx = np.full(w.shape[2], sc[0], np.float32)
    weff = (w * sx[None, None, :, None]).astype(np.float32)
    amax = np.abs(weff).max(axis=(0, 1, 2)); osc = (amax / np.float32(448)).astype(np.float32)
    wq = R.to_fp8_e4m3(weff / osc[None, None, None, :])
    cx = R.to_fp8_e4m3(dev[0] / sx)
    acc64 = R.conv2d_nhwc(cx.astype(np.float64), wq.astype(np.float64), 2) if False else None
    # im2col in float64 by hand for a few bad elements
    d = np.abs(dev[1] - outs[1]); m = np.maximum(np.abs(dev[1]), np.abs(outs[1]))
    idx = np.argwhere(d > 1.5 * np.maximum(0.125 * m, sc[1] * 2.0 ** -9))
    print("two-step mismatches at layer 1:", len(idx))
    xp = np.pad(cx.astype(np.float64), ((0, 0), (1, 1), (1, 1), (0, 0)))
    for (n_, oy, ox, co) in idx[:6]:
        patch = xp[n_, oy * 2:oy * 2 + 3, ox * 2:ox * 2 + 3, :]
        acc = float((patch * wq[..., co].astype(np.float64)).sum())
        v = acc * float(osc[co]) + float(b[co])
        v = v if v > 0 else 0.1 * v
        print("  elem", (n_, oy, ox, co), "exact v/scale %.6f (in 2^-9 units %.3f)" % (v / sc[1], v / sc[1] * 512), "device %.1f oracle %.1f" % (dev[1][n_, oy, ox, co] / sc[1] * 512, outs[1][n_, oy, ox, co] / sc[1] * 512), "acc %.3f bias %.5f osc %.3e" % (acc, b[co], osc[co]))
