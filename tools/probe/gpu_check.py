"""Developer smoke run on a GPU box: a few operators + a small network against the oracle, then timing."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import yolo_ref as R
from yolo_tensorflow_amd import hip, darknet_io as IO

rng = np.random.default_rng(0)

def rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))

print("== conv ops ==")
for (n, h, cin, cout, k, s) in [(2, 16, 64, 128, 3, 1), (1, 13, 32, 64, 1, 1), (2, 20, 3, 32, 3, 1), (2, 16, 128, 255, 1, 1), (1, 26, 64, 128, 3, 2), (3, 13, 384, 128, 1, 1)]:
    x = R.to_bf16(rng.standard_normal((n, h, h, cin)).astype(np.float32))
    w = R.to_bf16((rng.standard_normal((k, k, cin, cout)) * np.sqrt(2.0 / (k * k * cin))).astype(np.float32))
    b = rng.standard_normal(cout).astype(np.float32)
    ref = R.leaky_relu(R.conv2d_nhwc(x, w, s) + b)
    for cfg in range(hip.op_conv_num_cfgs()):
        got = hip.op_conv2d(x, w, b, stride=s, act=1, tile_cfg=cfg)
        e = rel(got, R.to_bf16(ref))
        print("conv", (n, h, cin, cout, k, s), "cfg", cfg, "relerr %.2e" % e, "OK" if e < 2e-2 else "BAD")
    got = hip.op_conv2d(x, w, b, stride=s, act=1, dtype=hip.FP32)
    print("  fp32 relerr %.2e" % rel(got, ref))

print("== small network ==")
for name, size in (("yolov3", 96), ("yolov2", 96), ("yolov3-tiny", 96)):
    txt = IO.with_input_size(IO.cfg_text(name), size)
    secs = IO.parse_cfg(txt); flat = IO.synth_weights(secs, seed=0)
    img = rng.integers(0, 256, (2, size, size, 3), dtype=np.uint8)
    osecs = R.parse_cfg(txt); params = R.unflatten_weights(flat, osecs)
    x01 = img.astype(np.float32) / np.float32(255)
    for dtype, emu in ((hip.BF16, True), (hip.FP32, False)):
        eng = hip.Engine(txt, max_batch=2, dtype=dtype, keep_layers=True)
        eng.set_weights(flat)
        det = eng.forward(img)
        heads, outs = R.forward(osecs, params, R.to_bf16(x01) if emu else x01, emulate_bf16=emu, collect=True)
        worst = 0
        for i, o in enumerate(outs):
            if o is None: continue
            g = eng.layer_output(i, 2)
            e = rel(g, o); worst = max(worst, e)
            if e > (3e-2 if emu else 1e-3): print("  layer", i, osecs[i + 1]["type"], "relerr %.3e" % e)
        is_v3 = any(s["type"] == "yolo" for s in osecs)
        if is_v3:
            ref_det = R.yolo_v3_detections(heads, size, ratio=True)
        else:
            s, raw = heads[0]
            bx, ob, cl = R.region_decode(raw, R.yolo_anchors(s), int(s["classes"]))
            ref_det = None
        print(name, "dtype", dtype, "worst layer relerr %.3e" % worst, "det relerr", None if ref_det is None else "%.3e" % rel(det, ref_det))
        eng.close()

print("== timing yolov3 416 ==")
txt = IO.cfg_text("yolov3"); secs = IO.parse_cfg(txt); flat = IO.synth_weights(secs, seed=0)
B = int(os.environ.get("B", "32"))
eng = hip.Engine(txt, max_batch=B)
eng.set_weights(flat)
img = rng.integers(0, 256, (B, 416, 416, 3), dtype=np.uint8)
eng.forward(img, want_detections=False); eng.synchronize()
t, c = eng.time_forward(B, 5)
fl = eng.conv_flops() * B
print("heuristic cfgs: total %.3f ms conv %.3f ms  -> %.1f img/s, conv %.1f TFLOP/s" % (t, c, B / t * 1e3, fl / c / 1e9))
ms = eng.time_layers(B, 3)
t0 = time.time(); eng.autotune(B, 3); print("autotune %.1fs" % (time.time() - t0))
t, c = eng.time_forward(B, 10)
print("autotuned: total %.3f ms conv %.3f ms  -> %.1f img/s, conv %.1f TFLOP/s" % (t, c, B / t * 1e3, fl / c / 1e9))
ms = eng.time_layers(B, 5)
lay = secs[1:]
for i, m in enumerate(ms):
    if m > 0.02: print("  L%03d %-14s %.3f ms" % (i, lay[i]["type"] + ("%sx%s/%s f%s" % (lay[i].get("size"), lay[i].get("size"), lay[i].get("stride"), lay[i].get("filters")) if lay[i]["type"] == "convolutional" else ""), m))
dets = eng.detect(img[:4])
print("detect counts", [len(d) for d in dets])
