"""Parity on the reference's OWN test images and on weights with the batch-norm statistics of a trained file.

  * the six jpgs the reference ships for its detect scripts (V3/images/{dog,eagle,giraffe,horses,kite,person}.jpg, committed as
    fixtures under tests/golden/images/) go through the detector entry point at their native, non-square size: uint8 -> /255 -> legacy
    bilinear stretch to 416 x 416 ON THE DEVICE (D2T/YOLO_V3_convert_darkenet_to_Tensorflow.py:106-111, V3/YOLO_V3_inference.py:98-107)
    -> network -> decode -> threshold -> TF NMS, against the oracle's input_process + forward + detect;
  * the synthetic weights come in two flavours: the benign statistics used everywhere else, and `stats="log"` -- gamma up to 4.7 (some
    negative), beta to -11, rolling variance 8e-4 .. 19: the ranges of the reference's dump of real files (D2T/log.txt).

Stated tolerances (every oracle candidate above the detector's 0.4 threshold by more than the score bound):
  fp32 device path: IoU >= 0.999, |dscore| <= 1e-3   (north_star's tolerance)
  bf16:             IoU >= BF16_IOU, |dscore| <= 1e-2 (measured values are printed; DESIGN.md section 4 quotes them)
  fp8 (e4m3):       measured and printed, loose sanity bound only (the scheme has no reference counterpart)"""
import glob
import os

import numpy as np
import pytest
from oracle import yolo_ref as R
from yolo_tensorflow_amd import darknet_io as IO
from test_gpu_tuned import box_deviation

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.abspath(__file__))
IMAGES = sorted(glob.glob(os.path.join(ROOT, "golden", "images", "*.jpg")))
BF16_IOU = 0.985


def _load(path):
    from PIL import Image
    return np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB")))


_W = {}


def _weights(stats):
    """benign: the statistics used everywhere else.  log: gamma / beta drawn from the trained-file ranges (darknet_io.synth_weights
    stats="log"), then rolling mean / variance CALIBRATED on the six images + two noise images the way training's running averages
    would be (oracle.calibrate_bn_statistics; variances 8e-4 .. 20) -- the generator's analytic statistics only hold for white noise."""
    if stats not in _W:
        txt = IO.cfg_text("yolov3"); secs = IO.parse_cfg(txt)
        flat = IO.synth_weights(secs, seed=3, stats=stats, obj_bias=-2.5 if stats == "log" else -0.75)
        if stats == "log":
            osecs = R.parse_cfg(txt); params = R.unflatten_weights(flat, osecs)
            x = np.concatenate([R.input_process(_load(p), 416) for p in IMAGES] + [np.random.default_rng(5).random((2, 416, 416, 3), dtype=np.float32)])
            flat = R.flatten_weights(R.calibrate_bn_statistics(osecs, params, x, seed=3), osecs)
            bn = [q for q in params if "var" in q]
            assert min(q["var"].min() for q in bn) < 2e-3 and max(q["var"].max() for q in bn) > 15 and max(q["gamma"].max() for q in bn) > 4.5
        _W[stats] = (txt, flat)
    return _W[stats]


_REF = {}


def _oracle(stats):
    """fp32 oracle decoded tensors of the six images (input_process = /255 + legacy bilinear stretch), cached per weight flavour."""
    if stats not in _REF:
        txt, flat = _weights(stats)
        osecs = R.parse_cfg(txt); params = R.unflatten_weights(flat, osecs)
        out = []
        for p in IMAGES:
            heads, _ = R.forward(osecs, params, R.input_process(_load(p), 416))
            out.append(R.yolo_v3_detections(heads, 416, ratio=True)[0])
        _REF[stats] = np.stack(out)
    return _REF[stats]


def test_fixture_images_are_the_reference_set():
    names = [os.path.basename(p) for p in IMAGES]
    assert names == ["dog.jpg", "eagle.jpg", "giraffe.jpg", "horses.jpg", "kite.jpg", "person.jpg"]
    shapes = {os.path.basename(p): _load(p).shape for p in IMAGES}
    # 768x576 .. 1352x900; all but giraffe.jpg (500 x 500) are non-square
    assert shapes["dog.jpg"] == (576, 768, 3) and shapes["kite.jpg"] == (900, 1352, 3) and sum(s[0] != s[1] for s in shapes.values()) == 5


@pytest.mark.parametrize("stats", ["benign", "log"])
@pytest.mark.parametrize("dtype_name", ["fp32", "bf16", "fp8"])
def test_reference_images_through_the_detector(hiplib, stats, dtype_name):
    from yolo_tensorflow_amd import detector
    dtype = {"fp32": hiplib.FP32, "bf16": hiplib.BF16, "fp8": hiplib.FP8}[dtype_name]
    txt, flat = _weights(stats)
    ref = _oracle(stats)
    d = detector.YOLOV3(None, weights=flat, dtype=dtype)
    thr = d.threshold                                            # the converter's flag: 0.4
    det = np.stack([d.engine.forward_image(_load(p))[0] for p in IMAGES])
    margin = {"fp32": 1e-3, "bf16": 1e-2, "fp8": 0.0}[dtype_name]
    miou, mds, cnt, lost = box_deviation(ref, det, margin, thr=thr)
    print("natural images, %s weights, %s: %d candidates over %d images, min IoU %.4f, max |dscore| %.5f, below threshold %d"
          % (stats, dtype_name, cnt, len(IMAGES), miou, mds, lost))
    assert cnt >= 20
    if dtype_name == "fp32":
        assert lost == 0 and miou >= 0.999 and mds <= 1e-3
        # the entry point itself: (scores, boxes, classes) of detect_from_image against the oracle's tail on the oracle's tensor
        for k, p in enumerate(IMAGES):
            scores, boxes, classes = d.detect_from_image(_load(p))
            ob, os_, oc = R.detect_v3_tf(ref[k], thr, d.iou_threshold, d.max_output_size)
            assert len(scores) == len(os_) and np.array_equal(classes, oc)
            np.testing.assert_allclose(scores, os_, rtol=0, atol=1e-3)
            np.testing.assert_allclose(boxes, ob, rtol=0, atol=2e-3)
    elif dtype_name == "bf16" and stats == "benign":
        assert lost == 0 and miou >= BF16_IOU and mds <= 1e-2
    elif dtype_name == "bf16":
        # trained-file statistics: large per-channel offsets (beta to -11, rolling means to +-20) make every conv a difference of large
        # numbers, and bf16's 8-bit significand then costs far more than on benign statistics -- the oracle's own bf16-storage emulation
        # shows the same loss (tools/study_precision.py: min IoU 0.76 / |dscore| 0.16; fp16 storage: 0.97 / 0.015).  What is asserted is
        # that the device IS that emulation (same roundings, different fp32 summation order), and a floor on the fp32 deviation.
        osecs = R.parse_cfg(txt); params = R.unflatten_weights(flat, osecs)
        emu = []
        for p in IMAGES[:2]:
            heads, _ = R.forward(osecs, params, R.to_bf16(R.input_process(_load(p), 416)), emulate_bf16=True)
            emu.append(R.yolo_v3_detections(heads, 416, ratio=True)[0])
        e = box_deviation(np.stack(emu), det[:2], 1e-2, thr=thr)
        print("   ... device vs the oracle's bf16-storage emulation (2 images): %d candidates, min IoU %.4f, max |dscore| %.5f, lost %d" % (e[2], e[0], e[1], e[3]))
        assert e[0] >= 0.85 and e[1] <= 0.1            # (measured 0.904 / 0.073: the same amplification acts on the summation-order differences)
        assert miou >= 0.5 and mds <= 0.3
    else:
        assert np.isfinite(det).all() and (stats == "log" or (miou >= 0.5 and mds <= 0.2))
    d.engine.close()


def test_log_statistics_32_images_bf16_and_fp8(hiplib):
    """DESIGN section 4's 32-image table repeated on weights with trained-file batch-norm statistics (416 x 416, batch 32, the
    committed tile plans)."""
    import json
    txt, flat = _weights("log")
    img = np.random.default_rng(1).integers(0, 256, (32, 416, 416, 3), dtype=np.uint8)
    osecs = R.parse_cfg(txt); params = R.unflatten_weights(flat, osecs)
    ref = []
    for b in range(32):
        heads, _ = R.forward(osecs, params, img[b:b + 1].astype(np.float32) / np.float32(255))
        ref.append(R.yolo_v3_detections(heads, 416, ratio=True)[0])
    ref = np.stack(ref)
    plans = os.path.join(os.path.dirname(ROOT), "yolo_tensorflow_amd", "tuned")
    eng = hiplib.Engine(txt, max_batch=32)
    eng.set_weights(flat); eng.set_tile_configs(json.load(open(os.path.join(plans, "yolov3_416_b32_bf16.json")))["cfgs"])
    det = eng.forward(img); eng.close()
    miou, mds, cnt, lost = box_deviation(ref, det, 1e-2)
    print("log-statistics weights, bf16 416 b32 vs fp32 oracle: %d candidates, min IoU %.4f, max |dscore| %.5f, lost %d" % (cnt, miou, mds, lost))
    assert cnt > 100 and miou >= 0.5 and mds <= 0.3          # (see the note in the test above: what bf16 storage costs on these statistics)
    # the mixed e4m3 / bf16 plan and fp16 storage on the same weights (the plan can be no better than the bf16 it falls
    # back to, and tools/study_bits.py says why: on this network 1 - IoU is set by the significand width alone, 13 bits for 0.99)
    mp = json.load(open(os.path.join(plans, "yolov3_416_b32_mixed.json")))
    eng = hiplib.Engine(IO.with_layer_store(txt, mp["store_bf16"]), max_batch=32, dtype=hiplib.FP8)
    eng.set_weights(flat); dm = eng.forward(img); eng.close()
    mm = box_deviation(ref, dm, 0.0)
    print("log-statistics weights, mixed e4m3 / bf16 plan 416 b32 vs fp32 oracle: %d candidates, min IoU %.4f, max |dscore| %.4f, below threshold %d" % (mm[2], mm[0], mm[1], mm[3]))
    eng = hiplib.Engine(txt, max_batch=32, dtype=hiplib.FP16)
    eng.set_weights(flat); dh = eng.forward(img); eng.close()
    mh = box_deviation(ref, dh, 1e-2)
    print("log-statistics weights, fp16 416 b32 vs fp32 oracle: %d candidates, min IoU %.4f, max |dscore| %.4f, lost %d" % (mh[2], mh[0], mh[1], mh[3]))
    assert np.isfinite(dm).all() and mh[0] >= 0.9 and mh[0] > miou        # three more significand bits buy what the bits law says: ~0.96
    _, outs32 = R.forward(osecs, params, img[:1].astype(np.float32) / np.float32(255), collect=True)
    for scales, name in ((None, "unit scales"), (R.fp8_calibrate_scales(osecs, outs32), "calibrated scales")):
        eng = hiplib.Engine(txt, max_batch=32, dtype=hiplib.FP8)
        if scales is not None:
            eng.set_act_scales(scales)
        eng.set_weights(flat); eng.set_tile_configs(json.load(open(os.path.join(plans, "yolov3_416_b32_fp8.json")))["cfgs"])
        d8 = eng.forward(img); eng.close()
        m8 = box_deviation(ref, d8, 0.0)
        print("log-statistics weights, fp8 416 b32, %s, vs fp32 oracle: %d candidates, min IoU %.4f, max |dscore| %.4f, below threshold %d" % (name, m8[2], m8[0], m8[1], m8[3]))
        assert np.isfinite(d8).all()
