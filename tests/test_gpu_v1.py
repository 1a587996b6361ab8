"""YOLOv1 (BASELINE config 1) on the device: the 7x7/2 first conv (space-to-depth form), bias convs,
SAME pools, [connected] layers (CHW flatten), [detection] decode, the reference's NMS call with its swapped width/height, and the
`Yolo(weights_file)` entry point -- against golden vectors of the compiled reference (mini topology, every layer) and against the
oracle's restatement of V1/YOLO_V1_Inference.py:124-270 (full 448x448 network)."""
import os

import numpy as np
import pytest
from conftest import golden
from oracle import yolo_ref as R
from yolo_tensorflow_amd import darknet_io as IO

pytestmark = pytest.mark.gpu


def _relmax(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


def test_mini_v1_matches_compiled_reference_fp32(hiplib):
    """Every layer of the reference's own C forward pass (tests/golden/mini_v1.npz, tools/make_golden.py): 7x7/2 conv, pools,
    3x3/2 conv, [connected] x2, and the decoded [detection] rows vs get_detection_detections (DN/detection_layer.c:225-254)."""
    g = golden("mini_v1.npz")
    cfg = str(g["cfg"])
    x = (g["image_u8"].astype(np.float32) / np.float32(255)) * np.float32(2) - np.float32(1)
    for dtype, tol in ((hiplib.FP32, 2e-4), (hiplib.BF16, 3e-2)):
        eng = hiplib.Engine(cfg, max_batch=1, dtype=dtype, semantics=hiplib.SEM_DARKNET, keep_layers=True)
        eng.set_weights(g["weights"])
        det = eng.forward(np.ascontiguousarray(x[None]), scale=1.0)[0]
        secs = IO.parse_cfg(cfg)
        for i, s in enumerate(secs[1:]):
            if s["type"] == "detection":
                continue
            ref = g["layer_%02d" % i]
            got = eng.layer_output(i, 1)
            assert _relmax(got.reshape(-1), ref.reshape(-1)) < tol, "layer %d (%s) dtype %d" % (i, s["type"], dtype)
        assert det.shape == (18, 25)
        bb = g["boxes_raw"] / np.float32(64)                      # the reference scales by the image size handed to get_network_boxes
        np.testing.assert_allclose(det[:, :4], bb, rtol=tol * 10, atol=tol)
        np.testing.assert_allclose(det[:, 4], g["obj_raw"], rtol=tol * 10, atol=tol)
        if dtype == hiplib.FP32:
            pr = det[:, 4:5] * det[:, 5:]
            clear = np.abs(pr - float(g["thresh"])) > 1e-3
            assert np.array_equal((pr > float(g["thresh"]))[clear], (g["prob_raw"] > 0)[clear])
            np.testing.assert_allclose(np.where(pr > float(g["thresh"]), pr, 0)[clear], g["prob_raw"][clear], rtol=2e-3, atol=2e-4)
        eng.close()


@pytest.fixture(scope="module")
def v1_full():
    txt = IO.cfg_text("yolov1")
    secs = IO.parse_cfg(txt)
    flat = IO.synth_weights(secs, seed=4)
    img = np.random.default_rng(21).integers(0, 256, (448, 448, 3), dtype=np.uint8)
    osecs = R.parse_cfg(txt); params = R.unflatten_weights(flat, osecs)
    x = (img.astype(np.float32) / np.float32(255)) * np.float32(2) - np.float32(1)
    heads, _ = R.forward(osecs, params, x[None])
    pred = heads[0][1].reshape(-1)
    return txt, flat, img, pred


def test_config1_yolov1_448_network_vs_oracle(hiplib, v1_full):
    """BASELINE config 1's network (YOLOv1 448x448, batch 1) on the GPU: the 1470 predictions (through the decoded rows) vs the fp32
    oracle; fp32 path 2e-3 of the tensor's scale, bf16 path 5e-2 (27 layers of bf16 storage, K up to 50176)."""
    txt, flat, img, pred = v1_full
    want = R.v1_rows(pred, 7, 2, 20)
    for dtype, tol in ((hiplib.FP32, 2e-3), (hiplib.BF16, 5e-2)):
        eng = hiplib.Engine(txt, max_batch=1, dtype=dtype)
        eng.set_weights(flat)
        det = eng.forward(img[None])[0]                           # uint8 in: /255, then the cfg's (x * 2 - 1)
        assert det.shape == (98, 25) and np.isfinite(det).all()
        assert _relmax(det, want) < tol, "dtype %d: %g" % (dtype, _relmax(det, want))
        # selection + NMS: the device's tail on its own decoded tensor == the oracle's `_build_detector` tail on the same numbers
        res = eng.postprocess(1, score_thr=0.2, iou_thr=0.4, max_out=10, nms_mode=hiplib.NMS_TF_V1, select_mode=hiplib.SELECT_GE)[0]
        f = np.float32
        sc = (det[:, 4:5] * det[:, 5:]); smax = sc.max(-1); lab = sc.argmax(-1)
        m = smax >= f(0.2)
        b = det[m, :4]
        _b = np.stack([b[:, 1] - f(0.5) * b[:, 2], b[:, 0] - f(0.5) * b[:, 3], b[:, 1] + f(0.5) * b[:, 2], b[:, 0] + f(0.5) * b[:, 3]], 1)
        sel = R.tf_nms(_b, smax[m], 10, 0.4)
        assert len(res) == len(sel) > 0
        assert np.array_equal(res["score"], smax[m][sel]) and np.array_equal(res["cls"], lab[m][sel])
        assert np.array_equal(res["x0"], _b[sel, 1]) and np.array_equal(res["y0"], _b[sel, 0])
        eng.close()


def test_yolo_v1_entry_point(hiplib, v1_full, tmp_path):
    """`Yolo(weights_file)` / `detect_from_file` (V1/YOLO_V1_Inference.py:32, :294): a Darknet .weights file and an image file in,
    (class, x, y, w, h, score) tuples out, equal to the oracle's detector on the oracle's own fp32 predictions up to the bf16
    tolerance of the scores; boxes.txt written in the reference's format."""
    from PIL import Image
    from yolo_tensorflow_amd.yolo_v1 import Yolo
    txt, flat, img, pred = v1_full
    wf = str(tmp_path / "yolov1.weights"); IO.write_weights_file(wf, flat, 0, 1)
    png = str(tmp_path / "img.png")
    Image.fromarray(img[:, :, ::-1]).save(png)                    # the entry point hands PIL's RGB over as BGR, like cv2.imread
    y = Yolo(wf, verbose=False, dtype=hiplib.FP32)
    out = y.detect_from_file(png, imshow=False, deteted_boxes_file=str(tmp_path / "boxes.txt"))
    y.close()
    wb, ws, wc = R.detect_v1_tf(pred, 7, 2, 20, 0.2, 0.4, 10)
    assert len(out) == len(ws) > 0
    for (name, bx, by, bw, bh, s), b, sc, c in zip(out, wb, ws, wc):
        assert name == IO.v1_classes()[int(c)]
        assert abs(s - sc) < 2e-3
        np.testing.assert_allclose([bx, by, bw, bh], b * np.float32(448), rtol=2e-3, atol=0.5)
    lines = open(str(tmp_path / "boxes.txt")).read().strip().splitlines()
    assert len(lines) == len(out) and lines[0].split(",")[0] == out[0][0]


def test_v1_errors_are_codes(hiplib):
    bad = IO.cfg_text("yolov1").replace("output=1470", "output=1400")
    with pytest.raises(hiplib.YoloError, match="expects 1470"):
        hiplib.Engine(bad, max_batch=1)
    with pytest.raises(hiplib.YoloError, match="unsupported|fp8"):       # the fp8 configuration serves neither the 7x7 conv nor [connected]
        hiplib.Engine(IO.cfg_text("yolov1"), max_batch=1, dtype=hiplib.FP8)


def test_yolov1_tiny_network_and_entry_point(hiplib, tmp_path):
    """The tiny YOLOv1 of D2T/YOLO_V1_Tiny_convert_darkenet_to_Tensorflow.py (eight BN convs, six max-pools, one FC; input x / 255,
    RGB): the 1470 predictions vs the fp32 oracle, and `YOLOV1_Tiny(weights_file).detect_from_file` vs the oracle's detector with
    that script's thresholds (0.1 / 0.6 / 10)."""
    from PIL import Image
    from yolo_tensorflow_amd.yolo_v1 import YOLOV1_Tiny
    txt = IO.cfg_text("yolov1-tiny")
    secs = IO.parse_cfg(txt); flat = IO.synth_weights(secs, seed=8)
    assert flat.size == IO.weights_count(secs)
    img = np.random.default_rng(22).integers(0, 256, (448, 448, 3), dtype=np.uint8)
    osecs = R.parse_cfg(txt); params = R.unflatten_weights(flat, osecs)
    heads, _ = R.forward(osecs, params, (img.astype(np.float32) / np.float32(255))[None])
    pred = heads[0][1].reshape(-1)
    want = R.v1_rows(pred, 7, 2, 20)
    for dtype, tol in ((hiplib.FP32, 2e-3), (hiplib.BF16, 5e-2)):
        eng = hiplib.Engine(txt, max_batch=2, dtype=dtype)
        eng.set_weights(flat)
        det = eng.forward(np.stack([img, img[::-1].copy()]))
        assert det.shape == (2, 98, 25) and np.isfinite(det).all()
        assert _relmax(det[0], want) < tol, "dtype %d: %g" % (dtype, _relmax(det[0], want))
        eng.close()
    wf = str(tmp_path / "tiny-yolov1.weights"); IO.write_weights_file(wf, flat, 0, 1)
    png = str(tmp_path / "img.png"); Image.fromarray(img).save(png)
    y = YOLOV1_Tiny(wf, verbose=False, dtype=hiplib.FP32)
    out = y.detect_from_file(png, imshow=False, deteted_boxes_file=str(tmp_path / "boxes.txt"))
    y.close()
    wb, ws, wc = R.detect_v1_tf(pred, 7, 2, 20, 0.1, 0.6, 10)
    assert len(out) == len(ws) > 0
    for (name, bx, by, bw, bh, s), b, sc, c in zip(out, wb, ws, wc):
        assert name == IO.v1_classes()[int(c)] and abs(s - sc) < 2e-3
        np.testing.assert_allclose([bx, by, bw, bh], b * np.float32(448), rtol=2e-3, atol=0.5)



def test_local_layers_match_compiled_reference(hiplib, tmp_path):
    """[local] on the device (k_local: one wave per location and filter, filters streamed once) in a topology with two of them, against
    every layer output of the reference's own C code (tests/golden/mini_local.npz): fp32 to 2e-4, bf16 / fp16 storage to their
    precision; batches (the reference's layer loops images over the same filters); the export artifact carries the layer."""
    g = golden("mini_local.npz")
    cfg = str(g["cfg"])
    x = g["image_u8"][None]
    secs = IO.parse_cfg(cfg)
    for dtype, tol in ((hiplib.FP32, 2e-4), (hiplib.FP16, 4e-3), (hiplib.BF16, 3e-2)):
        eng = hiplib.Engine(cfg, max_batch=3, dtype=dtype, semantics=hiplib.SEM_DARKNET, keep_layers=True)
        eng.set_weights(g["weights"])
        det = eng.forward(x)[0]
        for i, s in enumerate(secs[1:]):
            if s["type"] == "detection":
                continue
            assert _relmax(eng.layer_output(i, 1).reshape(-1), g["layer_%02d" % i].reshape(-1)) < tol, "layer %d (%s) dtype %d" % (i, s["type"], dtype)
        np.testing.assert_allclose(det[:, 4], g["obj_raw"], rtol=tol * 10, atol=tol)
        if dtype == hiplib.FP32:
            three = np.concatenate([x, x[:, ::-1], x])                     # image 0 and 2 equal, image 1 different
            d3 = eng.forward(three)
            assert np.array_equal(d3[0], det) and np.array_equal(d3[2], det) and not np.array_equal(d3[1], det)
            path = str(tmp_path / "local.yolohip")
            eng.export(path)
            e2 = hiplib.Engine.from_file(path, max_batch=1)
            assert np.array_equal(e2.forward(x)[0], det)
            e2.close()
        eng.close()
    with pytest.raises(hiplib.YoloError, match="fp8"):
        hiplib.Engine(cfg, dtype=hiplib.FP8)


def test_local_layer_pad_channels_are_zero_and_bad_pad_is_refused(hiplib):
    """a [local] layer whose filter count is not a multiple of 8 stores its pixel in roundup(F, 8) channels; the pad channels
    must be written as zeros (the consumer multiplies them by zero filters: 0 * stale Inf would be NaN), and pad=1 with size != 3 is
    refused (darknet's own output size and im2col padding disagree there, DN/local_layer.c:10-24,103)."""
    cfg = """[net]
height=16
width=16
channels=3
[convolutional]
filters=8
size=3
stride=2
pad=1
activation=leaky
[local]
filters=5
size=3
stride=1
pad=1
activation=leaky
[convolutional]
filters=8
size=1
stride=1
activation=leaky
[connected]
output=%d
activation=linear
[detection]
classes=2
coords=4
side=2
num=1
sqrt=1
""" % (2 * 2 * (2 + 5))
    secs = IO.parse_cfg(cfg); flat = IO.synth_weights(secs, seed=1)
    x = np.random.default_rng(2).integers(0, 256, (2, 16, 16, 3), dtype=np.uint8)
    osecs = R.parse_cfg(cfg)
    heads, _ = R.forward(osecs, R.unflatten_weights(flat, osecs), x.astype(np.float32) / np.float32(255), semantics="darknet")
    for rep in range(2):                     # twice: the second engine is likely to recycle the first one's buffers
        eng = hiplib.Engine(cfg, max_batch=2, dtype=hiplib.FP32, semantics=hiplib.SEM_DARKNET)
        eng.set_weights(flat)
        eng.forward(x)
        got = eng.last_layer_output(2)
        assert np.isfinite(got).all()
        np.testing.assert_allclose(got, heads[0][1].reshape(2, -1), rtol=2e-4, atol=2e-5)
        eng.close()
    with pytest.raises(hiplib.YoloError, match="size=3"):
        hiplib.Engine(cfg.replace("filters=5\nsize=3", "filters=5\nsize=5"), max_batch=1, dtype=hiplib.FP32)
