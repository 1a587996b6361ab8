"""Pin oracle/yolo_ref.py against vectors produced by the REFERENCE itself (tools/make_golden.py):
the reference's importable numpy NMS / V2 postprocess, and its darknet C code compiled CPU-only."""
import numpy as np
import pytest
from conftest import golden
from oracle import yolo_ref as R


def test_np_nms_v3_matches_reference_bit_for_bit():
    g = golden("nms_v3_numpy.npz")
    res = R.np_nms_v3(g["det"], float(g["conf"]), float(g["iou"]))
    keys = sorted(res.keys())
    assert keys == list(g["classes"])
    counts = [len(res[k]) for k in keys]
    assert counts == list(g["counts"])
    boxes = np.concatenate([np.array([b for b, _ in res[k]], dtype=np.float32) for k in keys])
    scores = np.concatenate([np.array([s for _, s in res[k]], dtype=np.float32) for k in keys])
    assert np.array_equal(boxes, g["boxes"])
    assert np.array_equal(scores, g["scores"])       # includes the reference's off-by-one score quirk


def test_np_iou_v3_matches_reference():
    g = golden("nms_v3_numpy.npz")
    got = np.array([R.np_iou_v3(p[0], p[1]) for p in g["pairs"]], dtype=np.float64)
    assert np.array_equal(got, g["pair_ious"])
    # the documented quirk (no clamp, V3/yolo_v3.py:366): boxes disjoint in BOTH axes get a positive
    # "intersection" (-4 * -4 = 16), so the ratio is 16 / (1 + 1 - 16) instead of 0
    assert R.np_iou_v3(np.float32([0, 0, 1, 1]), np.float32([5, 5, 6, 6])) == pytest.approx(16 / (2 - 16 + 1e-5))


def test_v2_postprocess_matches_reference_bit_for_bit():
    g = golden("v2_postprocess.npz")
    b, s, c = R.v2_postprocess(g["bboxes"], g["obj"], g["cls"], image_shape=tuple(g["image_shape"]), threshold=float(g["threshold"]))
    assert np.array_equal(b, g["out_boxes"]) and b.dtype == g["out_boxes"].dtype
    assert np.array_equal(s, g["out_scores"])
    assert np.array_equal(c, g["out_classes"])
    assert len(s) >= 9 and len(set(c.tolist())) >= 2   # different-class overlap survives (V2/utils.py:183)


def test_v2_bboxes_iou_matches_reference():
    g = golden("v2_postprocess.npz")
    ib = g["int_boxes"]
    assert np.array_equal(R.v2_bboxes_iou(ib[0], ib[1:]), g["int_ious"], equal_nan=True)


@pytest.mark.parametrize("name", ["mini_v3.npz", "mini_v2.npz"])
def test_forward_matches_compiled_darknet_every_layer(name):
    """semantics='darknet' + darknet's CPU batch-norm epsilon: every layer output of the reference's own C
    forward pass (conv/BN/leaky, shortcut, route, maxpool 2/2 and 2/1, nearest upsample, reorg)."""
    g = golden(name)
    secs = R.parse_cfg(str(g["cfg"]))
    params = R.unflatten_weights(g["weights"], secs)
    x = g["image_u8"].astype(np.float32)[None] / np.float32(255.0)
    heads, outs = R.forward(secs, params, x, semantics="darknet", bn_mode="darknet_cpu", collect=True)
    for i, o in enumerate(outs):
        ref = g["layer_%02d" % i]
        if o is None:      # yolo / region layer: darknet stores its activated copy, checked below
            continue
        assert o.shape == ref.shape, (i, o.shape, ref.shape)
        np.testing.assert_allclose(o, ref, rtol=2e-5, atol=2e-5, err_msg="layer %d" % i)


def test_yolo_decode_matches_darknet_boxes():
    """get_yolo_detections (DN/yolo_layer.c:316-343) == _ratio_detection_layer on the same head tensors."""
    g = golden("mini_v3.npz")
    secs = R.parse_cfg(str(g["cfg"]))
    layers = secs[1:]
    dets = []
    for i, s in enumerate(layers):
        if s["type"] == "yolo":
            raw = g["layer_%02d" % (i - 1)]
            dets.append(R.detection_layer_ratio(raw, R.yolo_anchors(s), (64, 64)))
    det = np.concatenate(dets, axis=1)[0]
    thresh = float(g["thresh"])
    keep = det[:, 4] > thresh          # darknet gates on objectness (DN/yolo_layer.c:327)
    mine = det[keep]
    ref_b = g["boxes_raw"]
    assert len(mine) == len(ref_b)
    np.testing.assert_allclose(mine[:, :4], ref_b, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(mine[:, 4], g["obj_raw"], rtol=1e-5, atol=1e-6)
    prob = mine[:, 4:5] * mine[:, 5:]
    prob = np.where(prob > thresh, prob, 0)
    np.testing.assert_allclose(prob, g["prob_raw"], rtol=1e-5, atol=1e-6)


def test_dn_nms_sort_matches_darknet():
    g = golden("mini_v3.npz")
    got = R.dn_nms_sort(g["boxes_raw"], g["prob_raw"], float(g["nms"]))
    # do_nms_sort reorders the array; compare as multisets of (box, prob-row)
    def key(b, p):
        return sorted(map(tuple, np.concatenate([b, p], axis=1).round(6).tolist()))
    assert key(g["boxes_raw"], got) == key(g["boxes_nms"], g["prob_nms"])
    assert (got > 0).sum() < (g["prob_raw"] > 0).sum()


def test_region_decode_matches_darknet_boxes():
    g = golden("mini_v2.npz")
    secs = R.parse_cfg(str(g["cfg"]))
    s = secs[-1]
    raw = g["layer_%02d" % (len(secs) - 3)]
    boxes, obj, cls = R.region_decode(raw, R.yolo_anchors(s), int(s["classes"]))
    # get_region_detections writes dets[n*w*h + i] (DN/region_layer.c:364-439): anchor-major order
    A = boxes.shape[2]
    b = boxes[0].transpose(1, 0, 2).reshape(-1, 4)
    obj = obj[0].transpose(1, 0); cls = cls[0].transpose(1, 0, 2)
    cxcywh = np.stack([(b[:, 0] + b[:, 2]) / 2, (b[:, 1] + b[:, 3]) / 2, b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]], -1)
    np.testing.assert_allclose(cxcywh, g["boxes_raw"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(obj.reshape(-1), g["obj_raw"], rtol=1e-5, atol=1e-6)
    prob = obj.reshape(-1, 1) * cls.reshape(-1, cls.shape[-1])
    thresh = float(g["thresh"])
    np.testing.assert_allclose(np.where(prob > thresh, prob, 0), g["prob_raw"], rtol=1e-4, atol=1e-6)


def test_v1_restatement_matches_compiled_reference():
    """Rows Net1 / D1: the oracle's [connected] (CHW flatten), 7x7/2 conv, [dropout] and [detection] decode against every layer
    output and get_network_boxes of the reference's own C code (tests/golden/mini_v1.npz, tools/make_golden.py gen_mini_v1)."""
    g = golden("mini_v1.npz")
    cfg = str(g["cfg"])
    secs = R.parse_cfg(cfg); params = R.unflatten_weights(g["weights"], secs)
    x = (g["image_u8"].astype(np.float32) / np.float32(255)) * np.float32(2) - np.float32(1)
    heads, outs = R.forward(secs, params, x[None], collect=True)
    for i, o in enumerate(outs):
        if o is not None:
            ref = g["layer_%02d" % i].reshape(o.shape)
            assert np.abs(o - ref).max() <= 4e-6 * max(1.0, float(np.abs(ref).max())), "layer %d" % i
    rows = R.v1_rows(heads[0][1][0], 3, 2, 20)
    np.testing.assert_allclose(rows[:, :4] * np.float32(64), g["boxes_raw"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(rows[:, 4], g["obj_raw"], rtol=1e-6, atol=1e-7)
    pr = rows[:, 4:5] * rows[:, 5:]
    np.testing.assert_allclose(np.where(pr > g["thresh"], pr, 0), g["prob_raw"], rtol=1e-5, atol=1e-6)
    # and the TF-side detector on the same predictions: v1_decode is the thresholded subset of v1_rows, best-first after the NMS
    b, s, c = R.detect_v1_tf(heads[0][1][0].reshape(-1), 3, 2, 20, 0.2, 0.4, 10)
    full = R.v1_rows(heads[0][1][0], 3, 2, 20)
    smax = (full[:, 4:5] * full[:, 5:]).max(-1)
    assert len(s) > 0 and (np.diff(s) <= 0).all() and set(np.round(s, 6)).issubset(set(np.round(smax[smax >= 0.2], 6)))


def test_local_layer_restatement_matches_compiled_reference():
    """[local] (locally connected, DN/local_layer.c:91-120; darknet's own yolov1.cfg): the oracle's restatement -- biases [filter][location],
    weights [location][filter][c][kh][kw], `pad` as flag and im2col amount -- against every layer output of the reference's C code on a
    topology with a same-size 3x3 / pad 1 and a 2x2 / stride 2 / unpadded local layer (tests/golden/mini_local.npz, gen_mini_local)."""
    g = golden("mini_local.npz")
    cfg = str(g["cfg"])
    secs = R.parse_cfg(cfg); params = R.unflatten_weights(g["weights"], secs)
    assert [s["type"] for s in secs[1:]].count("local") == 2
    np.testing.assert_array_equal(R.flatten_weights(params, secs), g["weights"])                 # the [local] parameter layout round-trips
    x = g["image_u8"].astype(np.float32) / np.float32(255)
    heads, outs = R.forward(secs, params, x[None], semantics="darknet", bn_mode="darknet", collect=True)
    for i, o in enumerate(outs):
        if o is not None:
            ref = g["layer_%02d" % i].reshape(o.shape)
            assert np.abs(o - ref).max() <= 4e-6 * max(1.0, float(np.abs(ref).max())), "layer %d" % i
    rows = R.v1_rows(heads[0][1][0], 3, 2, 2)
    np.testing.assert_allclose(rows[:, 4], g["obj_raw"], rtol=1e-5, atol=1e-6)
