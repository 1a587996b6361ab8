"""Parity of every production kernel, called through the C ABI (yolo_op_*), against the oracle on the same
seeded inputs.  Integer/index results (kept sets, labels, copies) are bit-exact; floating point tolerances
are stated next to each check."""
import numpy as np
import pytest
from conftest import golden
from oracle import yolo_ref as R

pytestmark = pytest.mark.gpu

# the 23 unique (k, stride, H, Cin, Cout) conv shapes of YOLOv3-416
V3_SHAPES = [
    (3, 1, 416, 3, 32), (3, 2, 416, 32, 64), (1, 1, 208, 64, 32), (3, 1, 208, 32, 64), (3, 2, 208, 64, 128),
    (1, 1, 104, 128, 64), (3, 1, 104, 64, 128), (3, 2, 104, 128, 256), (1, 1, 52, 256, 128), (1, 1, 52, 384, 128),
    (1, 1, 52, 256, 255), (3, 1, 52, 128, 256), (3, 2, 52, 256, 512), (1, 1, 26, 512, 256), (1, 1, 26, 768, 256),
    (1, 1, 26, 512, 255), (1, 1, 26, 256, 128), (3, 1, 26, 256, 512), (3, 2, 26, 512, 1024), (1, 1, 13, 1024, 512),
    (1, 1, 13, 1024, 255), (1, 1, 13, 512, 256), (3, 1, 13, 512, 1024),
]


def _conv_case(rng, k, s, h, cin, cout, n=2, residual=False):
    h = min(h, 26 if cin * cout < 65536 else 13)
    if s == 2:
        h += h % 2
    x = R.to_bf16(rng.standard_normal((n, h, h, cin)).astype(np.float32))
    w = R.to_bf16((rng.standard_normal((k, k, cin, cout)) * np.sqrt(2.0 / (k * k * cin))).astype(np.float32))
    b = rng.standard_normal(cout).astype(np.float32)
    ref = R.leaky_relu(R.conv2d_nhwc(x, w, s) + b)
    res = None
    if residual:
        res = R.to_bf16(rng.standard_normal(ref.shape).astype(np.float32))
        ref = R.to_bf16(ref) + res
    return x, w, b, res, ref


def _assert_bf16_close(got, ref, scale=None):
    # both operands are bf16-exact, accumulation is fp32: the only differences are summation order and the
    # final round to bf16 -> at most one bf16 ulp (2^-8 relative) plus slack for cancellation near zero.
    # `scale`: magnitude the ulp refers to when the result is a sum of separately rounded terms (residual).
    want = R.to_bf16(ref)
    err = np.abs(got - want)
    tol = 2.0 ** -7 * (np.abs(want) if scale is None else scale) + 2e-3
    assert (err <= tol).all(), "max err %.3e at %s" % (err.max(), np.unravel_index(err.argmax(), err.shape))


@pytest.mark.parametrize("shape", V3_SHAPES, ids=lambda s: "k%d_s%d_h%d_%dto%d" % s)
def test_conv_bf16_all_yolov3_shapes(hiplib, shape):
    k, s, h, cin, cout = shape
    rng = np.random.default_rng(hash(shape) % 2 ** 32)
    x, w, b, _, ref = _conv_case(rng, k, s, h, cin, cout)
    got = hiplib.op_conv2d(x, w, b, stride=s, act=1)
    assert got.shape == ref.shape
    _assert_bf16_close(got, ref)


@pytest.mark.parametrize("shape", [(3, 2, 208, 64, 128), (1, 1, 104, 128, 64), (3, 1, 104, 64, 128), (3, 1, 208, 32, 64), (3, 2, 416, 32, 64)],
                         ids=lambda s: "k%d_s%d_h%d_%dto%d" % s)
def test_conv_bf16_early_layers_at_full_size(hiplib, shape):
    """The large-spatial-extent layers of YOLOv3-416 at their REAL size (the 23-shape test above shrinks H to <= 26): 104x104 and
    208x208 stages, one image, against the oracle; with the fused shortcut where the network has one."""
    k, s, h, cin, cout = shape
    rng = np.random.default_rng(h * 31 + cin)
    x = R.to_bf16(rng.standard_normal((1, h, h, cin)).astype(np.float32))
    w = R.to_bf16((rng.standard_normal((k, k, cin, cout)) * np.sqrt(2.0 / (k * k * cin))).astype(np.float32))
    b = rng.standard_normal(cout).astype(np.float32)
    ref = R.leaky_relu(R.conv2d_nhwc(x, w, s) + b)
    _assert_bf16_close(hiplib.op_conv2d(x, w, b, stride=s, act=1), ref)
    if k == 3 and s == 1:
        res = R.to_bf16(rng.standard_normal(ref.shape).astype(np.float32))
        _assert_bf16_close(hiplib.op_conv2d(x, w, b, stride=s, act=1, residual=res), R.to_bf16(ref) + res, scale=np.abs(ref) + np.abs(res))


@pytest.mark.parametrize("shape", [(3, 1, 26, 64, 128), (1, 1, 13, 1024, 255), (3, 2, 26, 32, 64), (3, 1, 20, 3, 32)],
                         ids=lambda s: "k%d_s%d_h%d_%dto%d" % s)
def test_conv_bf16_every_tile_config_agrees(hiplib, shape):
    k, s, h, cin, cout = shape
    rng = np.random.default_rng(7)
    x, w, b, _, ref = _conv_case(rng, k, s, h, cin, cout, n=3)
    outs = []
    for c in range(hiplib.op_conv_num_cfgs()):
        try:
            outs.append(hiplib.op_conv2d(x, w, b, stride=s, act=1, tile_cfg=c))
        except hiplib.YoloError as e:            # the halo-staged forms only take 3x3 / stride 1 / multiples of 13
            assert "not applicable" in str(e)
    assert len(outs) >= 36
    _assert_bf16_close(outs[0], ref)
    for o in outs[1:]:
        assert np.array_equal(o, outs[0])       # same K order in every tiling -> bit-identical


HALO_CFGS = (36, 37, 38, 39, 40, 41, 42, 43)


@pytest.mark.parametrize("case", [(2, 13, 512, 1024, False), (2, 26, 256, 512, True), (1, 52, 128, 256, True), (1, 104, 64, 128, True),
                                  (3, 13, 64, 96, False), (1, 26, 192, 200, True)],
                         ids=lambda c: "n%d_h%d_%dto%d_res%d" % c)
@pytest.mark.parametrize("dtype_name", ["bf16", "fp8"])
def test_conv_halo_staged_form(hiplib, case, dtype_name):
    """The halo-staged 3x3 form (13x13 pixel block per workgroup, input halo staged once per channel chunk, taps as LDS
    offsets) at the real YOLOv3-416 stage shapes, un-shrunk: vs the oracle (bf16) and bit-identical to the tiled form
    (bf16 and fp8), with and without the fused shortcut, ragged channel counts included."""
    n, h, cin, cout, residual = case
    dtype = hiplib.BF16 if dtype_name == "bf16" else hiplib.FP8
    if dtype == hiplib.FP8 and cin % 128:
        pytest.skip("fp8 halo form needs whole 128-channel chunks")
    rng = np.random.default_rng(n * 1000 + h + cin)
    x = R.to_bf16(rng.standard_normal((n, h, h, cin)).astype(np.float32))
    w = R.to_bf16((rng.standard_normal((3, 3, cin, cout)) * np.sqrt(2.0 / (9 * cin))).astype(np.float32))
    b = rng.standard_normal(cout).astype(np.float32)
    res = R.to_bf16(rng.standard_normal((n, h, h, cout)).astype(np.float32)) if residual else None
    tiled = hiplib.op_conv2d(x, w, b, act=1, residual=res, dtype=dtype, tile_cfg=16)
    if dtype == hiplib.BF16:
        ref = R.leaky_relu(R.conv2d_nhwc(x, w, 1) + b)
        if residual:
            _assert_bf16_close(tiled, R.to_bf16(ref) + res, scale=np.abs(ref) + np.abs(res))
        else:
            _assert_bf16_close(tiled, ref)
    for cfg in HALO_CFGS:
        got = hiplib.op_conv2d(x, w, b, act=1, residual=res, dtype=dtype, tile_cfg=cfg)
        assert np.array_equal(got, tiled), "halo cfg %d differs from the tiled form" % cfg
    with pytest.raises(hiplib.YoloError, match="not applicable"):
        hiplib.op_conv2d(x[:, :12, :12], w, b, act=1, dtype=dtype, tile_cfg=36)


def test_conv_bf16_residual_and_linear(hiplib):
    rng = np.random.default_rng(9)
    x, w, b, res, ref = _conv_case(rng, 3, 1, 26, 64, 128, residual=True)
    _assert_bf16_close(hiplib.op_conv2d(x, w, b, act=1, residual=res), ref, scale=np.abs(ref - res) + np.abs(res))
    lin = R.conv2d_nhwc(x, w, 1) + b
    _assert_bf16_close(hiplib.op_conv2d(x, w, b, act=0), lin)
    _assert_bf16_close(hiplib.op_conv2d(x, w, None, act=0), R.conv2d_nhwc(x, w, 1))


def test_conv_ragged_edges(hiplib):
    """pixel count and channel count not multiples of any tile; single pixel; batch 1."""
    rng = np.random.default_rng(10)
    for (n, h, cin, cout, k, s) in ((1, 1, 8, 8, 1, 1), (1, 7, 24, 40, 3, 1), (3, 9, 16, 72, 3, 2), (1, 5, 8, 255, 1, 1)):
        x = R.to_bf16(rng.standard_normal((n, h, h, cin)).astype(np.float32))
        w = R.to_bf16(rng.standard_normal((k, k, cin, cout)).astype(np.float32) * 0.2)
        got = hiplib.op_conv2d(x, w, None, stride=s, act=0)
        _assert_bf16_close(got, R.conv2d_nhwc(x, w, s))


@pytest.mark.parametrize("shape", [(3, 1, 26, 64, 128), (1, 1, 13, 512, 255), (3, 2, 26, 32, 64), (3, 1, 20, 3, 32), (3, 1, 13, 256, 512)],
                         ids=lambda s: "k%d_s%d_h%d_%dto%d" % s)
def test_conv_fp32_exact_path(hiplib, shape):
    """f32-input MFMA: only summation order differs from the fp32 oracle -> rtol 1e-4 of the tensor scale."""
    k, s, h, cin, cout = shape
    rng = np.random.default_rng(11)
    x = rng.standard_normal((2, h, h, cin)).astype(np.float32)
    w = (rng.standard_normal((k, k, cin, cout)) * np.sqrt(2.0 / (k * k * cin))).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    res = rng.standard_normal((2, (h + 2 * (k // 2) - k) // s + 1, (h + 2 * (k // 2) - k) // s + 1, cout)).astype(np.float32)
    ref = R.leaky_relu(R.conv2d_nhwc(x, w, s) + b) + res
    got = hiplib.op_conv2d(x, w, b, stride=s, act=1, residual=res, dtype=hiplib.FP32)
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-4 * np.abs(ref).max())


def test_upsample(hiplib):
    rng = np.random.default_rng(12)
    for shape in ((2, 13, 13, 256), (1, 26, 26, 128), (1, 1, 1, 8), (3, 5, 7, 16)):
        x = R.to_bf16(rng.standard_normal(shape).astype(np.float32))
        # TF bilinear `_upsample`: same fp32 lerp order as the oracle, rounded once to bf16 -> bit-exact
        assert np.array_equal(hiplib.op_upsample2x(x, hiplib.SEM_TF), R.to_bf16(R.upsample_tf(x)))
        assert np.array_equal(hiplib.op_upsample2x(x, hiplib.SEM_DARKNET), R.upsample_nearest(x))


def test_reorg_and_maxpool(hiplib):
    rng = np.random.default_rng(13)
    x = R.to_bf16(rng.standard_normal((2, 26, 26, 64)).astype(np.float32))
    assert np.array_equal(hiplib.op_reorg(x, 2, hiplib.SEM_TF), R.space_to_depth(x, 2))
    assert np.array_equal(hiplib.op_reorg(x, 2, hiplib.SEM_DARKNET), R.reorg_darknet(x, 2))
    for shape in ((2, 26, 26, 64), (1, 13, 13, 512), (1, 7, 7, 8)):
        x = R.to_bf16(rng.standard_normal(shape).astype(np.float32))
        if shape[1] % 2 == 0:
            assert np.array_equal(hiplib.op_maxpool(x, 2, 2), R.max_pool(x, 2, 2))
        assert np.array_equal(hiplib.op_maxpool(x, 2, 1), R.max_pool(x, 2, 1))     # 'SAME' stride-1 pool, -inf padding


def test_resize_u8_legacy_bilinear(hiplib):
    rng = np.random.default_rng(14)
    for (h, w, s) in ((576, 768, 416), (37, 91, 64), (416, 416, 416), (900, 1352, 608)):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        got = hiplib.op_resize_u8(img, s)
        ref = R.input_process(img, s)[0]
        np.testing.assert_allclose(got, ref, rtol=0, atol=2e-6)      # same op order; a 1-ulp slack on the scale factor


@pytest.mark.parametrize("mode", ["ratio", "pixel"])
def test_decode_yolo(hiplib, mode):
    rng = np.random.default_rng(15)
    anchors = [(116, 90), (156, 198), (373, 326)]
    for g in (13, 26):
        raw = (rng.standard_normal((2, g, g, 255)) * 2).astype(np.float32)
        fn = R.detection_layer_ratio if mode == "ratio" else R.detection_layer_pixel
        ref = fn(raw, anchors, (32 * g, 32 * g))
        got = hiplib.op_decode(raw, anchors, 80, 32 * g, hiplib.DECODE_RATIO if mode == "ratio" else hiplib.DECODE_PIXEL)
        # expf/sigmoid differ from numpy's by a few ulp
        np.testing.assert_allclose(got, ref, rtol=3e-6, atol=1e-7)


def test_decode_region(hiplib):
    rng = np.random.default_rng(16)
    anchors = [(0.57273, 0.677385), (1.87446, 2.06253), (3.33843, 5.47434), (7.88282, 3.52778), (9.77052, 9.16828)]
    raw = (rng.standard_normal((2, 13, 13, 425)) * 2).astype(np.float32)
    boxes, obj, cls = R.region_decode(raw, anchors, 80)
    got = hiplib.op_decode(raw, anchors, 80, 416, region=True).reshape(2, 169, 5, 85)
    corners = np.stack([got[..., 0] - got[..., 2] / 2, got[..., 1] - got[..., 3] / 2,
                        got[..., 0] + got[..., 2] / 2, got[..., 1] + got[..., 3] / 2], -1)
    np.testing.assert_allclose(corners, boxes, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(got[..., 4], obj, rtol=3e-6)
    np.testing.assert_allclose(got[..., 5:], cls, rtol=1e-5, atol=1e-8)


# ---------------------------------------------------------------------------------------------------
def _planted(rng, n, rows, classes, clusters=10, per=8, unit=1.0):
    det = np.zeros((n, rows, 5 + classes), np.float32)
    det[..., 0:2] = rng.uniform(0.1, 0.9, (n, rows, 2)) * unit
    det[..., 2:4] = rng.uniform(0.02, 0.3, (n, rows, 2)) * unit
    det[..., 4] = rng.uniform(0, 0.45, (n, rows))
    det[..., 5:] = rng.uniform(0, 1, (n, rows, classes))
    for b in range(n):
        idx = rng.permutation(rows)[:clusters * per].reshape(clusters, per)
        for c in range(clusters):
            cx, cy, w, h = rng.uniform(0.2, 0.8) * unit, rng.uniform(0.2, 0.8) * unit, rng.uniform(0.1, 0.3) * unit, rng.uniform(0.1, 0.3) * unit
            cls = rng.integers(0, classes)
            for r in idx[c]:
                det[b, r, :4] = [cx + rng.normal(0, .01) * unit, cy + rng.normal(0, .01) * unit, w + rng.normal(0, .01) * unit, h + rng.normal(0, .01) * unit]
                det[b, r, 4] = rng.uniform(0.6, 1)
                det[b, r, 5:] = rng.uniform(0, 0.3, classes); det[b, r, 5 + cls] = rng.uniform(0.85, 1)
    return det.astype(np.float32)


def _check_tf(hiplib, det, thr, iou, max_out):
    got = hiplib.op_postprocess(det, thr, iou, max_out, hiplib.NMS_TF, hiplib.SELECT_GT)
    for b in range(det.shape[0]):
        boxes, scores, classes = R.detect_v3_tf(det[b], thr, iou, max_out)
        g = got[b]
        assert len(g) == len(scores)
        assert np.array_equal(g["score"], scores) and np.array_equal(g["cls"], classes)
        assert np.array_equal(np.stack([g["x0"], g["y0"], g["x1"], g["y1"]], -1).reshape(-1, 4), boxes.reshape(-1, 4))
    return got


def test_postprocess_tf_nms_bit_exact(hiplib):
    rng = np.random.default_rng(17)
    det = _planted(rng, 3, 10647, 80)
    got = _check_tf(hiplib, det, 0.5, 0.5, 20)
    assert all(5 <= len(g) <= 20 for g in got)
    _check_tf(hiplib, det, 0.4, 0.4, 10)          # the converter's flags (D2T ...py:42-44)
    _check_tf(hiplib, det, 0.5, 0.5, 3)           # max_output_size cap
    _check_tf(hiplib, det, 0.999999, 0.5, 20)     # nothing passes -> empty


def test_postprocess_edge_cases(hiplib):
    rng = np.random.default_rng(18)
    # every candidate passes (sort > LDS capacity -> global-memory sort path), ties in score
    det = _planted(rng, 1, 6000, 4, clusters=40, per=20)
    det[..., 4] = np.maximum(det[..., 4], 0.7); det[..., 5] = 0.9
    det[0, 100:200, 4] = 0.8; det[0, 100:200, 5:] = 0.5; det[0, 100:200, 5] = 1.0     # 100-way tie
    _check_tf(hiplib, det, 0.1, 0.5, 50)
    # a single row; rows that are not a multiple of the workgroup
    _check_tf(hiplib, _planted(rng, 2, 1, 3, clusters=1, per=1), 0.3, 0.5, 5)
    _check_tf(hiplib, _planted(rng, 2, 1025, 3), 0.3, 0.5, 100)
    # degenerate (zero-area) boxes never suppress each other
    d = _planted(rng, 1, 64, 2, clusters=2, per=4); d[..., 2] = 0
    _check_tf(hiplib, d, 0.3, 0.5, 64)


def test_postprocess_darknet_per_class(hiplib):
    rng = np.random.default_rng(19)
    det = _planted(rng, 2, 2028, 20, clusters=12, per=6)
    got = hiplib.op_postprocess(det, 0.5, 0.45, 400, hiplib.NMS_DARKNET, hiplib.SELECT_GT)
    for b in range(2):
        _, scores, labels, idx = R.select_threshold(det[b], 0.5)
        probs = np.zeros((len(idx), 20), np.float32); probs[np.arange(len(idx)), labels] = scores
        kept = R.dn_nms_sort(det[b, idx, :4], probs, 0.45)
        want = sorted((float(kept[i, labels[i]]), int(labels[i])) for i in range(len(idx)) if kept[i, labels[i]] > 0)
        have = sorted((float(s), int(c)) for s, c in zip(got[b]["score"], got[b]["cls"]))
        assert have == want and len(have) > 0


def test_postprocess_v2_numpy_flavour(hiplib):
    """V2 utils.postprocess: int32 pixel boxes, clip, score > thr, top-400, class-aware NMS in float64.
    The oracle restatement is pinned bit-for-bit to the reference by tests/test_oracle_golden.py."""
    rng = np.random.default_rng(20)
    det = _planted(rng, 2, 845, 80, clusters=10, per=9)
    h, w = 576, 768
    got = hiplib.op_postprocess(det, 0.5, 0.5, 400, hiplib.NMS_PER_CLASS, hiplib.SELECT_GT, image_hw=(h, w))
    for b in range(2):
        d = det[b]
        corners = np.stack([d[:, 0] - d[:, 2] * np.float32(.5), d[:, 1] - d[:, 3] * np.float32(.5),
                            d[:, 0] + d[:, 2] * np.float32(.5), d[:, 1] + d[:, 3] * np.float32(.5)], -1)
        bb, ss, cc = R.v2_postprocess(corners, d[:, 4], d[:, 5:], image_shape=(h, w), threshold=0.5)
        g = got[b]
        assert len(g) == len(ss) > 0
        assert np.array_equal(g["score"], ss.astype(np.float32)) and np.array_equal(g["cls"], cc)
        assert np.array_equal(np.stack([g["x0"], g["y0"], g["x1"], g["y1"]], -1).astype(np.int32), bb)


def test_nms_detections_op_vs_oracle(hiplib):
    """do_nms_sort semantics on caller arrays (yolo_op_nms_detections) against the oracle restatement of DN/box.c:58-89
    (itself pinned to the compiled reference by tests/test_oracle_golden.py), including detections with objectness 0
    (which must neither suppress nor be touched), a single detection and n = 0."""
    rng = np.random.default_rng(33)
    for n, classes in ((1, 3), (37, 5), (700, 20), (2500, 4)):
        clusters = max(1, n // 12)
        centres = rng.uniform(0.2, 0.8, (clusters, 2)).astype(np.float32)
        which = rng.integers(0, clusters, n)
        boxes = np.concatenate([centres[which] + rng.normal(0, 0.01, (n, 2)).astype(np.float32),
                                rng.uniform(0.1, 0.2, (n, 2)).astype(np.float32)], 1).astype(np.float32)
        prob = (rng.uniform(0, 1, (n, classes)) * (rng.uniform(0, 1, (n, classes)) > 0.5)).astype(np.float32)
        obj = rng.uniform(0.1, 1, n).astype(np.float32)
        obj[rng.uniform(0, 1, n) < 0.1] = 0.0
        live = obj != 0
        want = prob.copy()
        want[live] = R.dn_nms_sort(boxes[live], prob[live], 0.45)
        got_p, got_o = hiplib.op_nms_detections(boxes, prob, obj, 0.45)
        assert np.array_equal(got_o, obj)
        assert np.array_equal(got_p, want)
        assert n < 30 or (got_p == 0).sum() > (prob == 0).sum()
    p, o = hiplib.op_nms_detections(np.zeros((0, 4), np.float32), np.zeros((0, 3), np.float32), np.zeros(0, np.float32), 0.45)
    assert p.shape == (0, 3) and o.shape == (0,)
    with pytest.raises(hiplib.YoloError):
        hiplib.op_nms_detections(np.zeros((5000, 4), np.float32), np.zeros((5000, 2), np.float32), np.ones(5000, np.float32), 0.45)


@pytest.mark.parametrize("dtype", ["bf16", "fp16", "fp8"])
def test_halo_form_on_ragged_blocks_equals_tiled_form(hiplib, dtype):
    """Round 4: the halo-staged 3x3 forms on sizes that are NOT multiples of 13 (608 x 608 networks: 38 = 3 * 13 - 1, 76 = 6 * 13 - 2):
    ragged 13 x 13 blocks on the bottom / right edge, whose columns past the image are computed on zeros and never stored.  Every halo
    configuration (barrier-per-K-step, role-split, free-running, 3 stages) is bit-identical to the tiled form, with and without the fused
    shortcut; 19 x 19 (would compute 26 x 26) is refused."""
    dt = {"bf16": hiplib.BF16, "fp16": hiplib.FP16, "fp8": hiplib.FP8}[dtype]
    rng = np.random.default_rng(5)
    cin = 128
    for (n, h, cout) in ((3, 38, 256), (2, 76, 128), (2, 25, 128)):
        x = rng.standard_normal((n, h, h, cin)).astype(np.float32)
        w = (rng.standard_normal((3, 3, cin, cout)) * 0.05).astype(np.float32)
        b = rng.standard_normal(cout).astype(np.float32)
        res = rng.standard_normal((n, h, h, cout)).astype(np.float32)
        for r in (None, res):
            want = hiplib.op_conv2d(x, w, b, act=1, residual=r, dtype=dt, tile_cfg=16)
            for cfg in range(36, 44):
                got = hiplib.op_conv2d(x, w, b, act=1, residual=r, dtype=dt, tile_cfg=cfg)
                assert np.array_equal(got, want), (dtype, h, cout, cfg, r is not None)
    x = rng.standard_normal((1, 19, 19, cin)).astype(np.float32)
    with pytest.raises(hiplib.YoloError, match="not applicable"):
        hiplib.op_conv2d(x, (rng.standard_normal((3, 3, cin, 128)) * 0.05).astype(np.float32), None, dtype=dt, tile_cfg=40)


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_halo_form_on_rectangular_blocks_equals_tiled_form(hiplib, dtype):
    """Round 5: the free-running halo form on 10 x 19 blocks (tile configurations 54: x 256 channels, 55: x 128) and 5 x 19 strips (56: x 128) --
    what tiles the 608 x 608 network's 76 / 38 / 19 grids into exactly 256 workgroups at 8 images per GPU -- against
    the tiled form: bit for bit (same K order), with and without the fused shortcut, on the grids they are for, on grids where the last
    block row is ragged (76 = 8 x 10 - 4, 38 = 4 x 10 - 2, 19 = 4 x 5 - 1) and on ragged columns too (37 x 37); a grid they would waste more
    than 15 % of is refused, and so are e4m3 operands."""
    dt = {"bf16": hiplib.BF16, "fp16": hiplib.FP16}[dtype]
    rng = np.random.default_rng(7)
    for (n, h, cin, cout, cfgs) in ((2, 76, 128, 256, (54, 55, 56)), (3, 38, 256, 512, (54, 55, 56)), (3, 19, 128, 256, (54, 55, 56)), (1, 37, 64, 128, (55, 56)), (2, 57, 64, 128, (55, 56))):
        x = rng.standard_normal((n, h, h, cin)).astype(np.float32)
        w = (rng.standard_normal((3, 3, cin, cout)) * 0.05).astype(np.float32)
        b = rng.standard_normal(cout).astype(np.float32)
        res = rng.standard_normal((n, h, h, cout)).astype(np.float32)
        for r in (None, res):
            want = hiplib.op_conv2d(x, w, b, act=1, residual=r, dtype=dt, tile_cfg=16)
            for cfg in cfgs:
                got = hiplib.op_conv2d(x, w, b, act=1, residual=r, dtype=dt, tile_cfg=cfg)
                assert np.array_equal(got, want), (dtype, h, cout, cfg, r is not None)
    x = rng.standard_normal((1, 13, 13, 128)).astype(np.float32)          # 13 x 13 under 10 x 19 blocks: 2 x 1 blocks of 190 for 169 pixels = 2.2 x
    w = (rng.standard_normal((3, 3, 128, 128)) * 0.05).astype(np.float32)
    with pytest.raises(hiplib.YoloError, match="not applicable"):
        hiplib.op_conv2d(x, w, None, dtype=dt, tile_cfg=54)
    with pytest.raises(hiplib.YoloError):
        hiplib.op_conv2d(rng.standard_normal((1, 19, 19, 128)).astype(np.float32), w, None, dtype=hiplib.FP8, tile_cfg=56)
