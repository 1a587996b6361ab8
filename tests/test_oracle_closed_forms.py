"""TF-only operators have no reference artefact to pin them ("parity unpinned"); the oracle's
restatements are checked against hand-derived closed forms and small hand-worked cases."""
import numpy as np
from oracle import yolo_ref as R


def test_upsample_literal_equals_closed_form():
    rng = np.random.default_rng(0)
    for shape in ((1, 13, 13, 8), (2, 5, 7, 3), (1, 1, 1, 4), (1, 2, 2, 1)):
        x = rng.standard_normal(shape).astype(np.float32)
        a = R.upsample_tf(x)               # pad SYMMETRIC -> legacy resize -> crop, literally (V3/yolo_v3.py:162-192)
        b = R.upsample_tf_closed_form(x)   # out[2i]=in[i]; out[2i+1]=in[i]+(in[i+1]-in[i])/2, clamped
        assert a.shape == (shape[0], 2 * shape[1], 2 * shape[2], shape[3])
        assert np.array_equal(a, b)


def test_upsample_hand_case():
    x = np.array([[1., 3.], [5., 9.]], dtype=np.float32).reshape(1, 2, 2, 1)
    out = R.upsample_tf(x)[0, :, :, 0]
    want = np.array([[1, 2, 3, 3], [3, 4.5, 6, 6], [5, 7, 9, 9], [5, 7, 9, 9]], dtype=np.float32)
    assert np.array_equal(out, want)


def test_resize_legacy_no_half_pixel():
    # 2 -> 4 along x: src = dst*0.5 -> [a, (a+b)/2, b, b]  (legacy: clamped upper index, no half-pixel offset)
    img = np.array([[[10.], [20.]]], dtype=np.float32)
    out = R.resize_bilinear_legacy(img, 1, 4)[0, :, 0]
    assert np.array_equal(out, np.float32([10, 15, 20, 20]))
    # identity when sizes match; downscale by 2 picks even samples
    x = np.arange(16, dtype=np.float32).reshape(4, 4, 1)
    assert np.array_equal(R.resize_bilinear_legacy(x, 4, 4), x)
    assert np.array_equal(R.resize_bilinear_legacy(x, 2, 2)[:, :, 0], x[::2, ::2, 0])


def test_resize_cv2_half_pixel_rule():
    """oracle.resize_cv2_linear (cv2.resize INTER_LINEAR on float32, V2/utils.py:19) against hand-derived cases of OpenCV's rule
    (half-pixel centres, clamped borders; cv2 itself is absent: parity unpinned), and torch's align_corners=False bilinear, which uses
    the same sample positions (an independent implementation; last-ulp differences from its other operation order allowed)."""
    import torch
    x = np.arange(12, dtype=np.float32).reshape(3, 4, 1)
    assert np.array_equal(R.resize_cv2_linear(x, 3, 4), x)                        # same size: identity (fx == 0 everywhere)
    up = R.resize_cv2_linear(x[:1, :2], 1, 4)[0, :, 0]                             # [0, 1] -> 4 samples at -0.25, 0.25, 0.75, 1.25
    np.testing.assert_array_equal(up, np.array([0.0, 0.25, 0.75, 1.0], np.float32))
    dn = R.resize_cv2_linear(x[:1], 1, 2)[0, :, 0]                                 # [0,1,2,3] -> samples at 0.5, 2.5
    np.testing.assert_array_equal(dn, np.array([0.5, 2.5], np.float32))
    rng = np.random.default_rng(0)
    img = rng.random((37, 53, 3), dtype=np.float32) * 255
    for oh, ow in ((416, 416), (20, 31), (64, 40)):
        got = R.resize_cv2_linear(img, oh, ow)
        ref = torch.nn.functional.interpolate(torch.from_numpy(img).permute(2, 0, 1)[None], size=(oh, ow), mode="bilinear", align_corners=False)[0].permute(1, 2, 0).numpy()
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-4 * 255)
    bgr = rng.integers(0, 256, (9, 11, 3), dtype=np.uint8)
    out = R.v2_preprocess_image(bgr, (11, 9))
    assert out.shape == (1, 9, 11, 3) and np.array_equal(out[0], bgr[:, :, ::-1].astype(np.float32) / np.float32(225.0))   # BGR -> RGB, the 225 typo


def test_space_to_depth_index_map():
    x = np.arange(1 * 4 * 4 * 4, dtype=np.float32).reshape(1, 4, 4, 4)
    y = R.space_to_depth(x, 2)
    for h in range(2):
        for w in range(2):
            for dy in range(2):
                for dx in range(2):
                    for c in range(4):
                        assert y[0, h, w, (dy * 2 + dx) * 4 + c] == x[0, 2 * h + dy, 2 * w + dx, c]
    assert not np.array_equal(R.reorg_darknet(x, 2), y)     # darknet's reorg is a different scramble


def test_tf_nms_hand_cases():
    # boxes [y0,x0,y1,x1]; IoU(0,1)=0.81/1.19>0.5 suppressed; box 2 disjoint; max_output_size caps
    b = np.float32([[0, 0, 1, 1], [0, 0.1, 1, 1.0], [2, 2, 3, 3], [0, 0, 1, 1.02]])
    s = np.float32([0.9, 0.8, 0.7, 0.95])
    assert R.tf_nms(b, s, 10, 0.5).tolist() == [3, 2]
    assert R.tf_nms(b, s, 1, 0.5).tolist() == [3]
    assert R.tf_nms(b, s, 10, 0.99).tolist() == [3, 0, 1, 2]
    # flipped corners are normalised; zero-area boxes never suppress
    assert R.tf_nms(np.float32([[1, 1, 0, 0], [0, 0, 1, 1]]), np.float32([.9, .8]), 10, 0.5).tolist() == [0]
    assert R.tf_nms(np.float32([[0, 0, 0, 1], [0, 0, 0, 1]]), np.float32([.9, .8]), 10, 0.5).tolist() == [0, 1]
    # ties: lower index first (our stated rule)
    assert R.tf_nms(np.float32([[0, 0, 1, 1], [5, 5, 6, 6]]), np.float32([.5, .5]), 10, 0.5).tolist() == [0, 1]
    assert R.tf_nms(np.zeros((0, 4), np.float32), np.zeros(0, np.float32), 10, 0.5).tolist() == []


def test_detection_layer_pixel_vs_ratio_and_grid_order():
    rng = np.random.default_rng(1)
    g, A, C = 4, 3, 2
    raw = rng.standard_normal((2, g, g, A * (5 + C))).astype(np.float32)
    anchors = [(10, 13), (16, 30), (33, 23)]
    pix = R.detection_layer_pixel(raw, anchors, (32 * g, 32 * g))
    rat = R.detection_layer_ratio(raw, anchors, (32 * g, 32 * g))
    assert pix.shape == (2, g * g * A, 5 + C)
    np.testing.assert_allclose(pix[..., :4] / (32 * g), rat[..., :4], rtol=2e-6)
    assert np.array_equal(pix[..., 4:], rat[..., 4:])
    # row = (h*g + w)*A + a ; channel = a*(5+C) + attr ; x offset = column
    h, w, a = 2, 1, 1
    row = (h * g + w) * A + a
    t = raw[0, h, w, a * (5 + C):(a + 1) * (5 + C)]
    sx = 1 / (1 + np.exp(-t[0])); sy = 1 / (1 + np.exp(-t[1]))
    np.testing.assert_allclose(pix[0, row, 0], (sx + w) * 32, rtol=1e-6)
    np.testing.assert_allclose(pix[0, row, 1], (sy + h) * 32, rtol=1e-6)
    np.testing.assert_allclose(pix[0, row, 2], np.exp(t[2]) * 16, rtol=1e-6)
    np.testing.assert_allclose(pix[0, row, 3], np.exp(t[3]) * 30, rtol=1e-6)


def test_select_threshold_strict_and_order_preserving():
    det = np.zeros((4, 7), np.float32)
    det[:, 2:4] = 1
    det[:, 4] = [1.0, 0.5, 1.0, 1.0]
    det[:, 5:] = [[0.5, 0.2], [1.0, 0.3], [0.4, 0.9], [0.5001, 0.1]]
    boxes, scores, classes, idx = R.select_threshold(det, 0.5)
    assert idx.tolist() == [2, 3] and classes.tolist() == [1, 0]     # 0.5 is NOT > 0.5
    assert np.allclose(boxes[0], [-0.5, -0.5, 0.5, 0.5])


def test_v1_decode_layout():
    p = np.zeros((1, 1470), np.float32)
    S, B, C = 7, 2, 20
    cell = 3 * 7 + 4          # row 3, col 4
    p[0, cell * C + 5] = 0.9                          # class 5 prob
    p[0, 980 + cell * B + 1] = 0.8                    # confidence of box 1
    base = 980 + 98 + (cell * B + 1) * 4
    p[0, base:base + 4] = [0.5, 0.25, 0.5, 0.4]
    boxes, scores, labels = R.v1_decode(p, threshold=0.2)
    assert len(scores) == 1 and labels[0] == 5
    np.testing.assert_allclose(scores[0], 0.72, rtol=1e-6)
    np.testing.assert_allclose(boxes[0], [(0.5 + 4) / 7, (0.25 + 3) / 7, 0.25, 0.16], rtol=1e-6)


def test_conv_matches_independent_torch_conv():
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(3)
    for (k, s, cin, cout, h) in ((3, 1, 5, 7, 9), (3, 2, 4, 6, 10), (1, 1, 8, 3, 5), (3, 2, 3, 4, 13)):
        x = rng.standard_normal((2, h, h, cin)).astype(np.float32)
        w = rng.standard_normal((k, k, cin, cout)).astype(np.float32)
        got = R.conv2d_nhwc(x, w, s)
        ref = F.conv2d(torch.from_numpy(x).permute(0, 3, 1, 2), torch.from_numpy(w).permute(3, 2, 0, 1), stride=s, padding=k // 2)
        np.testing.assert_allclose(got, ref.permute(0, 2, 3, 1).numpy(), rtol=1e-4, atol=1e-4)


def test_bf16_rounding_helper():
    x = np.float32([1.0, 1.00390625, 1.005859375, -2.5, 3.1415927, 1e-40])
    r = R.to_bf16(x)
    assert r[0] == 1.0 and r[1] == 1.0 and r[2] == np.float32(1.0078125)    # ties to even, then up
    assert (r.view(np.uint32) & 0xFFFF == 0).all()


def test_fold_bn_equals_unfused():
    rng = np.random.default_rng(5)
    p = dict(beta=rng.normal(size=6).astype(np.float32), gamma=rng.uniform(.5, 2, 6).astype(np.float32),
             mean=rng.normal(size=6).astype(np.float32), var=rng.uniform(.1, 2, 6).astype(np.float32),
             w_hwio=rng.normal(size=(3, 3, 4, 6)).astype(np.float32))
    x = rng.normal(size=(1, 8, 8, 4)).astype(np.float32)
    w, b = R.fold_bn(p)
    np.testing.assert_allclose(R.conv2d_nhwc(x, w) + b, R.batch_norm(R.conv2d_nhwc(x, p["w_hwio"]), p, "tf"), rtol=1e-4, atol=1e-5)


def test_to_fp8_e4m3_grid_ties_and_saturation():
    """The e4m3 rounding used to emulate the device's fp8 storage: every representable value is a fixed point, exact
    midpoints go to the even code, |x| > 448 saturates, the subnormal quantum is 2^-9."""
    vals = []
    for code in range(127):                                   # 0x7f is NaN
        e, m = (code >> 3) & 15, code & 7
        vals.append((m / 8.0) * 2.0 ** -6 if e == 0 else (1 + m / 8.0) * 2.0 ** (e - 7))
    vals = np.array(vals, np.float32)                          # code order == value order
    assert vals.max() == 448.0 and vals[1] == 2.0 ** -9
    assert np.array_equal(R.to_fp8_e4m3(vals), vals) and np.array_equal(R.to_fp8_e4m3(-vals), -vals)
    mid = (vals[:-1] + vals[1:]) / 2
    want = np.where(np.arange(len(mid)) % 2 == 0, vals[:-1], vals[1:])
    assert np.array_equal(R.to_fp8_e4m3(mid), want)
    assert np.array_equal(R.to_fp8_e4m3(np.array([1e9, -500.0, 460.0], np.float32)), np.array([448.0, -448.0, 448.0], np.float32))
    x = np.random.default_rng(0).normal(0, 30, 20000).astype(np.float32)
    q = R.to_fp8_e4m3(x)
    nearest = vals[np.abs(np.abs(x)[:, None].clip(max=448) - vals[None, :]).argmin(1)]
    assert np.all(np.abs(np.abs(q) - np.abs(x).clip(max=448)) <= np.abs(nearest - np.abs(x).clip(max=448)) + 1e-12)


def test_split_f16_pairs_and_their_emulated_network():
    """The oracle's restatement of the device's split-fp16 configuration (no reference counterpart): a pair (hi, lo) of fp16
    numbers carries a float32 value to 2^-22 relative or 2^-25 absolute (the low half's subnormal quantum), saturates where fp16 does, and a
    small network run on pairs stays within 1e-5 of the fp32 forward -- three orders of magnitude closer than fp16 storage."""
    from yolo_tensorflow_amd import darknet_io as IO
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(4096) * 10.0 ** rng.uniform(-3, 3, 4096), [0.0, 65504.0, -70000.0, 131008.0, 1e-7]]).astype(np.float32)
    hi, lo = R.split_f16(x)
    assert np.array_equal(hi, R.to_f16(x)) and np.array_equal(lo, R.to_f16(x - hi))
    err = np.abs((hi + lo) - x)
    ok = np.abs(x) <= 131008
    assert (err[ok] <= np.maximum(np.abs(x[ok]) * 2.0 ** -22, 2.0 ** -25)).all()
    assert hi[-3] == -65504.0 and hi[-2] == 65504.0 and lo[-2] == 65504.0            # beyond fp16's range the high half saturates, the low half carries on
    txt = IO.with_input_size(IO.cfg_text("yolov3-tiny"), 64)
    secs = R.parse_cfg(txt); flat = IO.synth_weights(IO.parse_cfg(txt), seed=2); params = R.unflatten_weights(flat, secs)
    img = rng.random((1, 64, 64, 3), dtype=np.float32)
    h32, _ = R.forward(secs, params, img)
    hx2, outs = R.forward_f16x2(secs, params, img, collect=True)
    h16, _ = R.forward(secs, params, img, storage="f16")
    for (_, a), (_, b), (_, c) in zip(hx2, h32, h16):
        scale = np.abs(b).max()
        assert np.abs(a - b).max() <= 1e-5 * scale and np.abs(c - b).max() > 20 * np.abs(a - b).max()
    assert outs[0].shape == (1, 64, 64, 16)
