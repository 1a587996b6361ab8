"""YOLO_FP16: IEEE fp16 filters and activations on v_mfma_f32_16x16x32_f16 -- the bf16 configuration's kernels, tile table and plans
with an 11-bit significand instead of 8.  Operator parity against the oracle on fp16-exact operands (one fp16 ulp), every tile shape
bit-identical, the fused forms (stem, halo, 1x1 tails, shortcut) bit-identical to the layer-by-layer plan, whole networks against the
oracle's fp16-storage emulation, and boxes against the fp32 oracle -- on the reference's images, for benign and trained-file
batch-norm statistics (the numbers DESIGN.md section 4 quotes)."""
import json
import os

import numpy as np
import pytest
from oracle import yolo_ref as R
from yolo_tensorflow_amd import darknet_io as IO
from test_gpu_ops import V3_SHAPES, HALO_CFGS
from test_gpu_tuned import box_deviation

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _assert_f16_close(got, ref, scale=None):
    # operands are fp16-exact, accumulation fp32: differences are summation order and the final rounding(s) -> an fp16 ulp or two
    # plus slack for cancellation near zero
    want = R.to_f16(ref)
    err = np.abs(got - want)
    tol = 2.0 ** -9 * (np.abs(want) if scale is None else scale) + 3e-4          # fp16 spacing is 2^-10 of the binade: two ulps
    assert (err <= tol).all(), "max err %.3e at %s" % (err.max(), np.unravel_index(err.argmax(), err.shape))


def _case(rng, k, s, h, cin, cout, n=2):
    h = min(h, 26 if cin * cout < 65536 else 13)
    if s == 2:
        h += h % 2
    x = R.to_f16(rng.standard_normal((n, h, h, cin)).astype(np.float32))
    w = R.to_f16((rng.standard_normal((k, k, cin, cout)) * np.sqrt(2.0 / (k * k * cin))).astype(np.float32))
    b = rng.standard_normal(cout).astype(np.float32)
    return x, w, b, R.leaky_relu(R.conv2d_nhwc(x, w, s) + b)


@pytest.mark.parametrize("shape", V3_SHAPES, ids=lambda s: "k%d_s%d_h%d_%dto%d" % s)
def test_conv_fp16_all_yolov3_shapes(hiplib, shape):
    k, s, h, cin, cout = shape
    x, w, b, ref = _case(np.random.default_rng(hash(shape) % 2 ** 32), k, s, h, cin, cout)
    got = hiplib.op_conv2d(x, w, b, stride=s, act=1, dtype=hiplib.FP16)
    assert got.shape == ref.shape
    _assert_f16_close(got, ref)


@pytest.mark.parametrize("shape", [(3, 1, 26, 64, 128), (1, 1, 13, 1024, 255), (3, 2, 26, 32, 64), (3, 1, 20, 3, 32)], ids=lambda s: "k%d_s%d_h%d_%dto%d" % s)
def test_conv_fp16_every_tile_config_agrees(hiplib, shape):
    k, s, h, cin, cout = shape
    x, w, b, ref = _case(np.random.default_rng(7), k, s, h, cin, cout, n=3)
    outs = []
    for c in range(hiplib.op_conv_num_cfgs()):
        try:
            outs.append(hiplib.op_conv2d(x, w, b, stride=s, act=1, tile_cfg=c, dtype=hiplib.FP16))
        except hiplib.YoloError as e:
            assert "not applicable" in str(e)
    assert len(outs) >= 36
    _assert_f16_close(outs[0], ref)
    for o in outs[1:]:
        assert np.array_equal(o, outs[0])


@pytest.mark.parametrize("case", [(2, 13, 512, 1024, False), (2, 26, 256, 512, True), (1, 52, 128, 256, True), (1, 104, 64, 128, True), (1, 26, 192, 200, True)],
                         ids=lambda c: "n%d_h%d_%dto%d_res%d" % c)
def test_conv_fp16_halo_forms_and_shortcut(hiplib, case):
    n, h, cin, cout, residual = case
    rng = np.random.default_rng(n * 1000 + h + cin)
    x = R.to_f16(rng.standard_normal((n, h, h, cin)).astype(np.float32))
    w = R.to_f16((rng.standard_normal((3, 3, cin, cout)) * np.sqrt(2.0 / (9 * cin))).astype(np.float32))
    b = rng.standard_normal(cout).astype(np.float32)
    res = R.to_f16(rng.standard_normal((n, h, h, cout)).astype(np.float32)) if residual else None
    tiled = hiplib.op_conv2d(x, w, b, act=1, residual=res, dtype=hiplib.FP16, tile_cfg=16)
    ref = R.leaky_relu(R.conv2d_nhwc(x, w, 1) + b)
    if residual:
        _assert_f16_close(tiled, R.to_f16(ref) + res, scale=np.abs(ref) + np.abs(res))
    else:
        _assert_f16_close(tiled, ref)
    for cfg in HALO_CFGS:
        assert np.array_equal(hiplib.op_conv2d(x, w, b, act=1, residual=res, dtype=hiplib.FP16, tile_cfg=cfg), tiled), "halo cfg %d differs" % cfg


def test_conv_fp16_saturates_instead_of_overflowing(hiplib):
    x = np.full((1, 13, 13, 64), 60000.0, np.float32); x = R.to_f16(x)
    w = np.zeros((1, 1, 64, 64), np.float32); w[0, 0, 0, :] = 2.0
    got = hiplib.op_conv2d(x, w, None, act=0, dtype=hiplib.FP16)
    assert np.isfinite(got).all() and (got == 65504.0).all()


@pytest.mark.parametrize("name,size", [("yolov3", 96), ("yolov3-tiny", 160), ("yolov2", 160)])
def test_fp16_network_vs_oracle_fp16_storage(hiplib, name, size):
    txt = IO.with_input_size(IO.cfg_text(name), size)
    secs = IO.parse_cfg(txt); flat = IO.synth_weights(secs, seed=4)
    img = np.random.default_rng(9).integers(0, 256, (2, size, size, 3), dtype=np.uint8)
    osecs = R.parse_cfg(txt); params = R.unflatten_weights(flat, osecs)
    x01 = img.astype(np.float32) / np.float32(255)
    eng = hiplib.Engine(txt, max_batch=2, dtype=hiplib.FP16, keep_layers=True)
    eng.set_weights(flat)
    det = eng.forward(img)
    _, outs = R.forward(osecs, params, R.to_f16(x01), storage="f16", collect=True)
    worst = 0.0
    for i, o in enumerate(outs):
        if o is None or osecs[i + 1]["type"] in ("route",):
            continue
        got = eng.layer_output(i, 2)
        e = float(np.abs(got - o).max() / max(np.abs(o).max(), 1e-6)); worst = max(worst, e)
        assert e < 4e-3, "layer %d (%s): rel err %.2e" % (i, osecs[i + 1]["type"], e)
    print("%s fp16 vs the oracle's fp16-storage emulation: worst layer rel err %.2e" % (name, worst))
    eng.close()
    # fused plan (stem, halo, tails, shortcuts in the epilogue) == layer-by-layer plan, bit for bit
    eng2 = hiplib.Engine(txt, max_batch=2, dtype=hiplib.FP16)
    eng2.set_weights(flat)
    assert np.array_equal(eng2.forward(img), det)
    eng2.close()


def test_fp16_tuned_plan_and_boxes_416_b32(hiplib):
    """The fp16 bench configuration: the committed bf16 tile plan (same kernels shape for shape) on fp16 storage, batch 32 at 416 x 416:
    bit-identical to the default plan, and every one of the 32 images against the fp32 oracle."""
    from test_gpu_tuned import _setup, oracle_detections
    txt, flat, img = _setup(416, 32)
    plan = json.load(open(os.path.join(ROOT, "yolo_tensorflow_amd", "tuned", "yolov3_416_b32_bf16.json")))["cfgs"]
    eng = hiplib.Engine(txt, max_batch=32, dtype=hiplib.FP16)
    eng.set_weights(flat)
    d0 = eng.forward(img)
    eng.set_tile_configs(plan)
    det = eng.forward(img)
    assert np.array_equal(det, d0)
    eng.close()
    ref = oracle_detections(txt, flat, img, 416)
    miou, mds, cnt, lost = box_deviation(ref, det, 1e-3)
    print("fp16 416 b32 vs fp32 oracle: %d candidates over 32 images, min IoU %.4f, max |dscore| %.5f, lost %d" % (cnt, miou, mds, lost))
    assert cnt > 100 and lost == 0 and miou >= 0.9985 and mds <= 1e-3        # measured 0.9990 / 0.00015: AT north_star's 0.999, no margin (DESIGN.md section 4 says so)


@pytest.mark.parametrize("stats", ["benign", "log", "real"])
def test_fp16_reference_images(hiplib, stats):
    """The reference's six jpgs through the detector entry point in fp16 storage, against the fp32 oracle: benign statistics and
    trained-file statistics (where bf16 storage drops to IoU ~0.65, tests/test_gpu_natural.py)."""
    from yolo_tensorflow_amd import detector
    import test_gpu_natural as N
    txt, flat = N._weights(stats)
    ref = N._oracle(stats)
    d = detector.YOLOV3(None, weights=flat, dtype=hiplib.FP16)
    det = np.stack([d.engine.forward_image(N._load(p))[0] for p in N.IMAGES])
    miou, mds, cnt, lost = box_deviation(ref, det, 1e-2, thr=d.threshold)
    print("natural images, %s weights, fp16: %d candidates over 6 images, min IoU %.4f, max |dscore| %.5f, below threshold %d" % (stats, cnt, miou, mds, lost))
    d.engine.close()
    # measured: benign 0.9988 / 0.0002 -- just UNDER north_star's 0.999, and said so there; log (drawn vectors) 0.9619 /
    # 0.023; real (the reference's own vectors, round 5): the five ordinary jpgs and person.jpg apart (test_gpu_natural.real_split).
    # Guards sit just under the measured values.
    if stats == "real":
        (miou, mds, cnt, lost), hard = N.real_split(ref, det, 1e-2, d.threshold)
        print("   ... the five ordinary jpgs: %d candidates, min IoU %.4f, max |dscore| %.5f, lost %d;  person.jpg: %d candidates, min IoU %.4f, max |dscore| %.4f, lost %d"
              % (cnt, miou, mds, lost, hard[2], hard[0], hard[1], hard[3]))
        assert hard[0] >= 0.30 and hard[1] <= 0.12 and hard[3] <= 12        # measured 0.336 / 0.093 / 7 (the oracle's fp16 emulation: 0.42 / 0.147 / 9)
    lo, hi = {"benign": (0.9985, 1e-3), "log": (0.95, 3e-2), "real": (0.994, 2e-3)}[stats]        # (real, the five ordinary jpgs: measured 0.9955 / 0.0007)
    assert lost == 0 and miou >= lo and mds <= hi
