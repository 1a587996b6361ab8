"""Parity of what bench.py actually runs: every committed tile plan (yolo_tensorflow_amd/tuned/*.json) at its own size,
batch and dtype against the default plan (bit for bit) and against the fp32 oracle; the bf16 and fp8 headline configurations
over all 32 images of the batch with the measured min IoU / max |dscore| printed and asserted.

Stated tolerances (boxes of every oracle candidate whose score clears the 0.5 threshold by more than the score bound, so that
a legitimate flip across the threshold is not counted):
  bf16 (BASELINE config 3):  IoU >= 0.99,  |dscore| <= 1e-2      (SURVEY.md 8d asks 0.99 for bf16 storage)
  fp8  (BASELINE config 5):  IoU >= FP8_IOU, |dscore| <= FP8_DSCORE over EVERY oracle candidate (no margin), calibrated
                             power-of-two scales; e4m3 keeps 3 mantissa bits
The fp32 device path is held to IoU >= 0.999 in test_gpu_network.py."""
import glob
import json
import os
import re

import numpy as np
import pytest
from oracle import yolo_ref as R
from yolo_tensorflow_amd import darknet_io as IO

pytestmark = pytest.mark.gpu
FP8_IOU, FP8_DSCORE = 0.85, 0.03        # measured on MI355X (4910 candidates): min IoU 0.878, max |dscore| 0.021 -- DESIGN.md section 4
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PLANS = sorted(p for p in glob.glob(os.path.join(ROOT, "yolo_tensorflow_amd", "tuned", "yolov3_*.json")) if re.search(r"_(bf16|fp8)\.json$", p))


def _iou(a, b):
    ix = np.maximum(0, np.minimum(a[:, 2], b[:, 2]) - np.maximum(a[:, 0], b[:, 0])); iy = np.maximum(0, np.minimum(a[:, 3], b[:, 3]) - np.maximum(a[:, 1], b[:, 1]))
    inter = ix * iy
    return inter / ((a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]) + (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]) - inter + 1e-12)


def box_deviation(ref, det, margin, thr=0.5):
    """ref, det: [n, rows, 85] decoded tensors.  Over every reference candidate with score >= thr + margin:
    (min IoU, max |dscore|, number of candidates, number whose device score fell below the threshold)."""
    miou, mds, cnt, lost = 1.0, 0.0, 0, 0
    for b in range(ref.shape[0]):
        rb, rs, rc, ridx = R.select_threshold(ref[b], thr)
        ok = rs >= thr + margin
        if not ok.any():
            continue
        rows = ridx[ok]
        d = det[b][rows]
        sc = (d[:, 4:5] * d[:, 5:]).max(-1)
        bx = np.stack([d[:, 0] - d[:, 2] / 2, d[:, 1] - d[:, 3] / 2, d[:, 0] + d[:, 2] / 2, d[:, 1] + d[:, 3] / 2], -1)
        miou = min(miou, float(_iou(rb[ok], bx).min())); mds = max(mds, float(np.abs(rs[ok] - sc).max()))
        cnt += int(ok.sum()); lost += int((sc <= thr).sum())
    return miou, mds, cnt, lost


def _setup(size, batch, seed=0):
    txt = IO.cfg_text("yolov3") if size == 416 else IO.with_input_size(IO.cfg_text("yolov3"), size)
    secs = IO.parse_cfg(txt); flat = IO.synth_weights(secs, seed=seed)
    img = np.random.default_rng(1).integers(0, 256, (batch, size, size, 3), dtype=np.uint8)
    return txt, flat, img


_ORACLE = {}


def oracle_detections(txt, flat, img, size):
    """fp32 oracle (TF semantics) decoded tensor for `img`, image by image (bounded memory), cached per (size, batch)."""
    key = (size, img.shape[0])
    if key not in _ORACLE:
        osecs = R.parse_cfg(txt); params = R.unflatten_weights(flat, osecs)
        out = []
        for b in range(img.shape[0]):
            heads, _ = R.forward(osecs, params, img[b:b + 1].astype(np.float32) / np.float32(255))
            out.append(R.yolo_v3_detections(heads, size, ratio=True)[0])
        _ORACLE[key] = np.stack(out)
    return _ORACLE[key]


def test_committed_plans_exist():
    names = [os.path.basename(p) for p in PLANS]
    for want in ("yolov3_416_b32_bf16.json", "yolov3_416_b32_fp8.json", "yolov3_608_b8_bf16.json", "yolov3_608_b8_fp8.json"):
        assert want in names


@pytest.mark.parametrize("plan_path", PLANS, ids=[os.path.basename(p)[:-5] for p in PLANS])
def test_tuned_plan_equals_default_plan_and_tracks_oracle(hiplib, plan_path):
    """The plan bench.py loads for this (size, batch, dtype): decoded tensor and NMS output of the FULL batch bit-identical to
    the default plan's (every tile shape walks K in the same order), through the eager path and through the captured graph;
    one image of it against the fp32 oracle."""
    import torch
    m = re.match(r"yolov3_(\d+)_b(\d+)_(bf16|fp8)\.json", os.path.basename(plan_path))
    size, batch, dt = int(m.group(1)), int(m.group(2)), m.group(3)
    plan = json.load(open(plan_path))
    assert plan["num_cfgs"] == hiplib.op_conv_num_cfgs(), "plan was tuned against a different tile table"
    txt, flat, img = _setup(size, batch)
    eng = hiplib.Engine(txt, max_batch=batch, dtype=hiplib.BF16 if dt == "bf16" else hiplib.FP8)
    eng.set_weights(flat)
    det_default = eng.forward(img)
    box_default = eng.postprocess(batch, score_thr=0.5, iou_thr=0.5, max_out=20)
    eng.set_tile_configs(plan["cfgs"])
    assert np.array_equal(eng.get_tile_configs(), np.asarray(plan["cfgs"], np.int32))
    det_tuned = eng.forward(img)
    assert np.array_equal(det_tuned, det_default), "tuned plan %s changes the decoded tensor" % os.path.basename(plan_path)
    box_tuned = eng.postprocess(batch, score_thr=0.5, iou_thr=0.5, max_out=20)
    for b in range(batch):
        assert np.array_equal(box_tuned[b], box_default[b])
    # the graph replay bench.py times
    d_img = torch.from_numpy(img).cuda()
    boxes = torch.zeros((batch, 20 * 6), dtype=torch.int32, device="cuda"); counts = torch.zeros((batch,), dtype=torch.int32, device="cuda")
    for _ in range(3):
        eng.detect_graph(d_img, boxes, counts, score_thr=0.5, iou_thr=0.5, max_out=20)
        eng.synchronize()
    got = boxes.cpu().numpy().view(hiplib.BOX_DTYPE).reshape(batch, 20); gc = counts.cpu().numpy()
    for b in range(batch):
        assert gc[b] == len(box_default[b]) and np.array_equal(got[b, :gc[b]], box_default[b])
    eng.close()
    # one image against the fp32 oracle (same bound as the all-image tests below)
    ref = oracle_detections(txt, flat, img[:1], size)
    if dt == "bf16":
        miou, mds, cnt, lost = box_deviation(ref, det_tuned[:1], 1e-2)
        print("%s image 0 vs fp32 oracle: %d candidates, min IoU %.4f, max |dscore| %.4f" % (os.path.basename(plan_path), cnt, miou, mds))
        assert cnt > 3 and lost == 0 and miou >= 0.99 and mds <= 1e-2


def test_tuned_tile_configs_of_the_b32_plan_as_operators_at_batch_32(hiplib):
    """every distinct (layer class, tile configuration) the committed 416 b32 bf16 plan uses on a tunable layer, run as
    the stand-alone operator (`yolo_op_conv2d`, tile_cfg forced) at the PRODUCTION size -- batch 32, the layer's real spatial extent and
    channel counts, with the shortcut where the network folds one -- against the oracle's conv on the same bf16-rounded operands (one bf16
    ulp).  The network tests cover these configurations only through whole-network bit-identity with the default plan."""
    from test_gpu_ops import _assert_bf16_close
    txt = IO.cfg_text("yolov3"); secs = IO.parse_cfg(txt); L = secs[1:]; shapes = IO.layer_shapes(secs)
    plan = json.load(open(os.path.join(ROOT, "yolo_tensorflow_amd", "tuned", "yolov3_416_b32_bf16.json")))["cfgs"]
    seen = {}
    for i, s in enumerate(L):
        if s["type"] != "convolutional" or plan[i] < 0 or plan[i] == 1000:
            continue
        k, st, cin, cout = int(s["size"]), int(s.get("stride", 1)), shapes[i][4], int(s["filters"])
        h = shapes[i][1] * st                    # input extent (layer_shapes gives the output's)
        res = i + 1 < len(L) and L[i + 1]["type"] == "shortcut"
        if h > 52 or cout == 255:            # (the 104 / 208 / 416 layers run fixed fused kernels in this plan; the fp32 heads have their own test)
            continue
        seen.setdefault((k, st, h, cin, cout, res, plan[i] % 10000), i)
    assert len(seen) >= 10, seen
    rng = np.random.default_rng(77)
    for (k, st, h, cin, cout, res, cfg), layer in sorted(seen.items()):
        x = R.to_bf16(rng.standard_normal((32, h, h, cin)).astype(np.float32))
        w = R.to_bf16((rng.standard_normal((k, k, cin, cout)) * np.sqrt(2.0 / (k * k * cin))).astype(np.float32))
        b = rng.standard_normal(cout).astype(np.float32)
        ref = R.leaky_relu(R.conv2d_nhwc(x, w, st) + b)
        r = R.to_bf16(rng.standard_normal(ref.shape).astype(np.float32)) if res else None
        got = hiplib.op_conv2d(x, w, b, stride=st, act=1, residual=r, tile_cfg=cfg)
        if res:
            # conv output rounded to bf16, then the sum rounded again: one ulp of the LARGER magnitude -- and, at 11 M elements, the rare
            # element whose conv output sits on a rounding boundary (summation order decides) and whose sum then rounds the other way too: two
            want = R.to_bf16(R.to_bf16(ref) + r); scale = np.abs(ref) + np.abs(r)
            err = np.abs(got - want)
            assert (err <= 2.0 ** -6 * scale + 2e-3).all() and (err > 2.0 ** -7 * scale + 2e-3).mean() < 1e-5, float(err.max())
        else:
            _assert_bf16_close(got, ref)
        print("cfg layer %3d: %dx%d/%d %3d^2 %4d -> %4d%s at batch 32, tile configuration %d: within one bf16 ulp of the oracle" % (layer, k, k, st, h, cin, cout, " + shortcut" if res else "", cfg))


def test_tolerance_line_fp16x2_416_b32_all_images_vs_oracle(hiplib):
    """What bench.py prints as `tolerance_line`: split-fp16 pairs with the committed plan (tuned/yolov3_416_b32_fp16x2.json), every one of the
    32 images of the timed batch against the fp32 oracle at north_star's tolerance, IoU >= 0.999 (and |dscore| <= 1e-3, nothing lost); the
    tuned plan bit-identical to the built-in one.  The three weight flavours on natural images: test_gpu_fp16x2.py."""
    txt, flat, img = _setup(416, 32)
    plan = json.load(open(os.path.join(ROOT, "yolo_tensorflow_amd", "tuned", "yolov3_416_b32_fp16x2.json")))
    assert plan["num_cfgs"] == hiplib.op_conv_num_cfgs()
    eng = hiplib.Engine(txt, max_batch=32, dtype=hiplib.FP16X2)
    eng.set_weights(flat)
    det0 = eng.forward(img)
    eng.set_tile_configs(plan["cfgs"])
    det = eng.forward(img)
    eng.close()
    assert np.array_equal(det, det0)
    ref = oracle_detections(txt, flat, img, 416)
    miou, mds, cnt, lost = box_deviation(ref, det, 1e-3)
    print("tolerance line (fp16x2 416 b32) vs fp32 oracle: %d candidates over 32 images, min IoU %.5f, max |dscore| %.6f, lost %d -> meets north_star 0.999: %s"
          % (cnt, miou, mds, lost, "yes" if miou >= 0.999 and lost == 0 else "NO"))
    assert cnt > 100 and lost == 0 and miou >= 0.999 and mds <= 1e-3


def test_config3_bf16_416_b32_all_images_vs_oracle(hiplib):
    """BASELINE config 3 with the committed plan, every one of the 32 images against the fp32 oracle."""
    txt, flat, img = _setup(416, 32)
    eng = hiplib.Engine(txt, max_batch=32)
    eng.set_weights(flat)
    eng.set_tile_configs(json.load(open(os.path.join(ROOT, "yolo_tensorflow_amd", "tuned", "yolov3_416_b32_bf16.json")))["cfgs"])
    det = eng.forward(img)
    eng.close()
    ref = oracle_detections(txt, flat, img, 416)
    miou, mds, cnt, lost = box_deviation(ref, det, 1e-2)
    print("bf16 416 b32 vs fp32 oracle: %d candidates over 32 images, min IoU %.4f, max |dscore| %.5f, lost %d -> meets north_star 0.999: %s (asserted: the bf16 "
          "regression guard 0.99 -- an 8-bit significand cannot hold 0.999, DESIGN.md section 4; the configuration that does: tolerance line above)" % (cnt, miou, mds, lost, "yes" if miou >= 0.999 else "no"))
    assert cnt > 100 and lost == 0
    assert miou >= 0.99 and mds <= 1e-2
    # whole decoded tensor, relative to its largest value (box coordinates are O(1) ratios)
    assert float(np.abs(det - ref).max() / np.abs(ref).max()) < 2e-2


def test_config5_fp8_416_b32_full_size(hiplib):
    """BASELINE config 5 at its own size: YOLOv3 416x416, batch 32, e4m3 filters and activations with the committed tile plan.
    Properties (determinism, batch independence, ranges, NMS tail == oracle tail on the device's decoded tensor) for the unit
    scales bench.py uses, and boxes against the fp32 oracle with calibrated power-of-two scales (the accuracy statement that
    goes with the fp8 bench line)."""
    txt, flat, img = _setup(416, 32)
    plan = json.load(open(os.path.join(ROOT, "yolo_tensorflow_amd", "tuned", "yolov3_416_b32_fp8.json")))["cfgs"]
    eng = hiplib.Engine(txt, max_batch=32, dtype=hiplib.FP8)
    eng.set_weights(flat); eng.set_tile_configs(plan)
    det = eng.forward(img)
    assert det.shape == (32, 10647, 85) and np.isfinite(det).all()
    assert np.array_equal(det, eng.forward(img))
    for i in (0, 13, 31):
        assert np.array_equal(eng.forward(img[i:i + 1])[0], det[i])
    assert (det[..., 4:] >= 0).all() and (det[..., 4:] <= 1).all() and (det[..., 2:4] > 0).all()
    eng.forward(img, want_detections=False)
    res = eng.postprocess(32, score_thr=0.5, iou_thr=0.5, max_out=20)
    for b in range(32):
        ob, os_, oc = R.detect_v3_tf(det[b], 0.5, 0.5, 20)
        assert np.array_equal(res[b]["score"], os_) and np.array_equal(res[b]["cls"], oc)
    ref = oracle_detections(txt, flat, img, 416)
    u = box_deviation(ref, det, 0.0)
    print("fp8 416 b32, unit scales, vs fp32 oracle: %d candidates, min IoU %.4f, max |dscore| %.4f, below threshold %d" % (u[2], u[0], u[1], u[3]))
    eng.close()
    # calibrated scales from one image's fp32 per-layer maxima (oracle.fp8_calibrate_scales)
    osecs = R.parse_cfg(txt); params = R.unflatten_weights(flat, osecs)
    _, outs32 = R.forward(osecs, params, img[:1].astype(np.float32) / np.float32(255), collect=True)
    eng = hiplib.Engine(txt, max_batch=32, dtype=hiplib.FP8)
    eng.set_act_scales(R.fp8_calibrate_scales(osecs, outs32))
    eng.set_weights(flat); eng.set_tile_configs(plan)
    detc = eng.forward(img)
    eng.close()
    # every oracle candidate (score > 0.5, no margin: a candidate close to the threshold may legitimately fall below it)
    miou, mds, cnt, lost = box_deviation(ref, detc, 0.0)
    print("fp8 416 b32, calibrated scales, vs fp32 oracle: %d candidates, min IoU %.4f, max |dscore| %.4f, below threshold %d" % (cnt, miou, mds, lost))
    assert cnt > 1000
    assert miou >= FP8_IOU and mds <= FP8_DSCORE


MIXED_IOU, MIXED_DSCORE = 0.97, 5e-3


def test_config5_mixed_plan_416_b32_vs_oracle(hiplib):
    """a USABLE accuracy point for the e4m3 configuration.  The committed mixed plan (tuned/yolov3_416_b32_mixed.json:
    the 13x13 stage and the FPN blocks in bf16, the backbone up to 26x26 in e4m3) on all 32 images against the fp32 oracle, every oracle
    candidate, no margin; the plan's tile choices are bit-identical to the default ones."""
    txt0, flat, img = _setup(416, 32)
    mp = json.load(open(os.path.join(ROOT, "yolo_tensorflow_amd", "tuned", "yolov3_416_b32_mixed.json")))
    secs0 = IO.parse_cfg(txt0)
    assert IO.store_closure(secs0, mp["store_bf16"]) == sorted(mp["store_bf16"])              # the committed list is closed
    txt = IO.with_layer_store(txt0, mp["store_bf16"])
    share = IO.bf16_flop_share(IO.parse_cfg(txt))
    eng = hiplib.Engine(txt, max_batch=32, dtype=hiplib.FP8)
    eng.set_weights(flat)
    det = eng.forward(img)
    if "cfgs" in mp:
        assert mp["num_cfgs"] == hiplib.op_conv_num_cfgs()
        eng.set_tile_configs(mp["cfgs"])
        assert np.array_equal(eng.forward(img), det)
    eng.close()
    ref = oracle_detections(txt0, flat, img, 416)
    miou, mds, cnt, lost = box_deviation(ref, det, 0.0)
    print("mixed e4m3 / bf16 plan (%.1f %% of the FLOPs on the bf16 MFMA), 416 b32, vs fp32 oracle: %d candidates, min IoU %.4f, max |dscore| %.4f, below threshold %d"
          % (100 * share, cnt, miou, mds, lost))
    assert cnt > 1000 and miou >= MIXED_IOU and mds <= MIXED_DSCORE


def test_graph_survives_box_buffer_growth(hiplib):
    """A captured detect graph holds the box buffer's address; a later postprocess with a larger max_out reallocates that
    buffer, so the graph must be dropped and re-captured instead of replaying writes into freed memory."""
    import torch
    txt = IO.with_input_size(IO.cfg_text("yolov3-tiny"), 160)
    secs = IO.parse_cfg(txt); flat = IO.synth_weights(secs, seed=1, obj_bias=1.0)
    img = np.random.default_rng(4).integers(0, 256, (2, 160, 160, 3), dtype=np.uint8)
    eng = hiplib.Engine(txt, max_batch=2)
    eng.set_weights(flat)
    d_img = torch.from_numpy(img).cuda()
    boxes = torch.zeros((2, 10 * 6), dtype=torch.int32, device="cuda"); counts = torch.zeros((2,), dtype=torch.int32, device="cuda")
    eng.forward(img, want_detections=False)
    want = eng.postprocess(2, score_thr=0.3, iou_thr=0.5, max_out=10)

    def run_graph():
        boxes.zero_(); counts.zero_(); torch.cuda.synchronize()
        eng.detect_graph(d_img, boxes, counts, score_thr=0.3, iou_thr=0.5, max_out=10)
        eng.synchronize()
        got = boxes.cpu().numpy().view(hiplib.BOX_DTYPE).reshape(2, 10); gc = counts.cpu().numpy()
        for b in range(2):
            assert gc[b] == len(want[b]) and np.array_equal(got[b, :gc[b]], want[b])

    for _ in range(3):
        run_graph()                                   # eager, capture, replay
    big = eng.postprocess(2, score_thr=0.3, iou_thr=0.5, max_out=200)      # grows (reallocates) the box buffer
    assert all(len(b) >= len(w) for b, w in zip(big, want))
    for _ in range(3):
        run_graph()                                   # must not replay against the freed buffer
    eng.close()


@pytest.mark.parametrize("dtype_name", ["bf16", "fp8", "fp16"])
def test_autotune_runs_and_keeps_results(hiplib, dtype_name):
    """yolo_autotune itself (bench.py --retune): every tile configuration the dtype's table instantiates is probed in situ, including the
    fused-1x1-tail pass -- round 4 regression: a tail-capable shape that the e4m3 table does not instantiate must not be proposed there --
    and whatever plan comes out gives the default plan's decoded tensor bit for bit (416 x 416 input, four images: 13-multiples on every stage, so the
    halo forms and the tails take part)."""
    dt = {"bf16": hiplib.BF16, "fp8": hiplib.FP8, "fp16": hiplib.FP16}[dtype_name]
    txt = IO.with_input_size(IO.cfg_text("yolov3"), 416)
    secs = IO.parse_cfg(txt); flat = IO.synth_weights(secs, seed=0)
    img = np.random.default_rng(3).integers(0, 256, (4, 416, 416, 3), dtype=np.uint8)
    eng = hiplib.Engine(txt, max_batch=4, dtype=dt)
    eng.set_weights(flat)
    want = eng.forward(img)
    eng.autotune(4, 1)
    plan = eng.get_tile_configs()
    num = hiplib.op_conv_num_cfgs()
    assert all(v == 1000 or v % 10000 < num for v in plan[plan >= 0].tolist()), plan          # 1000 = CONV_CFG_DIRECT (first layer, stem off)
    assert np.array_equal(eng.forward(img), want)
    eng.close()


def test_fused_tail_on_ragged_halo_blocks_608(hiplib):
    """the fused 1x1 tail on the halo-staged 3x3 forms is admitted on ragged 13 x 13 blocks (608 x 608: the 76 x 76 stage is
    6 x 13 - 2) by conv_halo13_ok / set_tile_configs / autotune -- force it, with every halo configuration that can carry a tail (the
    176 x 256 ones: set_tile_configs refuses the others), on every 3x3 layer that has one to carry, and compare with the plan that launches
    the 1x1 convs themselves: bit for bit (the tail walks K in the stand-alone kernel's order)."""
    txt = IO.with_input_size(IO.cfg_text("yolov3"), 608)
    secs = IO.parse_cfg(txt); flat = IO.synth_weights(secs, seed=0)
    img = np.random.default_rng(4).integers(0, 256, (2, 608, 608, 3), dtype=np.uint8)
    eng = hiplib.Engine(txt, max_batch=2)
    eng.set_weights(flat)
    base = eng.get_tile_configs().copy()
    plain = np.where(base >= 10000, base - 10000, base)                 # no tails anywhere
    eng.set_tile_configs(plain)
    want = eng.forward(img)
    shapes = IO.layer_shapes(secs)
    forced = {}
    for cfg in range(36, 44):                                            # the halo-staged configurations (csrc/conv_halo13.hip)
        for i, s in enumerate(secs[1:]):
            if s["type"] == "convolutional" and int(s["size"]) == 3 and int(s.get("stride", 1)) == 1 and shapes[i][1] in (38, 76) and int(s["filters"]) <= 256:
                trial = plain.copy(); trial[i] = 10000 + cfg
                try:
                    eng.set_tile_configs(trial)
                except hiplib.YoloError:
                    continue                                             # this configuration / layer cannot carry a tail
                got = eng.forward(img)
                assert np.array_equal(got, want), "layer %d with tile config %d + tail differs from the separate launches" % (i, cfg)
                forced[cfg] = forced.get(cfg, 0) + 1
    eng.close()
    print("fused tail on ragged blocks, layers per halo configuration:", forced)
    assert sum(forced.values()) >= 4 and len(forced) >= 1, forced
