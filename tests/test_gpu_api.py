"""The Python entry points that keep the reference's names/signatures, run on the device, against golden vectors
captured from the reference's own numpy functions and against the oracle."""
import numpy as np
import pytest
from conftest import golden
from oracle import yolo_ref as R
from yolo_tensorflow_amd import darknet_io as IO

pytestmark = pytest.mark.gpu


def test_non_max_suppression_equals_reference_numpy(hiplib):
    """`non_max_suppression` (V3/yolo_v3.py:376-420) on the device == the reference's own output, bit for bit,
    including its shifted-score behaviour (golden captured by importing the reference)."""
    from yolo_tensorflow_amd import yolo_v3
    g = golden("nms_v3_numpy.npz")
    res = yolo_v3.non_max_suppression(g["det"], float(g["conf"]), float(g["iou"]))
    keys = sorted(res.keys())
    assert keys == list(g["classes"])
    assert [len(res[k]) for k in keys] == list(g["counts"])
    boxes = np.concatenate([np.array([b for b, _ in res[k]], dtype=np.float32) for k in keys])
    scores = np.concatenate([np.array([s for _, s in res[k]], dtype=np.float32) for k in keys])
    assert np.array_equal(boxes, g["boxes"])
    assert np.array_equal(scores, g["scores"])


def test_non_max_suppression_vs_oracle_random(hiplib):
    from yolo_tensorflow_amd import yolo_v3
    rng = np.random.default_rng(31)
    det = rng.uniform(0, 1, (2, 500, 5 + 12)).astype(np.float32)
    det[..., 2:4] = det[..., 0:2] + rng.uniform(0.05, 0.4, (2, 500, 2)).astype(np.float32)
    det[..., 4] = rng.uniform(0, 1, (2, 500)) ** 3
    want = R.np_nms_v3(det, 0.3, 0.4)
    got = yolo_v3.non_max_suppression(det, 0.3, 0.4)
    assert sorted(got) == sorted(want)
    for k in want:
        assert len(got[k]) == len(want[k])
        for (gb, gs), (wb, ws) in zip(got[k], want[k]):
            assert np.array_equal(gb, wb) and gs == ws
    assert yolo_v3.non_max_suppression(det, 2.0, 0.4) == {}


def test_v2_postprocess_equals_reference_numpy(hiplib):
    """V2 `postprocess` (V2/utils.py:30-62) on the device == the reference's own output (golden), bit for bit."""
    from yolo_tensorflow_amd import yolo_v2
    g = golden("v2_postprocess.npz")
    b, s, c = yolo_v2.postprocess(g["bboxes"], g["obj"], g["cls"], image_shape=tuple(int(v) for v in g["image_shape"]), threshold=float(g["threshold"]))
    assert np.array_equal(b, g["out_boxes"]) and np.array_equal(s, g["out_scores"]) and np.array_equal(c, g["out_classes"])


def test_yolo_v3_entry_points(hiplib, tmp_path):
    from yolo_tensorflow_amd import yolo_v3
    size = 96
    txt = IO.with_input_size(IO.cfg_text("yolov3"), size)
    secs = IO.parse_cfg(txt); flat = IO.synth_weights(secs, 0)
    wf = str(tmp_path / "yolov3.weights"); IO.write_weights_file(wf, flat, 0, 2)
    yolo_v3.load_weights(None, wf, size=size, max_batch=2, dtype=hiplib.FP32)
    rng = np.random.default_rng(3)
    inputs = rng.integers(0, 256, (2, size, size, 3)).astype(np.float32)       # the reference feeds float32 0..255
    det = yolo_v3.yolo_v3(inputs, 80, data_format='NHWC')
    osecs = R.parse_cfg(txt); params = R.unflatten_weights(flat, osecs)
    heads, _ = R.forward(osecs, params, inputs / np.float32(255))
    ref = R.yolo_v3_detections(heads, size, ratio=False)                      # pixel units (V3/yolo_v3.py:111-159)
    assert det.shape == ref.shape == (2, 9 * 3 + 36 * 3 + 144 * 3, 85)
    np.testing.assert_allclose(det, ref, rtol=2e-3, atol=2e-3)
    boxes = yolo_v3.detections_boxes(det)
    assert np.array_equal(boxes, R.detections_boxes(det))
    # 4-output variant: normalised detections + per-image TF NMS
    d2, bb, ss, cc = yolo_v3.yolo_v3_with_nms(inputs, 80, score_threshold=0.3, iou_threshold=0.5, data_format='NHWC')
    np.testing.assert_allclose(d2[..., :4] * size, det[..., :4], rtol=1e-4, atol=1e-3)
    for b in range(2):
        wb, ws, wc = R.detect_v3_tf(d2[b], 0.3, 0.5, 20)
        assert np.array_equal(bb[b], wb) and np.array_equal(ss[b], ws) and np.array_equal(cc[b], wc)
    with pytest.raises(hiplib.YoloError):
        yolo_v3.yolo_v3(np.zeros((1, 64, 64, 3), np.float32), 80)            # nothing bound at that size
    for m in list(yolo_v3._default.values()):
        m.close()
    yolo_v3._default.clear()


def test_detector_classes(hiplib):
    """YOLOV3 / YOLOV2 converter-class counterparts: uint8 image of any size in, (scores, boxes, classes) out."""
    from yolo_tensorflow_amd import detector
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (120, 160, 3), dtype=np.uint8)
    for cls, name in ((detector.YOLOV3Tiny, "yolov3-tiny"), (detector.YOLOV2TinyVoc, "yolov2-tiny-voc")):
        flat = IO.synth_weights(IO.parse_cfg(IO.cfg_text(name)), 0)
        d = cls(None, weights=flat, dtype=hiplib.FP32)
        scores, boxes, classes = d.detect_from_image(img)
        assert len(scores) == len(boxes) == len(classes) <= d.max_output_size
        # same tail from the oracle on the device's decoded tensor of the oracle-resized image
        det = d.engine.forward(np.ascontiguousarray(R.input_process(img, 416)), scale=1.0)[0]
        if "v3" in name:
            wb, ws, wc = R.detect_v3_tf(det, d.threshold, d.iou_threshold, d.max_output_size)
            assert len(ws) == len(scores)
            np.testing.assert_allclose(scores, ws, rtol=1e-3, atol=1e-4)
        d.engine.close()


def test_yolo_v2_entry_points(hiplib):
    """V2 demo chain (V2/Main.py): preprocess_image -> build_network -> decode (3 outputs) -> postprocess, and the
    `postprocess.decode` TF-NMS form, each stage against the oracle on the device's own previous-stage output."""
    from yolo_tensorflow_amd import yolo_v2
    size = 160
    txt = IO.with_input_size(IO.cfg_text("yolov2"), size)
    secs = IO.parse_cfg(txt); flat = IO.synth_weights(secs, 5)
    m = yolo_v2.Model(size=size, max_batch=1, dtype=hiplib.FP32, weights=flat)
    rng = np.random.default_rng(8)
    img = rng.integers(0, 256, (97, 131, 3), dtype=np.uint8)
    x = yolo_v2.preprocess_image(img, (size, size))              # the reference's rule: BGR in, cv2.resize's half-pixel bilinear, / 225
    assert np.array_equal(x, R.v2_preprocess_image(img, (size, size)))       # (same closed form, separately rounded float32 operations: bit for bit)
    xr = yolo_v2.preprocess_image(img, (120, 72), bgr=False)     # dsize = (width, height); an RGB caller
    assert xr.shape == (1, 72, 120, 3) and np.array_equal(xr[0], R.resize_cv2_linear(img.astype(np.float32), 72, 120) / np.float32(225.0))
    xl = yolo_v2.preprocess_image(img, (size, size), legacy_tf_resize=True)  # rounds 1-5's rule, kept behind a flag
    np.testing.assert_allclose(xl[0], R.resize_bilinear_legacy(img.astype(np.float32) / np.float32(255), size, size) * np.float32(255.0 / 225.0), rtol=0, atol=2e-6)
    g = size // 32
    raw = yolo_v2.build_network(x, model=m)                      # the reference's contract: the raw head (V2/model_darknet19_slim.py:198-200)
    assert raw.shape == (1, g, g, 425)
    out = yolo_v2.build_network(x, model=m, fused=True)          # the device's fused form: decoded rows
    assert out.shape == (1, g * g * 5, 85)
    osecs = R.parse_cfg(txt); params = R.unflatten_weights(flat, osecs)
    heads, _ = R.forward(osecs, params, x)
    np.testing.assert_allclose(raw, heads[0][1], rtol=2e-3, atol=2e-3 * float(np.abs(heads[0][1]).max()))
    rb, ro, rc = R.region_decode(heads[0][1], np.array(yolo_v2.anchors, np.float32), 80)
    bboxes, obj, cls = yolo_v2.decode(raw, output_sizes=(g, g), num_class=80)
    for got, fused in zip((bboxes, obj, cls), yolo_v2.decode(out, output_sizes=(g, g), num_class=80)):
        assert np.array_equal(got, fused)                        # decode(raw head) == decode(fused rows): the same region kernel on the same fp32 tensor
    assert bboxes.shape == (1, g * g, 5, 4) and obj.shape == (1, g * g, 5) and cls.shape == (1, g * g, 5, 80)
    np.testing.assert_allclose(bboxes, rb, rtol=2e-3, atol=2e-3)
    np.testing.assert_allclose(obj, ro, rtol=2e-3, atol=2e-3)
    np.testing.assert_allclose(cls, rc, rtol=2e-3, atol=2e-3)
    c = R.detections_boxes(out).reshape(1, g * g, 5, 85)
    assert np.array_equal(bboxes, c[..., :4]) and np.array_equal(obj, c[..., 4]) and np.array_equal(cls, c[..., 5:])
    thr = float(np.quantile((obj[..., None] * cls).max(-1), 0.9))
    b, s, k = yolo_v2.postprocess(bboxes, obj, cls, image_shape=img.shape[:2], threshold=thr)
    wb, ws, wk = R.v2_postprocess(bboxes, obj, cls, img.shape[:2], thr)
    assert len(ws) > 0
    assert np.array_equal(b, wb) and np.array_equal(s, ws) and np.array_equal(k, wk)
    m.close()


@pytest.mark.parametrize("dtype_name", ["bf16", "fp32", "fp8"])
def test_export_artifact_round_trip(hiplib, tmp_path, dtype_name):
    """yolo_export -> yolo_create_from_file: the artifact alone (no cfg, no .weights) reproduces the detections bit for
    bit; truncation, corruption and foreign files are rejected with an error (never a partial load)."""
    dtype = {"bf16": hiplib.BF16, "fp32": hiplib.FP32, "fp8": hiplib.FP8}[dtype_name]
    size = 96
    txt = IO.with_input_size(IO.cfg_text("yolov3"), size)
    secs = IO.parse_cfg(txt); flat = IO.synth_weights(secs, 11)
    img = np.random.default_rng(12).integers(0, 256, (2, size, size, 3), dtype=np.uint8)
    eng = hiplib.Engine(txt, max_batch=2, dtype=dtype, semantics=hiplib.SEM_DARKNET, decode=hiplib.DECODE_PIXEL)
    if dtype == hiplib.FP8:
        sc = np.ones(eng.num_layers, np.float32); sc[::3] = 0.5; sc[1::3] = 0.25
        eng.set_act_scales(sc)
    with pytest.raises(hiplib.YoloError):
        eng.export(str(tmp_path / "early.yolohip"))            # nothing to export before weights are loaded
    eng.set_weights(flat)
    want = eng.forward(img)
    path = str(tmp_path / "net.yolohip")
    eng.export(path); eng.close()
    eng2 = hiplib.Engine.from_file(path, max_batch=2)
    assert (eng2.size, eng2.rows, eng2.attrs) == (size, want.shape[1], 85)
    assert np.array_equal(eng2.forward(img), want)              # same semantics / decode mode / scales / parameters
    eng2.close()
    blob = open(path, "rb").read()
    bad = str(tmp_path / "bad.yolohip")
    for mutate in (lambda b: b[:len(b) // 2], lambda b: b[:-1], lambda b: b[:4096] + bytes([b[4096] ^ 1]) + b[4097:], lambda b: b"NOTYOLO1" + b[8:], lambda b: b""):
        open(bad, "wb").write(mutate(blob))
        with pytest.raises(hiplib.YoloError):
            hiplib.Engine.from_file(bad)
    with pytest.raises(hiplib.YoloError):
        hiplib.Engine.from_file(str(tmp_path / "missing.yolohip"))


def test_detect_skips_the_decoded_tensor_but_not_the_boxes(hiplib):
    """yolo_detect / yolo_detect_graph go from the head convs to box records without materialising the [n, rows, 5+C] decoded
    tensor (the decode writes scores, labels and the four box numbers only): same records as forward + postprocess, bit for bit,
    for the TF and the darknet NMS flavours; a flavour that needs the tensor afterwards is refused with a code, not served stale."""
    import ctypes as C
    txt = IO.with_input_size(IO.cfg_text("yolov3"), 160)
    secs = IO.parse_cfg(txt); flat = IO.synth_weights(secs, seed=2)
    img = np.random.default_rng(6).integers(0, 256, (3, 160, 160, 3), dtype=np.uint8)
    eng = hiplib.Engine(txt, max_batch=3)
    eng.set_weights(flat)
    for mode in (hiplib.NMS_TF, hiplib.NMS_DARKNET):
        eng.forward(img, want_detections=False)
        want = eng.postprocess(3, score_thr=0.3, iou_thr=0.45, max_out=30, nms_mode=mode)
        boxes = np.zeros((3, 30), dtype=hiplib.BOX_DTYPE); counts = np.zeros(3, np.int32)
        rc = eng.lib.yolo_detect(eng.ctx, img.ctypes.data, 3, hiplib.IMG_U8, hiplib.HOST, 1.0 / 255.0, 0.3, 0.45, 30, mode, hiplib.SELECT_GT,
                                 boxes.ctypes.data, counts.ctypes.data, hiplib.HOST)
        assert rc == 0
        assert sum(len(w) for w in want) > 10
        for b in range(3):
            assert counts[b] == len(want[b]) and np.array_equal(boxes[b, :counts[b]], want[b])
    with pytest.raises(hiplib.YoloError, match="without materialising"):
        eng.postprocess(3, score_thr=0.3, nms_mode=hiplib.NMS_NUMPY_V3)
    eng.forward(img, want_detections=False)
    assert len(eng.postprocess(3, score_thr=0.3, nms_mode=hiplib.NMS_NUMPY_V3)) == 3
    eng.close()


def test_detect_edge_thresholds_and_device_outputs(hiplib):
    """The objectness-first lean decode and the NMS kernel's own output handling at the edges: a threshold nothing passes (counts 0,
    every record slot zeroed even though the caller's device buffer held garbage), a threshold low enough that every box survives
    the objectness pre-filter (the per-lane decode does all the work), each equal to forward + postprocess."""
    import torch
    txt = IO.with_input_size(IO.cfg_text("yolov3"), 96)
    secs = IO.parse_cfg(txt); flat = IO.synth_weights(secs, seed=12)
    img = np.random.default_rng(16).integers(0, 256, (2, 96, 96, 3), dtype=np.uint8)
    eng = hiplib.Engine(txt, max_batch=2)
    eng.set_weights(flat)
    dimg = torch.from_numpy(img).cuda()
    for thr, mo in ((0.999999, 7), (0.02, 50), (0.3, 1)):
        eng.forward(img, want_detections=False)
        want = eng.postprocess(2, score_thr=thr, iou_thr=0.5, max_out=mo, nms_mode=hiplib.NMS_TF)
        boxes = torch.full((2, mo * 6), 0x7f7f7f7f, dtype=torch.int32, device="cuda"); counts = torch.full((2,), -5, dtype=torch.int32, device="cuda")
        for _ in range(3):                                        # eager, capture, replay
            eng.detect_graph(dimg, boxes, counts, score_thr=thr, iou_thr=0.5, max_out=mo, nms_mode=hiplib.NMS_TF)
        eng.synchronize()
        got_c = counts.cpu().numpy(); got = boxes.cpu().numpy().view(hiplib.BOX_DTYPE).reshape(2, mo)
        for b in range(2):
            assert got_c[b] == len(want[b]) <= mo
            assert np.array_equal(got[b, :got_c[b]], want[b])
            assert not got[b, got_c[b]:].view(np.uint8).any()     # unused slots are zero, whatever the buffer held
        if thr > 0.9:
            assert got_c.sum() == 0
        if thr < 0.1:
            assert got_c.min() > 5
    eng.close()



def test_postprocess_rows_and_graph_replay_state(hiplib):
    """(1) yolo_postprocess_rows: every record's row index points at the decoded row it was formed from (all NMS flavours);
    (2) ADVICE r02: forward(A) -> detect_graph replay on B -> postprocess must see image B's state (here: the replay went through the
    lean decode, so a flavour that needs the decoded tensor is refused, and the TF flavour equals detect(B)) -- never A's stale tensor."""
    import torch
    txt = IO.with_input_size(IO.cfg_text("yolov3"), 160)
    secs = IO.parse_cfg(txt); flat = IO.synth_weights(secs, seed=2)
    rng = np.random.default_rng(61)
    A = rng.integers(0, 256, (2, 160, 160, 3), dtype=np.uint8); B = rng.integers(0, 256, (2, 160, 160, 3), dtype=np.uint8)
    eng = hiplib.Engine(txt, max_batch=2)
    eng.set_weights(flat)
    det = eng.forward(A)
    for mode in (hiplib.NMS_TF, hiplib.NMS_DARKNET, hiplib.NMS_NUMPY_V3):
        recs, rows = eng.postprocess(2, score_thr=0.3, iou_thr=0.45, max_out=25, nms_mode=mode, return_rows=True)
        plain = eng.postprocess(2, score_thr=0.3, iou_thr=0.45, max_out=25, nms_mode=mode)
        assert sum(len(r) for r in recs) > 6
        for b in range(2):
            assert np.array_equal(recs[b], plain[b]) and len(rows[b]) == len(recs[b])
            d = det[b][rows[b]]
            if mode == hiplib.NMS_DARKNET:
                want = d[:, :4]
            else:
                h = np.float32(0.5)
                want = np.stack([d[:, 0] - d[:, 2] * h, d[:, 1] - d[:, 3] * h, d[:, 0] + d[:, 2] * h, d[:, 1] + d[:, 3] * h], -1)
            got = np.stack([recs[b]["x0"], recs[b]["y0"], recs[b]["x1"], recs[b]["y1"]], -1)
            assert np.array_equal(got, want)
            if mode != hiplib.NMS_NUMPY_V3:       # (that flavour reports shifted scores, V3/yolo_v3.py:414-418)
                assert np.array_equal(recs[b]["score"], (d[:, 4:5] * d[:, 5:]).max(-1))
                assert np.array_equal(recs[b]["cls"], (d[:, 4:5] * d[:, 5:]).argmax(-1))
    # single-operator form
    recs, rows = hiplib.op_postprocess(det, 0.3, 0.45, 25, return_rows=True)
    for b in range(2):
        assert np.array_equal(recs[b]["score"], (det[b][rows[b], 4:5] * det[b][rows[b], 5:]).max(-1))
    # ---- graph replay leaves the context in the state of the call it replays ----
    dB = torch.from_numpy(B).cuda()
    boxes = torch.zeros((2, 25 * 6), dtype=torch.int32, device="cuda"); counts = torch.zeros((2,), dtype=torch.int32, device="cuda")
    for _ in range(3):                              # eager, capture, replay
        eng.detect_graph(dB, boxes, counts, score_thr=0.3, iou_thr=0.45, max_out=25, nms_mode=hiplib.NMS_TF)
    eng.synchronize()
    wantB = eng.detect(B, score_thr=0.3, iou_thr=0.45, max_out=25, nms_mode=hiplib.NMS_TF)
    eng.forward(A)                                  # det_valid = True, decoded tensor of A resident
    eng.detect_graph(dB, boxes, counts, score_thr=0.3, iou_thr=0.45, max_out=25, nms_mode=hiplib.NMS_TF)      # replay on B
    eng.synchronize()
    got = eng.postprocess(2, score_thr=0.3, iou_thr=0.45, max_out=25, nms_mode=hiplib.NMS_TF)
    for b in range(2):
        assert np.array_equal(got[b], wantB[b])     # B's boxes with B's scores -- not A's stale geometry
    with pytest.raises(hiplib.YoloError, match="without materialising"):
        eng.postprocess(2, score_thr=0.3, nms_mode=hiplib.NMS_NUMPY_V3)
    with pytest.raises(hiplib.YoloError, match="lower threshold"):
        eng.postprocess(2, score_thr=0.1, nms_mode=hiplib.NMS_TF)
    eng.close()


def test_torch_default_stream_is_passed_on_as_the_legacy_stream(hiplib):
    """torch's default stream reads cuda_stream == 0, which at the C boundary would mean "create a stream": hip.Engine passes it on as
    hipStreamLegacy instead, so the engine's launches are ordered on the caller's stream and nothing is waited for on the host.  The legacy
    stream cannot be captured: detect_graph falls back to eager launches -- same records as an engine on a created stream, whose step
    replays as a graph."""
    import torch
    assert torch.cuda.current_stream().cuda_stream == 0
    txt = IO.with_input_size(IO.cfg_text("yolov3"), 160)
    secs = IO.parse_cfg(txt); flat = IO.synth_weights(secs, seed=2)
    img = torch.from_numpy(np.random.default_rng(8).integers(0, 256, (2, 160, 160, 3), dtype=np.uint8)).cuda()
    out = []
    side = torch.cuda.Stream()
    for stream in (torch.cuda.current_stream(), side):
        with torch.cuda.stream(stream):
            eng = hiplib.Engine(txt, max_batch=2, stream=stream.cuda_stream)
            assert not eng._own_stream
            eng.set_weights(flat)
            boxes = torch.zeros((2, 25 * 6), dtype=torch.int32, device="cuda"); counts = torch.zeros((2,), dtype=torch.int32, device="cuda")
            for _ in range(4):                                    # eager, capture (or its refusal), replays
                eng.detect_graph(img, boxes, counts, score_thr=0.3, iou_thr=0.45, max_out=25)
            stream.synchronize()
            out.append((boxes.cpu().numpy().copy(), counts.cpu().numpy().copy()))
            eng.close()
    assert out[0][1].sum() > 4
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])


def test_advice_r03_stale_uint8_pointer_and_dirty_list_counter(hiplib):
    """ADVICE r03.  (1) After a forward that read the caller's device-resident uint8 batch in place (fused stem), a timing / autotune pass
    over MORE images than that batch held must not read past the caller's buffer: it runs on the context's own input, and the next detect
    is unaffected.  (2) A detect call refused for its postprocess arguments is refused BEFORE the forward runs, and a replayed detect
    graph after such a call (or after any step that left the lean decode's list counter set) still equals the eager result -- no phantom
    boxes from stale list entries."""
    import torch
    txt = IO.with_input_size(IO.cfg_text("yolov3"), 160)
    secs = IO.parse_cfg(txt); flat = IO.synth_weights(secs, seed=4)
    rng = np.random.default_rng(9)
    img1 = torch.from_numpy(rng.integers(0, 256, (1, 160, 160, 3), dtype=np.uint8)).cuda()
    img4 = torch.from_numpy(rng.integers(0, 256, (4, 160, 160, 3), dtype=np.uint8)).cuda()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        eng = hiplib.Engine(txt, max_batch=4, stream=side.cuda_stream)
        eng.set_weights(flat)
        want = eng.detect(img4, score_thr=0.3, iou_thr=0.45, max_out=25)
        eng.forward(img1, want_detections=False)                 # the stem now points at a ONE-image buffer
        ms = eng.time_layers(4, 1)                               # four images: must not read 3 images past img1
        assert ms.shape[0] == eng.num_layers and np.isfinite(ms).all()
        t, cv = eng.time_forward(4, 1)
        assert t > 0 and cv > 0
        boxes = torch.zeros((4, 25 * 6), dtype=torch.int32, device="cuda"); counts = torch.zeros((4,), dtype=torch.int32, device="cuda")
        for _ in range(3):                                       # eager, capture, replay
            eng.detect_graph(img4, boxes, counts, score_thr=0.3, iou_thr=0.45, max_out=25)
        with pytest.raises(hiplib.YoloError, match="max_out"):   # refused before any launch
            eng.detect(img4, score_thr=0.3, iou_thr=0.45, max_out=0)
        with pytest.raises(hiplib.YoloError, match="nms/select"):
            eng.detect(img4, score_thr=0.3, iou_thr=0.45, max_out=25, nms_mode=17)
        eng.forward(img4, want_detections=False)                 # an eager pass in between (full decode: does not touch the list)
        for _ in range(2):
            eng.detect_graph(img4, boxes, counts, score_thr=0.3, iou_thr=0.45, max_out=25)
        side.synchronize()
        got_c = counts.cpu().numpy(); got = boxes.cpu().numpy().view(hiplib.BOX_DTYPE).reshape(4, 25)
        assert got_c.sum() > 4
        for b in range(4):
            assert got_c[b] == len(want[b]) and np.array_equal(got[b, :got_c[b]], want[b])
        eng.close()


def test_calibration_yardsticks(hiplib):
    """yolo_calibrate / yolo_calibrate_copy (what bench.py prints as roofline.calib_tflops / clock_ghz / calib_copy_gbs): a register-resident MFMA
    loop and a streaming copy.  Plausibility on an MI355X: below the quoted peaks (2.5 PFLOP/s at 2.4 GHz; 8 TB/s), above half of them, and
    repeatable to a few per cent."""
    t1, g1 = hiplib.calibrate(0.3)
    t2, g2 = hiplib.calibrate(0.3)
    assert 1200 < t1 < 2560 and 1.2 < g1 < 2.45, (t1, g1)
    assert abs(t1 - t2) / t1 < 0.05 and abs(g1 - g2) / g1 < 0.05
    th, gh = hiplib.calibrate(0.3, f16=True)
    assert 1200 < th < 2560 and 1.2 < gh < 2.45
    c = hiplib.calibrate_copy(0.2)
    assert 2000 < c < 8200, c
    with pytest.raises(hiplib.YoloError):
        hiplib.calibrate(0.0)

