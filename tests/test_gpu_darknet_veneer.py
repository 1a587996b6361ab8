"""libdarknet_hip.so (include/darknet_hip.h) against the reference's own libdarknet, compiled CPU-only from its sources
(oracle/_ref, oracle/Makefile): the same cfg, .weights file and image go through `load_network` ->
`network_predict_image` (letterbox) -> `get_network_boxes` -> `do_nms_sort` / `do_nms_obj` of BOTH libraries, called
through the same ctypes declarations the reference's binding uses (D2T/darknet.py:20-115)."""
import ctypes as C
import os
import numpy as np
import pytest
from oracle import darknet_ref as DR
from yolo_tensorflow_amd import darknet_io as IO

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class IMAGE(C.Structure):
    _fields_ = [("w", C.c_int), ("h", C.c_int), ("c", C.c_int), ("data", C.POINTER(C.c_float))]


def _bind(lib):
    lib.load_network.argtypes = [C.c_char_p, C.c_char_p, C.c_int]; lib.load_network.restype = C.c_void_p
    lib.free_network.argtypes = [C.c_void_p]
    lib.network_width.argtypes = [C.c_void_p]; lib.network_height.argtypes = [C.c_void_p]
    lib.network_predict_image.argtypes = [C.c_void_p, IMAGE]; lib.network_predict_image.restype = C.POINTER(C.c_float)
    lib.network_predict.argtypes = [C.c_void_p, C.POINTER(C.c_float)]; lib.network_predict.restype = C.POINTER(C.c_float)
    lib.get_network_boxes.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int)]
    lib.get_network_boxes.restype = C.POINTER(DR.DETECTION)
    lib.free_detections.argtypes = [C.POINTER(DR.DETECTION), C.c_int]
    lib.do_nms_sort.argtypes = [C.POINTER(DR.DETECTION), C.c_int, C.c_int, C.c_float]
    lib.do_nms_obj.argtypes = [C.POINTER(DR.DETECTION), C.c_int, C.c_int, C.c_float]
    return lib


def _collect(dets, n, classes):
    bb = np.zeros((n, 4), np.float32); obj = np.zeros(n, np.float32); pr = np.zeros((n, classes), np.float32)
    for i in range(n):
        d = dets[i]
        bb[i] = (d.bbox.x, d.bbox.y, d.bbox.w, d.bbox.h); obj[i] = d.objectness
        pr[i] = np.ctypeslib.as_array(d.prob, shape=(classes,))
    return bb, obj, pr


@pytest.fixture(scope="module")
def pair(tmp_path_factory, hiplib):
    if not DR.available():
        pytest.skip("oracle/_ref/libdarknet_ref.so not built")
    os.environ["DARKNET_HIP_DTYPE"] = "fp32"
    tmp = tmp_path_factory.mktemp("veneer")
    size = 160
    txt = IO.with_input_size(IO.cfg_text("yolov3-tiny"), size)
    secs = IO.parse_cfg(txt); flat = IO.synth_weights(secs, 21, obj_bias=0.0)
    cfg = str(tmp / "net.cfg"); wf = str(tmp / "net.weights")
    open(cfg, "w").write(txt); IO.write_weights_file(wf, flat, 0, 2)
    ref = _bind(DR.lib())
    ven = _bind(C.CDLL(os.path.join(ROOT, "yolo_tensorflow_amd", "libdarknet_hip.so")))
    with DR._Quiet():
        rnet = ref.load_network(cfg.encode(), wf.encode(), 0)
        ref.set_batch_network(rnet, 1)
    vnet = ven.load_network(cfg.encode(), wf.encode(), 0)
    assert vnet and ven.network_width(vnet) == size and ven.network_height(vnet) == size
    yield ref, rnet, ven, vnet, size
    ven.free_network(vnet)
    ref.free_network(rnet)


def _predict_both(pair, w, h, seed):
    ref, rnet, ven, vnet, size = pair
    img = np.ascontiguousarray(np.random.default_rng(seed).random((3, h, w), dtype=np.float32))
    im = IMAGE(w, h, 3, img.ctypes.data_as(C.POINTER(C.c_float)))
    ref.network_predict_image(rnet, im)
    out = ven.network_predict_image(vnet, im)
    assert bool(out)
    return img


@pytest.mark.parametrize("w,h,relative", [(200, 120, 1), (96, 250, 1), (160, 160, 0), (331, 207, 0)])
def test_predict_image_and_boxes_match_libdarknet(pair, w, h, relative):
    ref, rnet, ven, vnet, size = pair
    _predict_both(pair, w, h, seed=w * 7 + h)
    # a threshold that no objectness sits within 2e-3 of, so both sides select the same boxes
    n0 = C.c_int(0)
    d0 = ref.get_network_boxes(rnet, w, h, 0.0, .5, None, relative, C.byref(n0))
    _, obj_all, _ = _collect(d0, n0.value, 80); ref.free_detections(d0, n0.value)
    cand = np.sort(obj_all)[::-1]
    thresh = None
    for k in range(40, len(cand) - 1):
        if cand[k] - cand[k + 1] > 8e-3:
            thresh = float((cand[k] + cand[k + 1]) / 2); break
    assert thresh is not None
    nr, nv = C.c_int(0), C.c_int(0)
    dr = ref.get_network_boxes(rnet, w, h, thresh, .5, None, relative, C.byref(nr))
    dv = ven.get_network_boxes(vnet, w, h, thresh, .5, None, relative, C.byref(nv))
    assert nv.value == nr.value > 20
    br, orr, pr = _collect(dr, nr.value, 80); bv, ov, pv = _collect(dv, nv.value, 80)
    scale = max(w, h) if not relative else 1.0
    np.testing.assert_allclose(bv, br, rtol=2e-3, atol=2e-3 * scale)       # same order: head, cell, anchor
    np.testing.assert_allclose(ov, orr, rtol=2e-3, atol=2e-4)
    np.testing.assert_allclose(pv, pr, rtol=2e-3, atol=2e-3)               # entries near the threshold may flip to 0
    # NMS on identical inputs: give the veneer the reference's detections so that both sides start from the same bits
    for i in range(nr.value):
        dv[i].bbox = dr[i].bbox; dv[i].objectness = dr[i].objectness
        C.memmove(dv[i].prob, dr[i].prob, 80 * 4)
    ref.do_nms_sort(dr, nr.value, 80, 0.45); ven.do_nms_sort(dv, nv.value, 80, 0.45)
    _, _, pr2 = _collect(dr, nr.value, 80); br2 = _collect(dr, nr.value, 80)[0]
    bv2, _, pv2 = _collect(dv, nv.value, 80)
    # the reference qsorts its array: match rows by box before comparing
    key_r = {tuple(b): i for i, b in enumerate(br2)}
    order = [key_r[tuple(b)] for b in bv2]
    assert sorted(order) == list(range(nr.value))
    assert np.array_equal(pv2, pr2[order])
    assert (pv2 == 0).sum() > (pv == 0).sum()                              # something was suppressed
    ref.free_detections(dr, nr.value); ven.free_detections(dv, nv.value)


def test_do_nms_obj_matches_libdarknet(pair):
    ref, rnet, ven, vnet, size = pair
    w, h = 180, 140
    _predict_both(pair, w, h, seed=5)
    nr, nv = C.c_int(0), C.c_int(0)
    dr = ref.get_network_boxes(rnet, w, h, 0.35, .5, None, 1, C.byref(nr))
    dv = ven.get_network_boxes(vnet, w, h, 0.0, .5, None, 1, C.byref(nv))
    assert nv.value >= nr.value > 20
    for i in range(nr.value):                                              # identical starting bits on both sides
        dv[i].bbox = dr[i].bbox; dv[i].objectness = dr[i].objectness
        C.memmove(dv[i].prob, dr[i].prob, 80 * 4)
    ref.do_nms_obj(dr, nr.value, 80, 0.3); ven.do_nms_obj(dv, nr.value, 80, 0.3)
    br, orr, pr = _collect(dr, nr.value, 80); bv, ov, pv = _collect(dv, nr.value, 80)
    key_r = {tuple(b): i for i, b in enumerate(br)}
    order = [key_r[tuple(b)] for b in bv]
    assert np.array_equal(ov, orr[order]) and np.array_equal(pv, pr[order])
    assert (ov == 0).sum() > 0
    ref.free_detections(dr, nr.value); ven.free_detections(dv, nv.value)


def test_network_predict_returns_last_layer_output(pair):
    """network_predict's return value is net->output (DN/network.c:497-508): the last [yolo] layer's planar output with its
    logistic activations -- same floats as the compiled reference returns; and network_predict on a planar image already at
    network size equals network_predict_image of it (the letterbox of an S x S image is the identity)."""
    ref, rnet, ven, vnet, size = pair
    img = np.ascontiguousarray(np.random.default_rng(1).random((3, size, size), dtype=np.float32))
    g = size // 16                                                          # yolov3-tiny's last head: stride 16
    n_out = 255 * g * g
    pr = ref.network_predict(rnet, img.ctypes.data_as(C.POINTER(C.c_float)))
    want = np.ctypeslib.as_array(pr, shape=(n_out,)).copy()
    a = np.ctypeslib.as_array(ven.network_predict(vnet, img.ctypes.data_as(C.POINTER(C.c_float))), shape=(n_out,)).copy()
    np.testing.assert_allclose(a, want, rtol=2e-3, atol=2e-4)
    im = IMAGE(size, size, 3, img.ctypes.data_as(C.POINTER(C.c_float)))
    b = np.ctypeslib.as_array(ven.network_predict_image(vnet, im), shape=(n_out,)).copy()
    np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-6)


def test_make_network_boxes_and_batch(pair):
    """make_network_boxes (DN/network.c:526): as many zeroed detections as get_network_boxes reports; set_batch_network(2):
    two planar images in, the first image's boxes out (DN/network.c:339, get_yolo_detections reads batch element 0)."""
    ref, rnet, ven, vnet, size = pair
    ven.make_network_boxes.argtypes = [C.c_void_p, C.c_float, C.POINTER(C.c_int)]; ven.make_network_boxes.restype = C.POINTER(DR.DETECTION)
    ven.set_batch_network.argtypes = [C.c_void_p, C.c_int]
    rng = np.random.default_rng(31)
    two = np.ascontiguousarray(rng.random((2, 3, size, size), dtype=np.float32))
    assert bool(ven.network_predict(vnet, two[0].ctypes.data_as(C.POINTER(C.c_float))))
    n0, n1 = C.c_int(0), C.c_int(0)
    d0 = ven.get_network_boxes(vnet, size, size, 0.4, .5, None, 1, C.byref(n0))
    d1 = ven.make_network_boxes(vnet, 0.4, C.byref(n1))
    assert n1.value == n0.value > 5 and d1[0].classes == 80 and d1[0].prob[3] == 0.0 and d1[n1.value - 1].objectness == 0.0
    b0, o0, p0 = _collect(d0, n0.value, 80)
    ven.free_detections(d1, n1.value); ven.free_detections(d0, n0.value)
    ven.set_batch_network(vnet, 2)
    assert bool(ven.network_predict(vnet, two.ctypes.data_as(C.POINTER(C.c_float))))
    n2 = C.c_int(0)
    d2 = ven.get_network_boxes(vnet, size, size, 0.4, .5, None, 1, C.byref(n2))
    b2, o2, p2 = _collect(d2, n2.value, 80)
    assert n2.value == n0.value and np.array_equal(b2, b0) and np.array_equal(o2, o0) and np.array_equal(p2, p0)
    ven.free_detections(d2, n2.value)
    # net->output of a batch is batch * outputs contiguous floats (DN/network.c:497-508): every image's head, not just the first
    # (the reference's set_batch_network does not re-allocate a batch-1 network's buffers, so its two outputs come from two batch-1 calls)
    outs = (size // 16) ** 2 * 255                                            # last [yolo] layer of yolov3-tiny: the stride-16 head
    want = np.stack([np.ctypeslib.as_array(ref.network_predict(rnet, two[b].ctypes.data_as(C.POINTER(C.c_float))), shape=(outs,)).copy() for b in range(2)])
    v_out = ven.network_predict(vnet, two.ctypes.data_as(C.POINTER(C.c_float)))
    got = np.ctypeslib.as_array(v_out, shape=(2, outs)).copy()
    assert not np.array_equal(want[0], want[1])
    np.testing.assert_allclose(got, want, rtol=2e-3, atol=2e-4)
    ven.set_batch_network(vnet, 1)


def test_image_helpers_match_libdarknet(pair, tmp_path):
    """letterbox_image (device), rgbgr_image, load_image_color (PPM, with and without the resize) and get_metadata against the
    compiled reference's functions of the same names (load_image_color: against the reference's resize_image of the same pixels,
    its stb decoder is outside the path); free_ptrs releases what get_metadata allocated."""
    ref, rnet, ven, vnet, size = pair

    class METADATA(C.Structure):
        _fields_ = [("classes", C.c_int), ("names", C.POINTER(C.c_char_p))]
    for lib in (ref, ven):
        lib.letterbox_image.argtypes = [IMAGE, C.c_int, C.c_int]; lib.letterbox_image.restype = IMAGE
        lib.rgbgr_image.argtypes = [IMAGE]; lib.free_image.argtypes = [IMAGE]
        lib.get_metadata.argtypes = [C.c_char_p]; lib.get_metadata.restype = METADATA
        lib.free_ptrs.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    ref.resize_image.argtypes = [IMAGE, C.c_int, C.c_int]; ref.resize_image.restype = IMAGE
    ven.load_image_color.argtypes = [C.c_char_p, C.c_int, C.c_int]; ven.load_image_color.restype = IMAGE
    rng = np.random.default_rng(77)
    for (w, h, W, H) in ((200, 120, 160, 160), (96, 250, 208, 128), (64, 64, 64, 64)):
        img = np.ascontiguousarray(rng.random((3, h, w), dtype=np.float32))
        im = IMAGE(w, h, 3, img.ctypes.data_as(C.POINTER(C.c_float)))
        a = ref.letterbox_image(im, W, H); b = ven.letterbox_image(im, W, H)
        assert (b.w, b.h, b.c) == (a.w, a.h, a.c) == (W, H, 3)
        x = np.ctypeslib.as_array(a.data, shape=(3, H, W)).copy(); y = np.ctypeslib.as_array(b.data, shape=(3, H, W)).copy()
        np.testing.assert_allclose(y, x, rtol=0, atol=2e-6)
        ref.rgbgr_image(a); ven.rgbgr_image(b)
        assert np.array_equal(np.ctypeslib.as_array(b.data, shape=(3, H, W)), y[::-1])
        np.testing.assert_allclose(np.ctypeslib.as_array(b.data, shape=(3, H, W)), np.ctypeslib.as_array(a.data, shape=(3, H, W)), rtol=0, atol=2e-6)
        ref.free_image(a); ven.free_image(b)
    # PPM in, planar float out; with a target size: darknet's resize_image
    rgb = rng.integers(0, 256, (50, 70, 3), dtype=np.uint8)
    ppm = str(tmp_path / "a.ppm")
    with open(ppm, "wb") as f:
        f.write(b"P6\n# a comment\n70 50\n255\n" + rgb.tobytes())
    im = ven.load_image_color(ppm.encode(), 0, 0)
    assert (im.w, im.h, im.c) == (70, 50, 3)
    planar = np.ctypeslib.as_array(im.data, shape=(3, 50, 70)).copy()
    assert np.array_equal(planar, (rgb.transpose(2, 0, 1).astype(np.float32) / np.float64(255.)).astype(np.float32))
    im2 = ven.load_image_color(ppm.encode(), 96, 64)
    rim = ref.resize_image(im, 96, 64)
    np.testing.assert_allclose(np.ctypeslib.as_array(im2.data, shape=(3, 64, 96)), np.ctypeslib.as_array(rim.data, shape=(3, 64, 96)), rtol=0, atol=2e-6)
    ven.free_image(im); ven.free_image(im2); ref.free_image(rim)
    bad = ven.load_image_color(str(tmp_path / "missing.ppm").encode(), 0, 0)
    assert not bad.data
    # .data file + names list
    names = ["person", "bicycle", "traffic light", "dog"]
    (tmp_path / "n.names").write_text("\n".join(names) + "\n")
    (tmp_path / "n.data").write_text("classes= 4\ntrain  = /nowhere\nnames = %s\nbackup = /nowhere\n" % (tmp_path / "n.names"))
    with DR._Quiet():
        mr = ref.get_metadata(str(tmp_path / "n.data").encode())
    mv = ven.get_metadata(str(tmp_path / "n.data").encode())
    assert mv.classes == mr.classes == 4
    assert [mv.names[i] for i in range(4)] == [mr.names[i] for i in range(4)] == [n.encode() for n in names]
    ven.free_ptrs(C.cast(mv.names, C.POINTER(C.c_void_p)), 4)


def test_region_head_boxes_match_libdarknet(tmp_path, hiplib):
    """A [region] topology (yolov2-tiny-voc: maxpool stack, softmax head, 4-int .weights header): every box of
    get_network_boxes -- darknet reports all w*h*n of them, anchor-major, objectness / probabilities gated by thresh."""
    if not DR.available():
        pytest.skip("oracle/_ref/libdarknet_ref.so not built")
    os.environ["DARKNET_HIP_DTYPE"] = "fp32"
    size = 160
    txt = IO.with_input_size(IO.cfg_text("yolov2-tiny-voc"), size)
    secs = IO.parse_cfg(txt); flat = IO.synth_weights(secs, 23, obj_bias=0.0)
    cfg = str(tmp_path / "net.cfg"); wf = str(tmp_path / "net.weights")
    open(cfg, "w").write(txt); IO.write_weights_file(wf, flat, 0, 1)
    ref = _bind(DR.lib())
    ven = _bind(C.CDLL(os.path.join(ROOT, "yolo_tensorflow_amd", "libdarknet_hip.so")))
    with DR._Quiet():
        rnet = ref.load_network(cfg.encode(), wf.encode(), 0)
        ref.set_batch_network(rnet, 1)
    vnet = ven.load_network(cfg.encode(), wf.encode(), 0)
    assert vnet
    w, h = 210, 130
    img = np.ascontiguousarray(np.random.default_rng(4).random((3, h, w), dtype=np.float32))
    im = IMAGE(w, h, 3, img.ctypes.data_as(C.POINTER(C.c_float)))
    ref.network_predict_image(rnet, im); assert bool(ven.network_predict_image(vnet, im))
    nr, nv = C.c_int(0), C.c_int(0)
    dr = ref.get_network_boxes(rnet, w, h, 0.0, .5, None, 1, C.byref(nr))
    dv = ven.get_network_boxes(vnet, w, h, 0.0, .5, None, 1, C.byref(nv))
    g = size // 32
    assert nr.value == nv.value == g * g * 5
    br, orr, pr = _collect(dr, nr.value, 20); bv, ov, pv = _collect(dv, nv.value, 20)
    np.testing.assert_allclose(bv, br, rtol=2e-3, atol=2e-3)
    np.testing.assert_allclose(ov, orr, rtol=2e-3, atol=2e-4)
    np.testing.assert_allclose(pv, pr, rtol=2e-3, atol=2e-4)
    ref.free_detections(dr, nr.value); ven.free_detections(dv, nv.value)
    # with a threshold: gated objectness / probabilities (entries within 1e-3 of the threshold excluded from the comparison)
    thr = 0.3
    dr = ref.get_network_boxes(rnet, w, h, thr, .5, None, 1, C.byref(nr)); dv = ven.get_network_boxes(vnet, w, h, thr, .5, None, 1, C.byref(nv))
    _, o2r, p2r = _collect(dr, nr.value, 20); _, o2v, p2v = _collect(dv, nv.value, 20)
    clear_o = np.abs(orr - thr) > 1e-3
    assert np.array_equal(o2v[clear_o] == 0, o2r[clear_o] == 0) and (o2r == 0).sum() > 0
    clear_p = (np.abs(pr - thr) > 1e-3) & clear_o[:, None]
    assert np.array_equal((p2v == 0)[clear_p], (p2r == 0)[clear_p])
    ref.free_detections(dr, nr.value); ven.free_detections(dv, nv.value)
    ven.free_network(vnet); ref.free_network(rnet)


def test_python_detect_matches_reference_binding(pair, tmp_path):
    """`yolo_tensorflow_amd.darknet_hip.detect` (same flow as D2T/darknet.py:125-142) against that flow run on the compiled
    reference: same (class, prob, box) list."""
    from yolo_tensorflow_amd import darknet_hip as DK
    ref, rnet, ven, vnet, size = pair
    w, h = 190, 150
    rgb = np.random.default_rng(8).integers(0, 256, (h, w, 3), dtype=np.uint8)
    names = ["c%d" % i for i in range(80)]
    # the fixture's network, re-opened through the module's own loader
    size_txt = IO.with_input_size(IO.cfg_text("yolov3-tiny"), size)
    flat = IO.synth_weights(IO.parse_cfg(size_txt), 21, obj_bias=0.0)
    cfg = str(tmp_path / "n.cfg"); wf = str(tmp_path / "n.weights")
    open(cfg, "w").write(size_txt); IO.write_weights_file(wf, flat, 0, 2)
    net = DK.load_net(cfg, wf, 0)
    got = DK.detect(net, names, rgb, thresh=0.4, nms=0.45)
    # the reference's own calling convention: a file name and a METADATA (D2T/darknet.py:125-126, 162-164)
    ppm = str(tmp_path / "img.ppm")
    with open(ppm, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (w, h) + rgb.tobytes())
    (tmp_path / "c.names").write_text("\n".join(names) + "\n")
    (tmp_path / "c.data").write_text("classes=80\nnames=%s\n" % (tmp_path / "c.names"))
    meta = DK.load_meta(str(tmp_path / "c.data"))
    got_file = DK.detect(net, meta, ppm, thresh=0.4, nms=0.45)
    assert len(got_file) == len(got)
    for (n1, p1, b1), (n2, p2, b2) in zip(got_file, got):
        assert n1.decode() == n2 and p1 == p2 and b1 == b2
    DK.free_net(net)
    im = DK.array_to_image(rgb)
    rim = IMAGE(im.w, im.h, im.c, im.data)
    ref.network_predict_image(rnet, rim)
    num = C.c_int(0)
    dets = ref.get_network_boxes(rnet, w, h, 0.4, .5, None, 0, C.byref(num))
    ref.do_nms_obj(dets, num.value, 80, 0.45)
    want = []
    for j in range(num.value):
        for i in range(80):
            if dets[j].prob[i] > 0:
                b = dets[j].bbox
                want.append((names[i], dets[j].prob[i], (b.x, b.y, b.w, b.h)))
    want = sorted(want, key=lambda x: -x[1])
    ref.free_detections(dets, num.value)
    assert len(got) > 10 and abs(len(got) - len(want)) <= 2          # entries within float noise of the threshold may flip
    hits = 0
    for n_, p, bx in want:                                            # same class, same probability, same box
        for n2, p2, bx2 in got:
            if n2 == n_ and abs(p2 - p) < 3e-3 and np.allclose(bx2, bx, rtol=3e-3, atol=3e-3 * max(w, h)):
                hits += 1
                break
    assert hits >= len(want) - 3


def test_detection_head_matches_libdarknet(tmp_path, hiplib):
    """A [detection] topology (YOLOv1 style: convs, max-pools, [connected], [dropout], [detection]; 4-int .weights header) through
    both libraries: `network_predict_image` returns the prediction vector, `get_network_boxes` every one of the side*side*num boxes
    in layer order with probabilities gated by the threshold (get_detection_detections, DN/detection_layer.c:225-254)."""
    if not DR.available():
        pytest.skip("oracle/_ref/libdarknet_ref.so not built")
    os.environ["DARKNET_HIP_DTYPE"] = "fp32"
    S, B, Cn = 4, 2, 5
    txt = ("[net]\nbatch=1\nsubdivisions=1\nheight=64\nwidth=64\nchannels=3\n\n"
           "[convolutional]\nbatch_normalize=1\nfilters=16\nsize=3\nstride=1\npad=1\nactivation=leaky\n\n[maxpool]\nsize=2\nstride=2\n\n"
           "[convolutional]\nfilters=32\nsize=3\nstride=2\npad=1\nactivation=leaky\n\n[maxpool]\nsize=2\nstride=2\n\n"
           "[convolutional]\nfilters=24\nsize=1\nstride=1\npad=1\nactivation=leaky\n\n[maxpool]\nsize=2\nstride=2\n\n"
           "[connected]\noutput=96\nactivation=leaky\n\n[dropout]\nprobability=.5\n\n"
           "[connected]\noutput=%d\nactivation=linear\n\n"
           "[detection]\nclasses=%d\ncoords=4\nrescore=1\nside=%d\nnum=%d\nsoftmax=0\nsqrt=1\njitter=.2\n"
           "object_scale=1\nnoobject_scale=.5\nclass_scale=1\ncoord_scale=5\n" % (S * S * (Cn + 5 * B), Cn, S, B))
    secs = IO.parse_cfg(txt); flat = IO.synth_weights(secs, 31)
    cfg = str(tmp_path / "v1.cfg"); wf = str(tmp_path / "v1.weights")
    open(cfg, "w").write(txt); IO.write_weights_file(wf, flat, 0, 1)
    ref = _bind(DR.lib())
    ven = _bind(C.CDLL(os.path.join(ROOT, "yolo_tensorflow_amd", "libdarknet_hip.so")))
    with DR._Quiet():
        rnet = ref.load_network(cfg.encode(), wf.encode(), 0)
        ref.set_batch_network(rnet, 1)
    vnet = ven.load_network(cfg.encode(), wf.encode(), 0)
    assert vnet
    w, h = 90, 70
    img = np.ascontiguousarray(np.random.default_rng(3).random((3, h, w), dtype=np.float32))
    im = IMAGE(w, h, 3, img.ctypes.data_as(C.POINTER(C.c_float)))
    n_out = S * S * (Cn + 5 * B)
    outr = np.ctypeslib.as_array(ref.network_predict_image(rnet, im), shape=(n_out,)).copy()
    pv = ven.network_predict_image(vnet, im); assert bool(pv)
    outv = np.ctypeslib.as_array(pv, shape=(n_out,)).copy()
    np.testing.assert_allclose(outv, outr, rtol=2e-4, atol=2e-5)
    thresh = 0.25
    nr, nv = C.c_int(0), C.c_int(0)
    dr = ref.get_network_boxes(rnet, w, h, thresh, .5, None, 0, C.byref(nr))
    dv = ven.get_network_boxes(vnet, w, h, thresh, .5, None, 0, C.byref(nv))
    assert nv.value == nr.value == S * S * B
    br, orr, pr = _collect(dr, nr.value, Cn); bv, ov, pvv = _collect(dv, nv.value, Cn)
    np.testing.assert_allclose(bv, br, rtol=2e-4, atol=2e-3)
    np.testing.assert_allclose(ov, orr, rtol=2e-4, atol=2e-5)
    clear = np.abs(orr[:, None] * outr[:S * S * Cn].reshape(S * S, Cn).repeat(B, 0) - thresh) > 1e-3
    assert np.array_equal((pvv > 0)[clear], (pr > 0)[clear]) and (pr > 0).any() and (pr == 0).any()
    np.testing.assert_allclose(pvv[clear], pr[clear], rtol=2e-4, atol=2e-5)
    ref.free_detections(dr, nr.value); ven.free_detections(dv, nv.value)
    ven.free_network(vnet); ref.free_network(rnet)

