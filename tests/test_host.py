"""Host-side logic that needs no GPU: topology files, .weights streams, the C-ABI library's exports."""
import ctypes
import os
import re
import numpy as np
import pytest
from conftest import ROOT, has_gpu
from yolo_tensorflow_amd import darknet_io as IO


def test_yolov3_topology_matches_reference_tables():
    secs = IO.parse_cfg(IO.cfg_text("yolov3"))
    shapes = IO.layer_shapes(secs)
    assert len(shapes) == 107                                   # V3/yolov3.txt: layers 0..106
    convs = IO.conv_specs(secs)
    assert len(convs) == 75
    flops = 0.0
    for c in convs:
        t, H, W, C, cin = shapes[c["index"]]
        flops += 2.0 * c["size"] ** 2 * cin * C * H * W
    assert abs(flops / 1e9 - 65.86) < 0.01                      # V3/yolov3.txt sum == darknet numops
    assert IO.weights_count(secs) == 62001757                   # yolov3.weights payload (248 MB / 4 - header)
    assert shapes[36][1:4] == (52, 52, 256) and shapes[61][1:4] == (26, 26, 512) and shapes[86][1:4] == (26, 26, 768)
    assert shapes[98][1:4] == (52, 52, 384)
    rows = sum(s[1] * s[2] * 3 for s in shapes if s[0] == "yolo")
    assert rows == 10647
    secs608 = IO.parse_cfg(IO.cfg_text("yolov3-608"))
    assert sum(s[1] * s[2] * 3 for s in IO.layer_shapes(secs608) if s[0] == "yolo") == 22743


def test_yolov2_topology_matches_reference_tables():
    secs = IO.parse_cfg(IO.cfg_text("yolov2"))
    shapes = IO.layer_shapes(secs)
    assert len(shapes) == 32 and shapes[28][1:4] == (13, 13, 1280) and shapes[30][1:4] == (13, 13, 425)
    flops = sum(2.0 * c["size"] ** 2 * shapes[c["index"]][4] * shapes[c["index"]][3] * shapes[c["index"]][1] * shapes[c["index"]][2]
                for c in IO.conv_specs(secs))
    assert abs(flops / 1e9 - 29.46) < 0.02                      # V2/yolov2.txt
    assert IO.default_header(secs) == (0, 1)


def test_weights_file_round_trip(tmp_path):
    secs = IO.parse_cfg(IO.cfg_text("yolov3-tiny"))
    flat = IO.synth_weights(secs, seed=3)
    assert flat.size == IO.weights_count(secs)
    for (mj, mn, hdr) in ((0, 2, 5), (0, 1, 4)):
        p = str(tmp_path / ("w%d.weights" % hdr))
        IO.write_weights_file(p, flat, mj, mn, 0, seen=12345)
        assert os.path.getsize(p) == hdr * 4 + flat.size * 4
        back, ver = IO.read_weights_file(p)
        assert ver == (mj, mn, 0, 12345) and np.array_equal(back, flat)
        back2, _ = IO.read_weights_file(p, header_ints=hdr)
        assert np.array_equal(back2, flat)
    assert np.array_equal(IO.synth_weights(secs, seed=3), flat)          # seeded => reproducible


def test_with_input_size():
    secs = IO.parse_cfg(IO.with_input_size(IO.cfg_text("yolov3"), 608))
    assert secs[0]["width"] == "608" and IO.layer_shapes(secs)[-1][1] == 76


def _header_functions():
    text = open(os.path.join(ROOT, "include", "yolo_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(yolo_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    """The C-ABI library loads on a CPU-only box and exports exactly what include/yolo_hip.h declares."""
    from yolo_tensorflow_amd import hip
    lib = hip.load_library()
    declared = _header_functions()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), "libyolo_hip.so does not export " + name
    assert sorted(hip.EXPORTS) == declared
    assert ctypes.sizeof(hip._Config) == 48 and hip.BOX_DTYPE.itemsize == 24


def test_sharding_entry_points_match_the_python_host():
    """include/yolo_dist.h: every declared symbol is exported and bound; the C split of the batch and of the gathered record
    buffers equals yolo_tensorflow_amd/dist.py's (what bench.py runs over torch.distributed), ragged splits included."""
    import torch
    from yolo_tensorflow_amd import hip, dist
    lib = hip.load_library()
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "yolo_dist.h")).read(), flags=re.S)
    declared = sorted(set(re.findall(r"\b(yolo_[a-z0-9_]+)\s*\(", text)))
    assert declared == sorted(hip.DIST_EXPORTS)
    for name in declared:
        assert hasattr(lib, name), "libyolo_hip.so does not export " + name
    rng = np.random.default_rng(5)
    for world, batch, max_out in [(1, 3, 20), (2, 32, 20), (3, 8, 5), (4, 2, 7), (8, 64, 20), (8, 33, 3)]:
        for r in range(world):
            assert hip.shard_bounds(batch, world, r) == dist.shard_bounds(batch, world, r)
        per = -(-batch // world)
        flat = per * max_out * 6 + per
        assert lib.yolo_dist_flat_words(per, max_out) == flat
        gathered = rng.integers(-2**31, 2**31 - 1, (world, flat), dtype=np.int64).astype(np.int32)
        boxes, counts = hip.dist_split_records(gathered, world, batch, max_out)
        tb, tc = dist.split_flat_records_ragged(torch.from_numpy(gathered), batch, max_out)
        assert np.array_equal(boxes.view(np.int32).reshape(batch, max_out * 6), tb.numpy()) and np.array_equal(counts, tc.numpy())
    first, count = ctypes.c_int(0), ctypes.c_int(0)
    assert lib.yolo_shard_bounds(8, 2, 2, ctypes.byref(first), ctypes.byref(count)) == -1      # rank outside the world
    assert lib.yolo_dist_create(None, 1, 0, None, None, 1, 1, None, 0) is None


def test_plain_c_consumer_of_the_abi(tmp_path):
    """tests/c/abi_consumer.c: the public headers compile as C99 (-pedantic), the libraries link from C, and the host-side entry
    points return codes a C caller can act on.  No device call is made."""
    import subprocess
    exe = str(tmp_path / "abi_consumer")
    libdir = os.path.join(ROOT, "yolo_tensorflow_amd")
    cc = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(ROOT, "include"),
                         os.path.join(ROOT, "tests", "c", "abi_consumer.c"), "-o", exe, "-L" + libdir, "-lyolo_hip", "-ldarknet_hip",
                         "-Wl,-rpath," + libdir], capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0 and "abi_consumer ok" in run.stdout, run.stderr


@pytest.mark.skipif(has_gpu(), reason="checks the no-GPU failure mode")
def test_no_cpu_fallback_without_device():
    from yolo_tensorflow_amd import hip
    with pytest.raises(hip.YoloError, match="no HIP device|hipSetDevice|HIP"):
        hip.Engine(IO.cfg_text("yolov3-tiny"), max_batch=1)
    with pytest.raises(hip.YoloError):
        hip.op_upsample2x(np.zeros((1, 2, 2, 8), np.float32))


def test_create_rejects_bad_config():
    from yolo_tensorflow_amd import hip
    lib = hip.load_library()
    err = ctypes.create_string_buffer(256)
    assert lib.yolo_create(None, err, 256) is None and b"yolo_config" in err.value
    conf = hip._Config(ctypes.sizeof(hip._Config), b"[net]\nwidth=32\nheight=32\nchannels=3\n", 0, 0, 0, 0, 0, 0, None)
    assert lib.yolo_create(ctypes.byref(conf), err, 256) is None and b"max_batch" in err.value


def test_darknet_veneer_exports_every_declared_symbol():
    """libdarknet_hip.so loads and exports every function include/darknet_hip.h declares (no GPU call is made)."""
    import ctypes, re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "include", "darknet_hip.h")).read()
    names = re.findall(r"^\w[\w \*]*?\b(\w+)\(", text, flags=re.M)
    assert {"load_network", "network_predict_image", "get_network_boxes", "do_nms_sort", "do_nms_obj", "free_detections"} <= set(names)
    lib = ctypes.CDLL(os.path.join(root, "yolo_tensorflow_amd", "libdarknet_hip.so"))
    for n in names:
        assert hasattr(lib, n), n


# every symbol the reference's Python binding resolves from libdarknet.so when it is imported / when detect() runs
# (D2T/darknet.py:49-115: `lib.X` and `X = lib.Y` lines)
DARKNET_PY_SYMBOLS = ["network_width", "network_height", "network_predict", "cuda_set_device", "make_image", "get_network_boxes",
                      "make_network_boxes", "free_detections", "free_ptrs", "reset_rnn", "load_network", "do_nms_obj", "do_nms_sort",
                      "free_image", "letterbox_image", "get_metadata", "load_image_color", "rgbgr_image", "network_predict_image"]


def test_darknet_veneer_exports_everything_darknet_py_binds():
    import ctypes as C
    import re
    path = os.path.join(ROOT, "yolo_tensorflow_amd", "libdarknet_hip.so")
    assert os.path.exists(path), "libdarknet_hip.so is not built"
    lib = C.CDLL(path)
    for name in DARKNET_PY_SYMBOLS:
        assert hasattr(lib, name), "libdarknet_hip.so lacks %s" % name
    hdr = open(os.path.join(ROOT, "include", "darknet_hip.h")).read()
    for name in DARKNET_PY_SYMBOLS + ["free_network", "set_batch_network"]:
        assert re.search(r"\b%s\(" % name, hdr), "include/darknet_hip.h does not declare %s" % name
    ref = "/root/reference/Darknet2Tensorflow/darknet-master/darknet.py"
    if os.path.exists(ref):                       # build container only: the list above is what that file binds
        bound = set(re.findall(r"\blib\.(\w+)", open(ref).read()))
        assert bound == set(DARKNET_PY_SYMBOLS), bound ^ set(DARKNET_PY_SYMBOLS)


def test_draw_detection_counterpart(tmp_path):
    """yolo_tensorflow_amd/draw.py vs the reference's draw_detection rules (V2/utils.py:65-94, D2T/...V3...py:547-582): colour table
    (HSV wheel shuffled with seed 10101 -- restated here with the same stdlib calls), threshold, ratio -> pixel truncation, thickness,
    label text and placement; the rendered picture carries the class colour on the box outline and leaves the input untouched."""
    import colorsys, random
    from yolo_tensorflow_amd import draw
    labels = ["a%d" % i for i in range(20)]
    hsv = [(x / 20.0, 1., 1.) for x in range(20)]
    want = [tuple(int(v * 255) for v in colorsys.hsv_to_rgb(*c)) for c in hsv]
    random.seed(10101); random.shuffle(want); random.seed(None)
    assert draw.class_colors(20) == want
    im = np.full((300, 600, 3), 30, np.uint8)
    boxes = np.array([[0.1, 0.01, 0.5, 0.5], [0.6, 0.4, 0.9, 0.9], [0.2, 0.2, 0.3, 0.3]], np.float32)
    scores = np.array([0.9, 0.5, 0.1], np.float32); cls = np.array([3, 7, 1])
    ov = draw.detection_overlays(im.shape, boxes, scores, cls, labels, thr=0.3, ratio=True)
    assert len(ov) == 2                                              # the 0.1 one is below the threshold
    # (float32 0.01 * 300.0 in double = 2.99999993 -> int 2: the reference's `int(bboxes[i][1] * (1.0 * h))` truncates the same way)
    assert ov[0][0] == (60, 2, 300, 150) and ov[0][1] == want[3] and ov[0][2] == int(900 / 300) // 3
    assert ov[0][3] == "a3: 0.900" and ov[0][4] == (62, 17)          # box touches the top: label goes inside
    assert ov[1][0] == (360, 120, 539, 269) and ov[1][4] == (360, 110)       # (float32 0.9 is a hair below 0.9: truncation, as the reference)
    px = draw.detection_overlays(im.shape, [[60, 30, 300, 150]], [0.8], [2], labels)          # pixel boxes (V2 flavour): full thickness
    assert px[0][0] == (60, 30, 300, 150) and px[0][2] == 3
    out = draw.draw_detection(im, boxes, scores, cls, labels, thr=0.3, ratio=True)
    assert out.shape == im.shape and (im == 30).all()
    assert tuple(out[150, 60]) == want[3] and tuple(out[120, 450]) == want[7] and tuple(out[200, 100]) == (30, 30, 30)


def test_real_batch_norm_vectors_fixture():
    """tests/golden/yolov3_bn_real.npz / yolov2_bn_real.npz: the batch-norm vectors the reference itself printed (D2T/log.txt:224-949 and
    :1-222 through DN/parser.c:1176-1228; tools/make_golden.py bn_real).  Shapes follow the topologies, the first numbers are the log's, and
    the stand-in built from them keeps beta / gamma / variance exactly and is a valid weight stream."""
    import numpy as np
    from yolo_tensorflow_amd import darknet_io as IO
    secs = IO.parse_cfg(IO.cfg_text("yolov3"))
    real = IO.bn_real_vectors(secs)
    convs = [s for s in secs[1:] if s["type"] == "convolutional"]
    assert len(real) == len(convs) == 75 and sum(r is not None for r in real) == 72
    for s, r in zip(convs, real):
        assert (r is None) == (int(s.get("batch_normalize", 0)) == 0)
        if r is not None:
            assert all(r[k].shape == (int(s["filters"]),) for k in ("beta", "gamma", "mean", "var", "w_first"))
    r0 = real[0]      # D2T/log.txt:225-231
    np.testing.assert_allclose(r0["beta"][:3], [-4.31688, -0.757808, -2.1098], rtol=1e-6)
    np.testing.assert_allclose(r0["gamma"][:3], [2.6224, 1.35365, 1.62867], rtol=1e-6)
    np.testing.assert_allclose(r0["var"][:2], [0.0816501, 0.141673], rtol=1e-6)
    assert min(r["var"].min() for r in real if r is not None) < 1e-12          # the file has dead channels
    flat = IO.synth_weights(secs, seed=3, stats="real")
    from oracle import yolo_ref as R
    osecs = R.parse_cfg(IO.cfg_text("yolov3"))
    params = R.unflatten_weights(flat, osecs)
    bn = [p for p in params if "var" in p]
    assert len(bn) == 72 and np.array_equal(bn[0]["beta"], r0["beta"]) and np.array_equal(bn[5]["gamma"], [r for r in real if r is not None][5]["gamma"])
    v2 = IO.bn_real_vectors(IO.parse_cfg(IO.cfg_text("yolov2")))
    assert len(v2) == 23 and sum(r is not None for r in v2) == 22 and abs(float(v2[0]["beta"][0]) + 11.1273) < 1e-4      # D2T/log.txt:2
    with pytest.raises(ValueError):
        IO.bn_real_vectors(IO.parse_cfg(IO.cfg_text("yolov3-tiny")))


def test_pair_closure_and_cfg_keys():
    """Mixed fp16 / split-fp16 plans (darknet_io.pair_closure / with_layer_pairs): the wish list is closed under 'a shortcut's operands and a
    concatenation's inputs share one form', layers that move data inherit, heads are never pairs, and the cfg text carries the result."""
    from yolo_tensorflow_amd import darknet_io as IO
    txt = IO.cfg_text("yolov3"); secs = IO.parse_cfg(txt); L = secs[1:]
    pair = IO.pair_closure(secs, set(range(-1, 12)))
    assert [i for i, v in pair.items() if v] == list(range(-1, 12))                 # the first residual stages close on themselves
    assert abs(IO.pair_flop_share(secs, pair) - 0.158) < 2e-3
    # asking for one conv inside a residual stream pulls in the whole stream (every shortcut of the 52 x 52 stage), nothing outside it
    p2 = IO.pair_closure(secs, {14})
    stream = [i for i in range(12, 37) if L[i]["type"] == "shortcut"]
    assert all(p2[i] for i in stream) and p2[12] and not p2[11] and not p2[13] and not p2[37]
    for i, s in enumerate(L):
        if s["type"] == "shortcut":
            f = int(s["from"]); f = f if f >= 0 else i + f
            assert p2[i] == p2[i - 1] == p2[f]
        if s["type"] == "route" and "," in s["layers"]:
            ins = [int(v) if int(v) >= 0 else i + int(v) for v in s["layers"].split(",")]
            assert len({p2[j] for j in ins} | {p2[i]}) == 1
        if s["type"] == "yolo":
            assert not p2[i] and not p2[i - 1]
    # everything: heads stay fp32
    allp = IO.pair_closure(secs, set(range(-1, len(L))))
    assert sum(allp.values()) == len(L) + 1 - 6
    t2 = IO.with_layer_pairs(txt, pair)
    assert t2.count("yolo_pair=0") == 75 - 9 and "yolo_pair_input" not in t2
    s2 = IO.parse_cfg(t2)
    assert [i for i, s in enumerate(s2[1:]) if s["type"] == "convolutional" and s.get("yolo_pair") != "0"] == [0, 1, 2, 3, 5, 6, 7, 9, 10]
    t3 = IO.with_layer_pairs(txt, IO.pair_closure(secs, set()))
    assert "yolo_pair_input=0" in t3 and t3.count("yolo_pair=0") == 75
    assert IO.with_layer_pairs(t3, allp).count("yolo_pair") == 3                   # (idempotent: old keys are dropped; only the three heads say plain)
