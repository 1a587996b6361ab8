#!/usr/bin/env python3
"""Generate the committed golden vectors under tests/golden/ FROM THE REFERENCE ITSELF.

Runs only in the build container (it reads /root/reference and oracle/_ref); the outputs are small
`.npz` data files (inputs + expected outputs), never reference source text.

  nms_v3_numpy.npz   inputs/outputs of the reference's numpy `non_max_suppression` + `_iou`
                     (V3/yolo_v3.py:350-420), imported with stub `tensorflow` modules
  v2_postprocess.npz inputs/outputs of V2 `postprocess` / `bboxes_iou` (V2/utils.py:30-187), imported
                     with stub `cv2` (and np.bool shim)
  mini_v3.npz / mini_v2.npz
                     a small darknet topology covering every hot-path layer type, its synthetic weights,
                     an input image, EVERY layer output, and the boxes after get_network_boxes /
                     do_nms_sort -- produced by the reference's own C code compiled CPU-only
                     (oracle/Makefile -> oracle/_ref/libdarknet_ref.so)
  yolov3_bn_real.npz / yolov2_bn_real.npz
                     the COMPLETE batch-norm vectors (beta, gamma, rolling mean, rolling variance) of every
                     batch-normalised conv of yolov3.weights / yolov2.weights, and the first l.n filter
                     values of every conv, as the reference's own load_convolutional_weights printed them
                     (DN/parser.c:1176-1228) into D2T/log.txt:224-949 / :1-222 -- numbers, not source
"""
import os
import sys
import types
import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")

MINI_V3 = """[net]
batch=1
width=64
height=64
channels=3

[convolutional]
batch_normalize=1
filters=8
size=3
stride=1
pad=1
activation=leaky

[convolutional]
batch_normalize=1
filters=16
size=3
stride=2
pad=1
activation=leaky

[convolutional]
batch_normalize=1
filters=8
size=1
stride=1
pad=1
activation=leaky

[convolutional]
batch_normalize=1
filters=16
size=3
stride=1
pad=1
activation=leaky

[shortcut]
from=-3
activation=linear

[convolutional]
batch_normalize=1
filters=32
size=3
stride=2
pad=1
activation=leaky

[maxpool]
size=2
stride=2

[convolutional]
batch_normalize=1
filters=32
size=3
stride=1
pad=1
activation=leaky

[maxpool]
size=2
stride=1

[convolutional]
batch_normalize=1
filters=16
size=1
stride=1
pad=1
activation=leaky

[convolutional]
size=1
stride=1
pad=1
filters=27
activation=linear

[yolo]
mask=3,4,5
anchors=4,5,  8,6,  10,14,  20,18,  30,40,  50,44
classes=4
num=6

[route]
layers=-3

[convolutional]
batch_normalize=1
filters=8
size=1
stride=1
pad=1
activation=leaky

[upsample]
stride=2

[route]
layers=-1,5

[convolutional]
batch_normalize=1
filters=24
size=3
stride=1
pad=1
activation=leaky

[convolutional]
size=1
stride=1
pad=1
filters=27
activation=linear

[yolo]
mask=0,1,2
anchors=4,5,  8,6,  10,14,  20,18,  30,40,  50,44
classes=4
num=6
"""

MINI_V2 = """[net]
batch=1
width=64
height=64
channels=3

[convolutional]
batch_normalize=1
filters=8
size=3
stride=1
pad=1
activation=leaky

[maxpool]
size=2
stride=2

[convolutional]
batch_normalize=1
filters=16
size=3
stride=1
pad=1
activation=leaky

[maxpool]
size=2
stride=2

[convolutional]
batch_normalize=1
filters=16
size=3
stride=1
pad=1
activation=leaky

[maxpool]
size=2
stride=2

[convolutional]
batch_normalize=1
filters=32
size=3
stride=1
pad=1
activation=leaky

[route]
layers=-4

[convolutional]
batch_normalize=1
filters=8
size=1
stride=1
pad=1
activation=leaky

[reorg]
stride=2

[route]
layers=-1,-4

[convolutional]
batch_normalize=1
filters=32
size=3
stride=1
pad=1
activation=leaky

[convolutional]
size=1
stride=1
pad=1
filters=30
activation=linear

[region]
anchors=0.6,0.7,  1.9,2.1,  3.3,5.5
bias_match=1
classes=5
coords=4
num=3
softmax=1
"""


def stub_modules():
    tf = types.ModuleType("tensorflow")
    contrib = types.ModuleType("tensorflow.contrib")
    slim = types.ModuleType("tensorflow.contrib.slim")
    fw = types.ModuleType("tensorflow.contrib.framework")
    fw.add_arg_scope = lambda f: f
    contrib.slim = slim; contrib.framework = fw; tf.contrib = contrib
    sys.modules.update({"tensorflow": tf, "tensorflow.contrib": contrib, "tensorflow.contrib.slim": slim,
                        "tensorflow.contrib.framework": fw, "cv2": types.ModuleType("cv2")})
    if not hasattr(np, "bool"):
        np.bool = bool


def planted_detections(rng, n_img, rows, classes, clusters=6, per=7, size=416.0):
    """[n,rows,5+C] with clusters of jittered boxes (corners, pixels) so NMS has real work."""
    det = np.zeros((n_img, rows, 5 + classes), dtype=np.float32)
    det[..., :4] = rng.uniform(0, size, (n_img, rows, 4)).astype(np.float32)
    det[..., 4] = rng.uniform(0.0, 0.3, (n_img, rows)).astype(np.float32)
    det[..., 5:] = rng.uniform(0.01, 1, (n_img, rows, classes)).astype(np.float32)
    for b in range(n_img):
        r = 0
        for c in range(clusters):
            cx, cy = rng.uniform(60, size - 60, 2); w, h = rng.uniform(30, 120, 2)
            cls = int(rng.integers(0, classes))
            for _ in range(per):
                j = rng.normal(0, 6, 4)
                det[b, r, :4] = [cx - w / 2 + j[0], cy - h / 2 + j[1], cx + w / 2 + j[2], cy + h / 2 + j[3]]
                det[b, r, 4] = rng.uniform(0.55, 0.999)
                det[b, r, 5:] = rng.uniform(0.01, 0.3, classes); det[b, r, 5 + cls] = rng.uniform(0.7, 1.0)
                r += 1
    perm = rng.permutation(rows)
    return det[:, perm]


def gen_nms_v3():
    sys.path.insert(0, os.path.join(REF, "YOLO_V3", "YOLOv3-Tensorflow-detect-export"))
    import yolo_v3 as ref
    rng = np.random.default_rng(2)
    det = planted_detections(rng, 2, 300, 6)
    res = ref.non_max_suppression(det, confidence_threshold=0.5, iou_threshold=0.4)
    keys = sorted(res.keys())
    boxes = [np.array([b for b, _ in res[k]], dtype=np.float32) for k in keys]
    scores = [np.array([s for _, s in res[k]], dtype=np.float32) for k in keys]
    pairs = rng.uniform(0, 100, (64, 2, 4)).astype(np.float32)
    ious = np.array([ref._iou(p[0], p[1]) for p in pairs], dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "nms_v3_numpy.npz"), det=det, conf=0.5, iou=0.4, classes=np.array(keys),
                        counts=np.array([len(b) for b in boxes]), boxes=np.concatenate(boxes), scores=np.concatenate(scores),
                        pairs=pairs, pair_ious=ious)
    print("nms_v3_numpy:", {int(k): len(res[k]) for k in keys})


def gen_v2_post():
    sys.path.insert(0, os.path.join(REF, "YOLO_V2", "YOLOv2-Tensorflow-detect-export"))
    cwd = os.getcwd()
    os.chdir(os.path.join(REF, "YOLO_V2", "YOLOv2-Tensorflow-detect-export"))   # config.py reads ./yolo2_data at import
    import utils as ref
    os.chdir(cwd)
    rng = np.random.default_rng(3)
    n = 13 * 13 * 5
    # normalised corner boxes with planted overlapping clusters
    cx = rng.uniform(0.1, 0.9, n); cy = rng.uniform(0.1, 0.9, n); w = rng.uniform(0.02, 0.5, n); h = rng.uniform(0.02, 0.5, n)
    for c in range(8):
        base = c * 9
        cx[base:base + 9] = cx[base] + rng.normal(0, .01, 9); cy[base:base + 9] = cy[base] + rng.normal(0, .01, 9)
        w[base:base + 9] = w[base] + rng.normal(0, .01, 9); h[base:base + 9] = h[base] + rng.normal(0, .01, 9)
    bboxes = np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], -1).astype(np.float32).reshape(1, 169, 5, 4)
    obj = rng.uniform(0, 0.4, n).astype(np.float32); obj[:72] = rng.uniform(0.7, 1, 72)
    cls = rng.dirichlet(np.ones(80) * 0.05, n).astype(np.float32)
    for c in range(8):
        v = rng.uniform(0.0, 0.01, 80); v[int(rng.integers(0, 80))] = 0.9
        cls[c * 9:(c + 1) * 9] = (v / v.sum()).astype(np.float32)
    # two overlapping clusters of DIFFERENT classes must both survive (V2/utils.py:183)
    cx[72:81] = cx[0] + rng.normal(0, .01, 9); cy[72:81] = cy[0] + rng.normal(0, .01, 9)
    w[72:81] = w[0]; h[72:81] = h[0]; obj[72:81] = rng.uniform(0.7, 1, 9)
    v = rng.uniform(0.0, 0.01, 80); v[(int(np.argmax(cls[0])) + 1) % 80] = 0.9
    cls[72:81] = (v / v.sum()).astype(np.float32)
    bboxes = np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], -1).astype(np.float32).reshape(1, 169, 5, 4)
    out_b, out_s, out_c = ref.postprocess(bboxes.copy(), obj.reshape(1, 169, 5).copy(), cls.reshape(1, 169, 5, 80).copy(),
                                          image_shape=(576, 768), threshold=0.5)
    ib = rng.integers(0, 400, (40, 4)).astype(np.int32); ib[:, 2:] += ib[:, :2]
    with np.errstate(all="ignore"):
        pair_iou = ref.bboxes_iou(ib[0], ib[1:])
    np.savez_compressed(os.path.join(OUT, "v2_postprocess.npz"), bboxes=bboxes, obj=obj.reshape(1, 169, 5),
                        cls=cls.reshape(1, 169, 5, 80), image_shape=np.array([576, 768]), threshold=0.5,
                        out_boxes=out_b, out_scores=out_s, out_classes=out_c, int_boxes=ib, int_ious=pair_iou)
    print("v2_postprocess: kept", len(out_s))


MINI_V1 = """[net]
batch=1
width=64
height=64
channels=3

[convolutional]
filters=16
size=7
stride=2
pad=1
activation=leaky

[maxpool]
size=2
stride=2

[convolutional]
filters=32
size=3
stride=1
pad=1
activation=leaky

[maxpool]
size=2
stride=2

[convolutional]
filters=64
size=3
stride=2
pad=1
activation=leaky

[connected]
output=96
activation=leaky

[dropout]
probability=.5

[connected]
output=270
activation=linear

[detection]
classes=20
coords=4
rescore=1
side=3
num=2
softmax=0
sqrt=1
"""


def gen_mini_v1():
    """YOLOv1-style topology (7x7/2 bias conv, SAME pools, [connected] x2 with a [dropout] between, [detection]) through the
    compiled reference: every layer output, and get_network_boxes -> get_detection_detections (DN/detection_layer.c:225-254).
    The input is already in the reference's (x/255)*2-1 range (V1/YOLO_V1_Inference.py:67-71)."""
    from oracle import darknet_ref as D
    from yolo_tensorflow_amd import darknet_io as IO
    secs = IO.parse_cfg(MINI_V1)
    flat = IO.synth_weights(secs, seed=9)
    net = D.RefNet(MINI_V1, flat, 0, 1)
    rng = np.random.default_rng(13)
    img = rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)
    x = (img.astype(np.float32) / np.float32(255.0)) * np.float32(2) - np.float32(1)
    net.predict(x)
    data = {"cfg": np.array(MINI_V1), "weights": flat, "image_u8": img, "header": np.array([0, 1])}
    for i in range(net.n):
        data["layer_%02d" % i] = net.layer_output_nhwc(i).astype(np.float32)
    thresh = 0.2
    bb, obj, pr = net.boxes(thresh, None, 20)
    data["boxes_raw"], data["obj_raw"], data["prob_raw"] = bb, obj, pr
    data["thresh"] = np.float32(thresh)
    np.savez_compressed(os.path.join(OUT, "mini_v1.npz"), **data)
    print("mini_v1 layers", net.n, "boxes", len(bb), "nonzero probs", int((pr > 0).sum()))
    net.close()


MINI_LOCAL = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "mini_local.cfg")).read()


def gen_mini_local():
    """[local] layers (locally connected: unshared filters per output location, DN/local_layer.c:91-120; darknet's own yolov1.cfg has one
    between its last conv and the fully connected head): a same-size 3x3 / pad 1 one and a 2x2 / stride 2 / unpadded one, then
    [connected] + [detection], through the compiled reference.  (The first conv is there for its workspace: the reference sizes a
    [local] layer's im2col workspace in elements, not bytes -- DN/local_layer.c:64 -- and relies on an earlier conv's being larger.)"""
    from oracle import darknet_ref as D
    from yolo_tensorflow_amd import darknet_io as IO
    secs = IO.parse_cfg(MINI_LOCAL)
    flat = IO.synth_weights(secs, seed=2)
    net = D.RefNet(MINI_LOCAL, flat, 0, 1)
    img = np.random.default_rng(17).integers(0, 256, (48, 48, 3), dtype=np.uint8)
    net.predict(img.astype(np.float32) / np.float32(255.0))
    data = {"cfg": np.array(MINI_LOCAL), "weights": flat, "image_u8": img, "header": np.array([0, 1])}
    for i in range(net.n):
        data["layer_%02d" % i] = net.layer_output_nhwc(i).astype(np.float32)
    bb, obj, pr = net.boxes(0.2, None, 2)
    data["boxes_raw"], data["obj_raw"], data["prob_raw"] = bb, obj, pr
    data["thresh"] = np.float32(0.2)
    np.savez_compressed(os.path.join(OUT, "mini_local.npz"), **data)
    print("mini_local layers", net.n, "boxes", len(bb))
    net.close()


def gen_mini(name, cfg, classes, nms_thresh=0.3, thresh=0.15):
    from oracle import darknet_ref as D
    from yolo_tensorflow_amd import darknet_io as IO
    secs = IO.parse_cfg(cfg)
    flat = IO.synth_weights(secs, seed=7, obj_bias=0.5)
    mj, mn = IO.default_header(secs)
    net = D.RefNet(cfg, flat, mj, mn)
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)
    x = img.astype(np.float32) / np.float32(255.0)
    net.predict(x)
    data = {"cfg": np.array(cfg), "weights": flat, "image_u8": img, "header": np.array([mj, mn])}
    for i in range(net.n):
        data["layer_%02d" % i] = net.layer_output_nhwc(i).astype(np.float32)
    bb, obj, pr = net.boxes(thresh, None, classes)
    data["boxes_raw"], data["obj_raw"], data["prob_raw"] = bb, obj, pr
    net.predict(x)
    bb2, obj2, pr2 = net.boxes(thresh, nms_thresh, classes)
    data["boxes_nms"], data["obj_nms"], data["prob_nms"] = bb2, obj2, pr2
    data["thresh"] = np.float32(thresh); data["nms"] = np.float32(nms_thresh)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **data)
    print(name, "layers", net.n, "boxes", len(bb), "nonzero probs before/after nms", int((pr > 0).sum()), int((pr2 > 0).sum()))
    net.close()


def gen_bn_real():
    """D2T/log.txt is the reference's stdout of two detect runs (yolov2 then yolov3) with the printf block of DN/parser.c:1176-1228
    enabled: per batch-normalised conv five lines of numbers (beta, gamma, rolling mean, rolling variance -- l.n values each -- and the
    first l.n of the l.nweights filter values), per plain conv (the heads) the filter line only; line 223 / 950 are the detections."""
    import re
    lines = open(os.path.join(REF, "Darknet2Tensorflow", "darknet-master", "log.txt")).read().split("\n")
    for name, lo, hi, n_bn, n_conv in (("yolov2", 1, 222, 22, 23), ("yolov3", 224, 949, 72, 75)):
        convs = []; cur = None; k = lo - 1
        while k < hi:
            h = lines[k]
            m = re.match(r"\*+robin#convolutional_(\w+?)(?:/\w+)?\((?:l\.n|num)=(\d+)\)", h)
            assert m, (k + 1, h[:80])
            vals = np.array([float(v) for v in lines[k + 1].replace(" ", "").split(",") if v], dtype=np.float32)
            what, cnt = m.group(1), int(m.group(2))
            if what == "biases":
                cur = {"n": cnt, "beta": vals}
            elif what in ("scales", "rolling_mean", "rolling_variance"):
                cur[{"scales": "gamma", "rolling_mean": "mean", "rolling_variance": "var"}[what]] = vals
            else:
                assert what == "weights"
                c = cur if cur is not None else {"n": len(vals)}
                c["nweights"] = cnt; c["w_first"] = vals
                assert all(len(c[q]) == c["n"] for q in ("beta", "gamma", "mean", "var") if q in c) and len(vals) == c["n"]
                convs.append(c); cur = None
            k += 2
        assert len(convs) == n_conv and sum("beta" in c for c in convs) == n_bn, (len(convs), name)
        data = {"n_conv": np.int32(len(convs)), "filters": np.array([c["n"] for c in convs], np.int32),
                "nweights": np.array([c["nweights"] for c in convs], np.int64), "bn": np.array(["beta" in c for c in convs])}
        for i, c in enumerate(convs):
            for q in ("beta", "gamma", "mean", "var", "w_first"):
                if q in c:
                    data["%s_%d" % (q, i)] = c[q]
        np.savez_compressed(os.path.join(OUT, name + "_bn_real.npz"), **data)
        bn = [c for c in convs if "beta" in c]
        print(name + "_bn_real: %d convs (%d batch-normalised), gamma %.4g..%.4g (%.1f%% negative), beta %.3g..%.3g, mean %.3g..%.3g, var %.3g..%.3g"
              % (len(convs), len(bn), min(c["gamma"].min() for c in bn), max(c["gamma"].max() for c in bn),
                 100.0 * sum((c["gamma"] < 0).sum() for c in bn) / sum(c["n"] for c in bn),
                 min(c["beta"].min() for c in bn), max(c["beta"].max() for c in bn), min(c["mean"].min() for c in bn), max(c["mean"].max() for c in bn),
                 min(c["var"].min() for c in bn), max(c["var"].max() for c in bn)))


def gen_known_answers():
    """The reference's only recorded END-TO-END results: the detections its darknet binding printed for dog.jpg with the genuine
    yolov2.weights (D2T/log.txt:223) and yolov3.weights (:950) -- `detect()` of D2T/darknet.py:125-142, thresh .5, nms .45: a list of
    (class name, probability, (cx, cy, w, h) in pixels of the 768 x 576 image).  Numbers only -> tests/golden/dog_known_answers.json; they
    become checkable the day somebody supplies the files (tests/test_gpu_real_weights.py, YOLO_REAL_WEIGHTS / YOLO_REAL_WEIGHTS_V2)."""
    import ast, json
    lines = open(os.path.join(REF, "Darknet2Tensorflow", "darknet-master", "log.txt")).read().split("\n")
    out = {"image": "dog.jpg", "image_size_wh": [768, 576], "call": "darknet.py detect(net, meta, image, thresh=.5, hier_thresh=.5, nms=.45)",
           "coco_index": {"bicycle": 1, "truck": 7, "dog": 16}}
    for name, ln in (("yolov2", 223), ("yolov3", 950)):
        dets = ast.literal_eval(lines[ln - 1])
        out[name] = {"log_line": ln, "detections": [{"name": n, "prob": p, "box_cxcywh": list(b)} for n, p, b in dets]}
        print(name, [(n, round(p, 4)) for n, p, _ in dets])
    json.dump(out, open(os.path.join(OUT, "dog_known_answers.json"), "w"), indent=1)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    if sys.argv[1:] == ["bn_real"]:
        gen_bn_real(); sys.exit(0)
    if sys.argv[1:] == ["known_answers"]:
        gen_known_answers(); sys.exit(0)
    stub_modules()
    gen_nms_v3()
    gen_v2_post()
    gen_mini("mini_v3", MINI_V3, 4)
    gen_mini("mini_v2", MINI_V2, 5)
    gen_mini_v1()
    gen_mini_local()
    gen_bn_real()
    gen_known_answers()
