#!/usr/bin/env python3
"""Author the Darknet topology descriptions (`*.cfg`) shipped in yolo_tensorflow_amd/cfg/.

The reference ships no cfg/ directory (SURVEY.md 8c gotcha 1); these are written from the
reference's layer tables and TF graph builders:
  yolov3          V3/yolov3.txt:2-107  == V3/yolo_v3.py:15-44,93-102,195-267
  yolov3-tiny     D2T/YOLO_V3_Tiny_convert_darkenet_to_Tensorflow.py:376-465 (anchors :29)
  yolov2          V2/yolov2.txt:2-32   == V2/model_darknet19_slim.py:119-200 (anchors V2/config.py:7-11)
  yolov2-tiny-voc D2T/YOLO_V2_Tiny_Voc_convert_darkenet_to_Tensorflow.py:162-225
  yolov1          V1/YOLO_V1_Inference.py:124-210 (`_build_network`: 24 bias convs, 7x7/2 first, four SAME pools, CHW flatten,
                  FC 50176 -> 512 -> 4096 -> 1470) + :213-270 ([detection]: side 7, 2 boxes, 20 classes, sqrt sizes); the two
                  `yolo_input_*` keys of [net] state its input normalisation (x/255)*2-1 (:67-71) for the HIP planner (darknet ignores them)
The key names are the ones the reference's cfg parser reads (DN/parser.c:177-205 convolutional,
:303-339 yolo, :341-391 region, :471-486 maxpool, :527-545 shortcut, :580-587 upsample,
:589-628 route, reorg :447-459), so the same text drives three consumers: the HIP library's
planner, oracle/yolo_ref.py and the compiled reference in oracle/_ref.

Run:  python tools/make_cfgs.py   (idempotent; output is committed)
"""
import os

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "yolo_tensorflow_amd", "cfg")


class Cfg:
    def __init__(self, size, channels=3, extra=()):
        self.lines = ["[net]", "batch=1", "subdivisions=1", f"width={size}", f"height={size}",
                      f"channels={channels}"] + list(extra) + [""]
        self.n = 0  # next layer index

    def _sec(self, name, **kv):
        self.lines.append(f"[{name}]")
        for k, v in kv.items():
            self.lines.append(f"{k}={v}")
        self.lines.append("")
        self.n += 1
        return self.n - 1

    def conv(self, filters, size, stride=1, bn=True, act="leaky"):
        kv = {}
        if bn:
            kv["batch_normalize"] = 1
        kv.update(filters=filters, size=size, stride=stride, pad=1, activation=act)
        return self._sec("convolutional", **kv)

    def shortcut(self, frm):
        return self._sec("shortcut", **{"from": frm, "activation": "linear"})

    def route(self, *layers):
        return self._sec("route", layers=",".join(str(l) for l in layers))

    def upsample(self, stride=2):
        return self._sec("upsample", stride=stride)

    def maxpool(self, size=2, stride=2):
        return self._sec("maxpool", size=size, stride=stride)

    def reorg(self, stride=2):
        return self._sec("reorg", stride=stride)

    def yolo(self, mask, anchors, classes):
        return self._sec("yolo", mask=",".join(map(str, mask)),
                         anchors=",  ".join(f"{a},{b}" for a, b in anchors),
                         classes=classes, num=len(anchors), jitter=.3, ignore_thresh=.7,
                         truth_thresh=1, random=0)

    def region(self, anchors, classes):
        return self._sec("region", anchors=",  ".join(f"{a},{b}" for a, b in anchors),
                         bias_match=1, classes=classes, coords=4, num=len(anchors), softmax=1,
                         jitter=.3, rescore=1, object_scale=5, noobject_scale=1, class_scale=1,
                         coord_scale=1, absolute=1, thresh=.6, random=0)

    def connected(self, output, act="leaky"):
        return self._sec("connected", output=output, activation=act)

    def dropout(self, p=.5):
        return self._sec("dropout", probability=p)

    def detection(self, classes=20, side=7, num=2):
        return self._sec("detection", classes=classes, coords=4, rescore=1, side=side, num=num, softmax=0, sqrt=1,
                         jitter=.2, object_scale=1, noobject_scale=.5, class_scale=1, coord_scale=5)

    def text(self):
        return "\n".join(self.lines)


V3_ANCHORS = [(10, 13), (16, 30), (33, 23), (30, 61), (62, 45), (59, 119), (116, 90), (156, 198), (373, 326)]
V3_TINY_ANCHORS = [(10, 14), (23, 27), (37, 58), (81, 82), (135, 169), (344, 319)]
V2_ANCHORS = [(0.57273, 0.677385), (1.87446, 2.06253), (3.33843, 5.47434), (7.88282, 3.52778), (9.77052, 9.16828)]
V2_TINY_VOC_ANCHORS = [(1.08, 1.19), (3.42, 4.41), (6.63, 11.38), (9.42, 5.11), (16.62, 10.52)]


def yolov3(size=416, classes=80):
    c = Cfg(size)
    c.conv(32, 3)

    def stage(filters, blocks):
        c.conv(filters, 3, stride=2)
        for _ in range(blocks):
            c.conv(filters // 2, 1)
            c.conv(filters, 3)
            c.shortcut(-3)
    stage(64, 1); stage(128, 2); stage(256, 8)
    route_1 = c.n - 1           # 36
    stage(512, 8)
    route_2 = c.n - 1           # 61
    stage(1024, 4)

    def yolo_block(f):
        for _ in range(2):
            c.conv(f, 1); c.conv(2 * f, 3)
        r = c.conv(f, 1)
        c.conv(2 * f, 3)
        return r
    nout = 3 * (5 + classes)
    r = yolo_block(512)
    c.conv(nout, 1, bn=False, act="linear"); c.yolo([6, 7, 8], V3_ANCHORS, classes)
    c.route(r); c.conv(256, 1); up = c.upsample(); c.route(up, route_2)
    r = yolo_block(256)
    c.conv(nout, 1, bn=False, act="linear"); c.yolo([3, 4, 5], V3_ANCHORS, classes)
    c.route(r); c.conv(128, 1); up = c.upsample(); c.route(up, route_1)
    yolo_block(128)
    c.conv(nout, 1, bn=False, act="linear"); c.yolo([0, 1, 2], V3_ANCHORS, classes)
    return c.text()


def yolov3_tiny(size=416, classes=80):
    c = Cfg(size)
    for f in (16, 32, 64, 128):
        c.conv(f, 3); c.maxpool()
    route_1 = c.conv(256, 3)    # 8
    c.maxpool()
    c.conv(512, 3)
    c.maxpool(2, 1)             # 'SAME' stride-1 pool (D2T V3_Tiny :445)
    c.conv(1024, 3)
    route_2 = c.conv(256, 1)    # 13
    c.conv(512, 3)
    nout = 3 * (5 + classes)
    c.conv(nout, 1, bn=False, act="linear"); c.yolo([3, 4, 5], V3_TINY_ANCHORS, classes)
    c.route(route_2); c.conv(128, 1); up = c.upsample(); c.route(up, route_1)
    c.conv(256, 3)
    c.conv(nout, 1, bn=False, act="linear"); c.yolo([0, 1, 2], V3_TINY_ANCHORS, classes)
    return c.text()


def yolov2(size=416, classes=80):
    c = Cfg(size)
    c.conv(32, 3); c.maxpool()
    c.conv(64, 3); c.maxpool()
    c.conv(128, 3); c.conv(64, 1); c.conv(128, 3); c.maxpool()
    c.conv(256, 3); c.conv(128, 1); c.conv(256, 3); c.maxpool()
    c.conv(512, 3); c.conv(256, 1); c.conv(512, 3); c.conv(256, 1)
    sc = c.conv(512, 3)         # 16
    c.maxpool()
    c.conv(1024, 3); c.conv(512, 1); c.conv(1024, 3); c.conv(512, 1); c.conv(1024, 3)
    c.conv(1024, 3)
    main = c.conv(1024, 3)      # 24
    c.route(sc)
    c.conv(64, 1)
    ro = c.reorg()
    c.route(ro, main)
    c.conv(1024, 3)
    c.conv(len(V2_ANCHORS) * (5 + classes), 1, bn=False, act="linear")
    c.region(V2_ANCHORS, classes)
    return c.text()


def yolov2_tiny_voc(size=416, classes=20):
    c = Cfg(size)
    for f in (16, 32, 64, 128, 256):
        c.conv(f, 3); c.maxpool()
    c.conv(512, 3); c.maxpool(2, 1)
    c.conv(1024, 3)
    c.conv(1024, 3)
    c.conv(len(V2_TINY_VOC_ANCHORS) * (5 + classes), 1, bn=False, act="linear")
    c.region(V2_TINY_VOC_ANCHORS, classes)
    return c.text()


def yolov1(size=448, classes=20):
    c = Cfg(size, extra=("yolo_input_mul=2", "yolo_input_add=-1"))
    b = dict(bn=False)
    c.conv(64, 7, stride=2, **b); c.maxpool()
    c.conv(192, 3, **b); c.maxpool()
    c.conv(128, 1, **b); c.conv(256, 3, **b); c.conv(256, 1, **b); c.conv(512, 3, **b); c.maxpool()
    for _ in range(4):
        c.conv(256, 1, **b); c.conv(512, 3, **b)
    c.conv(512, 1, **b); c.conv(1024, 3, **b); c.maxpool()
    for _ in range(2):
        c.conv(512, 1, **b); c.conv(1024, 3, **b)
    c.conv(1024, 3, **b); c.conv(1024, 3, stride=2, **b); c.conv(1024, 3, **b); c.conv(1024, 3, **b)
    c.connected(512); c.connected(4096); c.dropout(); c.connected(7 * 7 * (classes + 2 * 5), act="linear")
    c.detection(classes, 7, 2)
    return c.text()


def yolov1_tiny(size=448, classes=20):
    """D2T/YOLO_V1_Tiny_convert_darkenet_to_Tensorflow.py:256-322 `_build_network`: eight BN + leaky 3x3 convs (16 .. 1024, 256), a
    2x2 max-pool after each of the first six, the CHW flatten and one fully connected layer of S*S*(C + 5 B) = 1470 linear outputs;
    input x / 255 (`_input_process`, :212-216)."""
    c = Cfg(size)
    for f in (16, 32, 64, 128, 256, 512):
        c.conv(f, 3); c.maxpool()
    c.conv(1024, 3); c.conv(256, 3)
    c.connected(7 * 7 * (classes + 2 * 5), act="linear")
    c.detection(classes, 7, 2)
    return c.text()


def main():
    os.makedirs(OUT, exist_ok=True)
    files = {
        "yolov1.cfg": yolov1(448), "yolov1-tiny.cfg": yolov1_tiny(448),
        "yolov3.cfg": yolov3(416), "yolov3-608.cfg": yolov3(608),
        "yolov3-tiny.cfg": yolov3_tiny(416),
        "yolov2.cfg": yolov2(416), "yolov2-tiny-voc.cfg": yolov2_tiny_voc(416),
    }
    for name, text in files.items():
        with open(os.path.join(OUT, name), "w") as f:
            f.write("# generated by tools/make_cfgs.py -- do not edit\n" + text)
        print("wrote", name)


if __name__ == "__main__":
    main()
