"""Host-side counterpart of the reference's `yolo_v3.py` / `YOLOV3.py` (V3/yolo_v3.py, V3/YOLOV3.py): the same
function names and argument meaning, with the TF graph + `sess.run` replaced by the HIP library.

    detections = yolo_v3(inputs, num_classes, data_format='NHWC')            # V3/yolo_v3.py:195  -> [N,10647,5+C]
    load_ops   = load_weights(model_or_none, weights_file)                   # V3/yolo_v3.py:270
    boxes      = detections_boxes(detections)                                # V3/yolo_v3.py:329
    result     = non_max_suppression(boxes, confidence_threshold, iou_threshold)   # V3/yolo_v3.py:376

Differences forced by leaving TensorFlow: `inputs` is an array (numpy / torch, host or device) instead of a
placeholder and the functions compute eagerly; `load_weights` binds a `.weights` file to a model handle instead
of returning `tf.assign` ops.  All arithmetic runs in libyolo_hip.so -- there is no CPU path here.
"""
import numpy as np
from . import hip, darknet_io as IO

_BATCH_NORM_EPSILON = 1e-05      # folded into the filters at load time (csrc/yolo_pack.cpp pack_conv)
_LEAKY_RELU = 0.1
_ANCHORS = [(10, 13), (16, 30), (33, 23), (30, 61), (62, 45), (59, 119), (116, 90), (156, 198), (373, 326)]


class Model:
    """What `tf.variable_scope('detector')` + the graph were: one planned network and its weights."""

    def __init__(self, cfg="yolov3", size=416, num_classes=80, max_batch=1, dtype=hip.BF16, decode=hip.DECODE_PIXEL,
                 semantics=hip.SEM_TF, device=0, weights_file=None, weights=None, keep_layers=False, stream=None):
        text = IO.cfg_text(cfg)
        if size != int(IO.parse_cfg(text)[0]["width"]):
            text = IO.with_input_size(text, size)
        if num_classes != 80 and cfg.startswith("yolov3"):
            text = _with_classes(text, num_classes)
        self.cfg_text = text
        self.secs = IO.parse_cfg(text)
        self.size, self.num_classes = size, num_classes
        self.engine = hip.Engine(text, max_batch=max_batch, dtype=dtype, semantics=semantics, decode=decode, device=device,
                                 keep_layers=keep_layers, stream=stream)
        if weights_file is not None:
            self.engine.load_weights(weights_file)
        elif weights is not None:
            self.engine.set_weights(weights)

    def close(self):
        self.engine.close()


def _with_classes(text, num_classes):
    out = []
    nout = str(3 * (5 + num_classes))
    lines = text.splitlines()
    for i, line in enumerate(lines):
        key = line.split("=")[0].strip()
        if key == "classes":
            line = "classes=%d" % num_classes
        if key == "filters" and line.split("=")[1].strip() == "255":
            line = "filters=" + nout
        out.append(line)
    return "\n".join(out)


_default = {}


def _model_for(inputs, num_classes, decode):
    n, h, w, c = inputs.shape
    key = (int(h), int(num_classes), decode)
    m = _default.get(key)
    if m is None or m.engine.max_batch < n:
        raise hip.YoloError("no weights are bound for a %dx%d / %d-class network: call load_weights(None, file, size=%d) "
                            "or pass model=Model(...)" % (h, w, num_classes, h))
    return m


def yolo_v3(inputs, num_classes, is_training=False, data_format='NCHW', reuse=False, model=None):
    """V3/yolo_v3.py:195.  inputs: [N,S,S,3] values 0..255 (the `inputs / 255` of :215 happens on the device).
    Returns detections [N, rows, 5+num_classes] = (cx, cy, w, h in input pixels, objectness, class scores),
    concatenated over the three scales in the reference's order (13x13, 26x26, 52x52).
    data_format only selected TF's internal layout in the reference; the input is NHWC either way (:208-212)."""
    if is_training:
        raise hip.YoloError("inference only: is_training=True is outside the hot path")
    m = model or _model_for(inputs, num_classes, hip.DECODE_PIXEL)
    return m.engine.forward(inputs, scale=1.0 / 255.0)


def yolo_v3_with_nms(inputs, num_classes, score_threshold=0.5, iou_threshold=0.5, is_training=False, data_format='NCHW',
                     reuse=False, model=None, max_output_size=20):
    """The 4-output variant V3/YOLOV3.py:274: returns (detections [N,rows,5+C] normalised, bboxes, scores, classes) after
    select-threshold + tf.image.non_max_suppression(max_output_size=20) (:353-379).  The reference flattens the batch in
    `tf.boolean_mask` (image identity is lost for N > 1, row S); here the tail runs per image and lists are returned."""
    if is_training:
        raise hip.YoloError("inference only")
    m = model or _model_for(inputs, num_classes, hip.DECODE_RATIO)
    det = m.engine.forward(inputs, scale=1.0 / 255.0)
    res = m.engine.postprocess(int(inputs.shape[0]), score_thr=score_threshold, iou_thr=iou_threshold, max_out=max_output_size,
                               nms_mode=hip.NMS_TF, select_mode=hip.SELECT_GT)
    boxes = [np.stack([r["x0"], r["y0"], r["x1"], r["y1"]], -1).reshape(-1, 4) for r in res]
    return det, boxes, [r["score"] for r in res], [r["cls"] for r in res]


def load_weights(var_list, weights_file, size=416, num_classes=80, max_batch=32, dtype=hip.BF16, device=0, cfg="yolov3"):
    """V3/yolo_v3.py:270.  `var_list` was the list of TF variables to assign; pass a `Model` to (re)load its weights,
    or None to create and register the default models `yolo_v3()` / `yolo_v3_with_nms()` use.  The file format and
    the per-layer order (beta, gamma, mean, var | bias, then OIHW filters) are the reference's (:277-323)."""
    if isinstance(var_list, Model):
        var_list.engine.load_weights(weights_file)
        return [var_list]
    flat, _ = IO.read_weights_file(weights_file)
    models = []
    for decode in (hip.DECODE_PIXEL, hip.DECODE_RATIO):
        m = Model(cfg, size, num_classes, max_batch, dtype, decode, device=device, weights=flat)
        _default[(size, num_classes, decode)] = m
        models.append(m)
    return models


def detections_boxes(detections):
    """V3/yolo_v3.py:329: (cx, cy, w, h, ...) -> (x0, y0, x1, y1, ...), on the device (k_boxes_to_corners)."""
    d = np.ascontiguousarray(detections, dtype=np.float32)
    squeeze = d.ndim == 2
    out = hip.op_detections_boxes(d[None] if squeeze else d)
    return out[0] if squeeze else out


def non_max_suppression(predictions_with_boxes, confidence_threshold, iou_threshold=0.4, device=0):
    """V3/yolo_v3.py:376 -> dict: class -> [(box[4], score)], shared across the batch like the reference (:388).
    Runs on the device (nms mode YOLO_NMS_NUMPY_V3) and reproduces the reference's behaviours: objectness-only gate,
    class = argmax of the class scores, unclamped `_iou` with +1e-05, strict `iou < threshold` keep, and the shifted
    score indexing of :414-418."""
    p = np.ascontiguousarray(predictions_with_boxes, dtype=np.float32)
    if p.ndim != 3:
        raise hip.YoloError("predictions_with_boxes must be [N, rows, 5+C]")
    res = hip.op_postprocess(p, confidence_threshold, iou_threshold, max_out=p.shape[1], nms_mode=hip.NMS_NUMPY_V3,
                             select_mode=hip.SELECT_GT, corners=True, device=device)
    out = {}
    for r in res:
        for i in range(len(r)):
            out.setdefault(int(r["cls"][i]), []).append(
                (np.array([r["x0"][i], r["y0"][i], r["x1"][i], r["y1"][i]], dtype=np.float32), r["score"][i]))
    return out
