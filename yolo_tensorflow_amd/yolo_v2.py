"""Counterpart of the reference's YOLOv2 modules: `model_darknet19_slim.build_network`, `postprocess.decode` /
`decode.decode`, `utils.preprocess_image` / `utils.postprocess`, `config.anchors` (V2/*.py).  Same names and
argument meaning; the darknet-19 graph, the region decode and both NMS flavours run in libyolo_hip.so."""
import numpy as np
from . import hip, darknet_io as IO

anchors = [[0.57273, 0.677385], [1.87446, 2.06253], [3.33843, 5.47434], [7.88282, 3.52778], [9.77052, 9.16828]]  # V2/config.py:7-11


class Model:
    def __init__(self, cfg="yolov2", size=416, max_batch=1, dtype=hip.FP32, device=0, weights_file=None, weights=None, semantics=hip.SEM_TF):
        text = IO.cfg_text(cfg)
        if size != int(IO.parse_cfg(text)[0]["width"]):
            text = IO.with_input_size(text, size)
        self.cfg_text, self.size = text, size
        self.engine = hip.Engine(text, max_batch=max_batch, dtype=dtype, semantics=semantics, device=device)
        if weights_file is not None:
            self.engine.load_weights(weights_file)
        elif weights is not None:
            self.engine.set_weights(weights)

    def close(self):
        self.engine.close()


def preprocess_image(image, image_size=(416, 416), bgr=True, legacy_tf_resize=False):
    """V2/utils.py:13-27, same arguments: `image` is what the reference's callers pass -- the uint8 [H,W,3] array of `cv2.imread`, i.e. BGR
    (V2/Main.py, V2/YOLO_v2.py:49-52) -- and the result is [1, h, w, 3] float32 = cv2.resize(float32(image) as RGB, image_size) / 225.0:
    the colour swap (`cv2.cvtColor(.., COLOR_BGR2RGB)`), OpenCV's INTER_LINEAR rule (half-pixel centres) and the reference's 225.0 (a typo
    for 255, :22; reproduced) all on the device (`yolo_op_resize_cv2`).  image_size is cv2's dsize = (width, height).
    bgr=False: the image is already RGB (PIL / the other modules of this package) -- no swap.
    legacy_tf_resize=True: rounds 1-5's behaviour (TF's legacy bilinear rule, RGB input), what the V3 graph's `_input_process` uses; kept for
    callers that relied on it.  OpenCV is not installed here: the rule is restated from its published source and checked against the
    oracle's closed form (`oracle.resize_cv2_linear`), not against cv2 itself."""
    if legacy_tf_resize:
        if image_size[0] != image_size[1]:
            raise hip.YoloError("legacy rule: square network input only")
        # (value/255 -> legacy bilinear) * 255/225 on the device == resize(value)/225 up to rounding
        return hip.op_resize_u8(np.ascontiguousarray(image, dtype=np.uint8), image_size[0], post_scale=255.0 / 225.0)[None]
    return hip.op_resize_cv2(image, (image_size[1], image_size[0]), swap_rb=bgr, divisor=225.0)[None]


def build_network(images, num_outputs=425, alpha=0.1, keep_prob=0.5, is_training=False, scope='yolov2', model=None, fused=False):
    """V2/model_darknet19_slim.py:119-200: images [N,416,416,3] already normalised -> the RAW head tensor [N,13,13,num_outputs]
    (the `logits` of :198-200), which `decode` consumes and which a caller may inspect or save (the freeze script
    V2/yOLO_v2_export_graph.py does).  fused=True returns the device's fused result instead -- the decoded rows [N, 845, 85] =
    (bx, by, bw, bh, objectness, softmax classes) -- which `decode` accepts as well (one host copy less)."""
    if model is None:
        raise hip.YoloError("pass model=yolo_v2.Model(weights_file=...)")
    images = np.ascontiguousarray(images, dtype=np.float32)
    if fused:
        return model.engine.forward(images, scale=1.0)
    model.engine.forward(images, scale=1.0, want_detections=False)
    raw = model.engine.head_raw(0, images.shape[0])
    if raw.shape[-1] != num_outputs:
        raise hip.YoloError("the topology's head has %d outputs per cell, num_outputs=%d was asked for" % (raw.shape[-1], num_outputs))
    return raw


def _decoded_rows(model_output, output_sizes, num_class, anchors_):
    """raw head [N,H,W,A*(5+C)] -> decoded rows [N, H*W*A, 5+C] with the region decode kernel (V2/decode.py:13-47); rows pass through."""
    d = np.asarray(model_output, dtype=np.float32)
    if d.ndim == 3:
        return d
    if d.ndim != 4 or d.shape[1] != output_sizes[0] or d.shape[2] != output_sizes[1] or d.shape[1] != d.shape[2]:
        raise hip.YoloError("decode: expected the raw head [N,%d,%d,A*(5+C)] or decoded rows [N,rows,5+C]" % tuple(output_sizes))
    a = np.asarray(anchors if anchors_ is None else anchors_, dtype=np.float32)
    if d.shape[3] != len(a) * (5 + num_class):
        raise hip.YoloError("decode: head depth %d does not match %d anchors x (5 + %d)" % (d.shape[3], len(a), num_class))
    return hip.op_decode(d, a, num_class, 32 * output_sizes[0], region=True)


def decode(model_output, output_sizes=(13, 13), num_class=80, threshold=None, iou_threshold=0.5, anchors=None, model=None):
    """Two reference functions share this name:
       V2/decode.py:13      decode(model_output, output_sizes, num_class, anchors) -> (bboxes, obj_probs, class_probs)
       V2/postprocess.py:10 decode(..., threshold=0.5, iou_threshold=0.5, anchors) -> (boxes, scores, classes) after TF NMS (max 10)
    `model_output` is what build_network returned: the raw head [N,13,13,425] (decoded here by the region kernel) or, from
    build_network(fused=True), the already decoded rows.  threshold=None selects the first form."""
    d = _decoded_rows(model_output, output_sizes, num_class, anchors)
    n = d.shape[0]
    H, W = output_sizes
    A = d.shape[1] // (H * W)
    if threshold is None:
        c4 = hip.op_detections_boxes(d).reshape(n, H * W, A, 5 + num_class)      # bbox_x - bbox_w/2 ... on the device
        return np.ascontiguousarray(c4[..., 0:4]), np.ascontiguousarray(c4[..., 4]), np.ascontiguousarray(c4[..., 5:])
    res = hip.op_postprocess(d, threshold, iou_threshold, 10, hip.NMS_TF, hip.SELECT_GE)
    r = res[0] if n == 1 else res
    if n == 1:
        return np.stack([r["x0"], r["y0"], r["x1"], r["y1"]], -1).reshape(-1, 4), r["score"], r["cls"]
    return [np.stack([q["x0"], q["y0"], q["x1"], q["y1"]], -1).reshape(-1, 4) for q in r], [q["score"] for q in r], [q["cls"] for q in r]


def postprocess(bboxes, obj_probs, class_probs, image_shape=(416, 416), threshold=0.5):
    """V2/utils.py:30-62 on the device: scale to the image, int32 cast, clip, score = obj * max class, > threshold,
    top-400, class-aware NMS (0.5).  Inputs as returned by the 3-output `decode`.  -> (bboxes int32, scores, classes)."""
    b = np.asarray(bboxes, dtype=np.float32).reshape(-1, 4)
    o = np.asarray(obj_probs, dtype=np.float32).reshape(-1)
    c = np.asarray(class_probs, dtype=np.float32).reshape(len(o), -1)
    det = np.concatenate([b, o[:, None], c], axis=1)[None]
    r = hip.op_postprocess(det, threshold, 0.5, 400, hip.NMS_PER_CLASS, hip.SELECT_GT, image_hw=image_shape, corners=True)[0]
    return np.stack([r["x0"], r["y0"], r["x1"], r["y1"]], -1).reshape(-1, 4).astype(np.int32), r["score"], r["cls"].astype(np.int64)
