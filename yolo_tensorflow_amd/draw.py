"""Thin counterpart of the reference's drawing helpers -- `draw_detection` (V2/utils.py:65-94 pixel boxes; the converter classes' method
D2T/YOLO_V3_convert_darkenet_to_Tensorflow.py:547-582 and V2/utils.py `_draw_detection` for ratio boxes) -- with PIL instead of OpenCV.
Outside the inference hot path: nothing here touches the device.  What is kept from the reference is what a caller can observe in the
picture: the per-class colour table (HSV wheel, shuffled with the fixed seed 10101), the score threshold, the ratio -> pixel truncation,
the line thickness rule `int((h + w) / 300)` (a third of it for ratio boxes), the label text '%s: %.3f' and where it goes (inside the box
when the box touches the top edge).  The glyphs are PIL's default font, not OpenCV's Hershey face."""
import colorsys
import random

import numpy as np


def class_colors(n):
    """The reference's colour table for n classes (V2/utils.py:67-74): evenly spaced hues, shuffled with seed 10101."""
    hsv = [(x / float(n), 1., 1.) for x in range(n)]
    colors = [colorsys.hsv_to_rgb(*c) for c in hsv]
    colors = [(int(c[0] * 255), int(c[1] * 255), int(c[2] * 255)) for c in colors]
    state = random.getstate()
    random.seed(10101)
    random.shuffle(colors)
    random.setstate(state)              # (the reference re-seeds from the clock; restoring the caller's state is kinder)
    return colors


def detection_overlays(shape, bboxes, scores, cls_inds, labels, thr=0.3, ratio=False):
    """What would be drawn, as data: [(box (x0, y0, x1, y1) int pixels, colour, thickness, text, text_loc)] for every detection with
    score >= thr.  ratio=True: boxes are fractions of the image (the V3 converter's `detected_boxes`), truncated to int pixels as the
    reference does; the outline is a third as thick there (D2T/...V3...py:572 `thick//3`)."""
    h, w = int(shape[0]), int(shape[1])
    b = np.array(bboxes, dtype=np.float64).reshape(-1, 4)
    if ratio:
        b = np.stack([(b[:, 0] * (1.0 * w)).astype(np.int64), (b[:, 1] * (1.0 * h)).astype(np.int64),
                      (b[:, 2] * (1.0 * w)).astype(np.int64), (b[:, 3] * (1.0 * h)).astype(np.int64)], -1)
    b = b.astype(np.int32)
    colors = class_colors(len(labels))
    thick = int((h + w) / 300)
    out = []
    for i, box in enumerate(b):
        if scores[i] < thr:
            continue
        k = int(cls_inds[i])
        name = labels[k] if not isinstance(labels, dict) else labels.get(k, str(k))
        text_loc = (int(box[0]) + 2, int(box[1]) + 15) if box[1] < 20 else (int(box[0]), int(box[1]) - 10)
        out.append((tuple(int(v) for v in box), colors[k], thick // 3 if ratio else thick, "%s: %.3f" % (name, scores[i]), text_loc))
    return out


def draw_detection(im, bboxes, scores, cls_inds, labels, thr=0.3, ratio=False):
    """im: uint8 [h, w, 3] (any channel order: colours are applied as given).  Returns a copy with boxes and labels drawn."""
    from PIL import Image, ImageDraw
    arr = np.ascontiguousarray(np.asarray(im, dtype=np.uint8))
    img = Image.fromarray(arr.copy())
    d = ImageDraw.Draw(img)
    for box, color, thick, text, loc in detection_overlays(arr.shape, bboxes, scores, cls_inds, labels, thr, ratio):
        x0, y0, x1, y1 = box
        d.rectangle([min(x0, x1), min(y0, y1), max(x0, x1), max(y0, y1)], outline=color, width=max(1, thick))
        d.text((loc[0], loc[1] - 10), text, fill=(255, 255, 255) if not ratio else (0, 0, 255))      # cv2 anchors text at its baseline, PIL at its top
    return np.asarray(img)
