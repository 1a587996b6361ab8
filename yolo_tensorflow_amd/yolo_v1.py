"""Host-side counterpart of the reference's YOLOv1 script (V1/YOLO_V1_Inference.py): class `Yolo(weights_file, input_image=None,
verbose=True)` with `detect_from_file` / `_detect_from_image`, the TF-Slim graph + `sess.run` (:32-70, :371-392) replaced by the HIP
library: the 24-conv + 3-FC network (`_build_network`, :124-210) is the shipped topology `cfg/yolov1.cfg`, the input normalisation
`(x / 255) * 2 - 1` + `tf.image.resize_images` (:67-71) runs on the device (yolo_forward_image_u8), `_build_detector` (:213-270) is
the [detection] decode kernel + `YOLO_NMS_TF_V1` (including the reference's swapped width/height in the NMS boxes, :259-262).

Differences forced by leaving TensorFlow / OpenCV: `weights_file` is a Darknet `.weights` stream for that topology (biases then
filters per layer, `[connected]` weights as [output][input] with the CHW flatten order of :196-198) instead of a TF checkpoint;
images are read with PIL and handed over in OpenCV's BGR channel order like `cv2.imread` does (:297); the drawing half of
`show_results` (:394-430) is outside the inference path -- the boxes file is still written.  There is no CPU path here."""
import numpy as np
from . import hip, darknet_io as IO


class Yolo(object):
    CFG = "yolov1"
    THRESHOLD, IOU_THRESHOLD, MAX_OUTPUT_SIZE = 0.2, 0.4, 10       # V1/YOLO_V1_Inference.py:47-49
    BGR = True                                                       # the image reaches the network in cv2.imread's channel order

    def __init__(self, weights_file, input_image=None, verbose=True, dtype=hip.BF16, device=0, weights=None):
        self.verbose = verbose
        self.S = 7          # cells per side
        self.B = 2          # boxes per cell
        self.classes = IO.v1_classes()
        self.C = len(self.classes)
        self.threshold = self.THRESHOLD           # class-specific confidence threshold (`>=`)
        self.iou_threshold = self.IOU_THRESHOLD
        self.max_output_size = self.MAX_OUTPUT_SIZE
        self.cfg_text = IO.cfg_text(self.CFG)
        self.engine = hip.Engine(self.cfg_text, max_batch=1, dtype=dtype, semantics=hip.SEM_TF, decode=hip.DECODE_RATIO, device=device)
        self._load_weights(weights_file, weights)
        if input_image is not None:
            self.detect_from_file(input_image)

    def _load_weights(self, weights_file, weights=None):
        if self.verbose:
            print("Start to load weights from file:%s" % (weights_file,))
        if weights is not None:
            self.engine.set_weights(weights)
        else:
            self.engine.load_weights(weights_file)

    def close(self):
        self.engine.close()

    # ---- V1/YOLO_V1_Inference.py:371-392 ----
    def _detect_from_image(self, image):
        """`image`: uint8 [h, w, 3] as cv2.imread returns it.  -> (scores [K], boxes [K,4] = (cx, cy, w, h) in pixels of `image`,
        box_classes [K]), K <= 10, best first."""
        image = np.ascontiguousarray(image, dtype=np.uint8)
        img_h, img_w, _ = image.shape
        det = self.engine.forward_image(image)[0]                       # [S*S*B, 5 + C] rows (cx, cy, w, h, conf, cls...)
        kept, rows = self.engine.postprocess(1, score_thr=self.threshold, iou_thr=self.iou_threshold, max_out=self.max_output_size,
                                             nms_mode=hip.NMS_TF_V1, select_mode=hip.SELECT_GE, return_rows=True)
        kept, rows = kept[0], rows[0].astype(np.int64)
        # the records carry the corners the reference hands to tf.image.non_max_suppression (horizontal extent from h, vertical from
        # w); `self.boxes` of the reference is the decoded (cx, cy, w, h) row itself, gathered with the NMS indices (:264-268): the
        # library reports that row index with every record (yolo_postprocess_rows)
        f = np.float32
        scores = kept["score"].copy(); box_classes = kept["cls"].astype(np.int64)
        boxes = det[rows, :4].copy() if len(rows) else np.zeros((0, 4), np.float32)
        boxes[:, 0] *= f(1.0 * img_w); boxes[:, 1] *= f(1.0 * img_h); boxes[:, 2] *= f(1.0 * img_w); boxes[:, 3] *= f(1.0 * img_h)
        return scores, boxes, box_classes

    # ---- V1/YOLO_V1_Inference.py:294-307 ----
    def detect_from_file(self, image_file, imshow=True, deteted_boxes_file="boxes.txt", detected_image_file="detected_image.jpg"):
        from PIL import Image
        rgb = np.asarray(Image.open(image_file).convert("RGB"))
        image = np.ascontiguousarray(rgb[:, :, ::-1]) if self.BGR else np.ascontiguousarray(rgb)      # cv2.imread: BGR
        scores, boxes, box_classes = self._detect_from_image(image)
        predict_boxes = []
        for i in range(len(scores)):
            predict_boxes.append((self.classes[box_classes[i]], boxes[i, 0], boxes[i, 1], boxes[i, 2], boxes[i, 3], scores[i]))
        self.show_results(image, predict_boxes, imshow, deteted_boxes_file, detected_image_file)
        return predict_boxes

    # ---- V1/YOLO_V1_Inference.py:394-430, the text half ----
    def show_results(self, image, results, imshow=True, deteted_boxes_file=None, detected_image_file=None):
        f = open(deteted_boxes_file, "w") if deteted_boxes_file else None
        for r in results:
            x = int(r[1]); y = int(r[2]); w = int(r[3]) // 2; h = int(r[4]) // 2
            if self.verbose:
                print("class: %s, [x, y, w, h]=[%d, %d, %d, %d], confidence=%f" % (r[0], x, y, w, h, r[-1]))
            if f:
                f.write(r[0] + "," + str(x) + "," + str(y) + "," + str(w) + "," + str(h) + "," + str(r[5]) + "\n")
        if f:
            f.close()


class YOLOV1_Tiny(Yolo):
    """D2T/YOLO_V1_Tiny_convert_darkenet_to_Tensorflow.py:53-98 `YOLOV1_Tiny(weights_file, verbose=True)`: the eight-conv tiny YOLOv1
    (`cfg/yolov1-tiny.cfg`), input x / 255 in RGB order (:212-216, :473-474), thresholds from that script's flags (:30-32), the same
    `_build_detector` (:326-386) and `detect_from_image` / `detect_from_file` (:470-500).  `weights_file` is darknet's
    tiny-yolov1.weights stream (4-int header, `_load_weights` :99-203)."""
    CFG = "yolov1-tiny"
    THRESHOLD, IOU_THRESHOLD, MAX_OUTPUT_SIZE = 0.1, 0.6, 10
    BGR = False

    def __init__(self, weights_file, verbose=True, dtype=hip.BF16, device=0, weights=None):
        Yolo.__init__(self, weights_file, None, verbose, dtype, device, weights)

    def detect_from_image(self, image):
        return self._detect_from_image(image)

