"""Counterparts of the converter classes `YOLOV3(weights_file)` / `YOLOV2(weights_file)`
(D2T/YOLO_V3_convert_darkenet_to_Tensorflow.py:51-618, D2T/YOLO_V2_convert_darkenet_to_Tensorflow.py:50-538):
"Darknet weights in, boxes out" -- uint8 image of any size -> /255 -> legacy bilinear stretch -> network -> decode ->
threshold -> TF NMS -> (scores, boxes, classes).  Flag names and defaults are the reference's (:34-49)."""
import numpy as np
from . import hip, darknet_io as IO


class _Flags:
    def __init__(self, **kw):
        self.__dict__.update(kw)


FLAGS_V3 = _Flags(input_size=416, conf_threshold=0.4, iou_threshold=0.4, max_output_size=10, class_names="data/coco.names",
                  darknet_weights_file="weight/yolov3.weights")
FLAGS_V2 = _Flags(input_size=416, conf_threshold=0.5, iou_threshold=0.5, max_output_size=10, class_names="data/coco.names",
                  darknet_weights_file="weight/yolov2.weights")


class _Detector:
    cfg = None
    flags = None
    select_mode = hip.SELECT_GT

    def __init__(self, weights_file, verbose=False, dtype=hip.BF16, device=0, weights=None, max_output_size=None):
        f = self.flags
        self.verbose = verbose
        self.threshold, self.iou_threshold = f.conf_threshold, f.iou_threshold
        # the V3 converter stores FLAGS.max_output_size (10) but never passes it: the graph default 20 is what runs
        # (D2T/...V3...py:60 vs :440,66-69); the V2 converter passes its value through.
        self.max_output_size = max_output_size if max_output_size is not None else self.graph_max_output
        self.input_size = f.input_size
        text = IO.cfg_text(self.cfg)
        if self.input_size != int(IO.parse_cfg(text)[0]["width"]):
            text = IO.with_input_size(text, self.input_size)
        self.engine = hip.Engine(text, max_batch=1, dtype=dtype, semantics=hip.SEM_TF, decode=hip.DECODE_RATIO, device=device)
        if weights is not None:
            self.engine.set_weights(weights)
        else:
            self.engine.load_weights(weights_file, self.header_ints)
        try:
            self.class_names = self.load_coco_names(f.class_names)
        except OSError:
            self.class_names = {}

    @staticmethod
    def load_coco_names(file_name):
        names = {}
        with open(file_name) as fh:
            for i, name in enumerate(fh):
                names[i] = name.strip()
        return names

    def detect_from_image(self, image):
        """image: RGB uint8 [H,W,3] (what cv2.cvtColor(BGR2RGB) produced in the reference) -> (scores, boxes, classes);
        boxes are normalised (x0,y0,x1,y1) like `detected_boxes:0` (D2T/...V3...py:585-592)."""
        self.engine.forward_image(np.ascontiguousarray(image, dtype=np.uint8))
        r = self.engine.postprocess(1, score_thr=self.threshold, iou_thr=self.iou_threshold, max_out=self.max_output_size,
                                    nms_mode=hip.NMS_TF, select_mode=self.select_mode)[0]
        boxes = np.stack([r["x0"], r["y0"], r["x1"], r["y1"]], -1).reshape(-1, 4)
        return r["score"], boxes, r["cls"]

    def detect_from_file(self, image_file, imshow=False, deteted_boxes_file="boxes.txt", detected_image_file=None):
        """(sic: `deteted_boxes_file` is the reference's spelling.)  Reads the image with PIL instead of OpenCV and
        returns the predictions; with `detected_image_file` the boxes are drawn (draw.draw_detection) and the picture is saved."""
        from PIL import Image
        img = np.asarray(Image.open(image_file).convert("RGB"))
        scores, boxes, classes = self.detect_from_image(img)
        preds = [(self.class_names.get(int(c), int(c)), float(b[0]), float(b[1]), float(b[2]), float(b[3]), float(s))
                 for s, b, c in zip(scores, boxes, classes)]
        if deteted_boxes_file:
            with open(deteted_boxes_file, "w") as fh:
                for p in preds:
                    fh.write(",".join(str(v) for v in p) + "\n")
        if detected_image_file:           # the reference's draw_detection + cv2.imwrite (D2T/...V3...py:547-582, :600-612), with PIL
            from . import draw
            labels = self.class_names if self.class_names else {int(c): str(int(c)) for c in classes}
            n = max(len(labels), int(max(classes, default=0)) + 1) if isinstance(labels, dict) else len(labels)
            names = [labels.get(i, str(i)) for i in range(n)] if isinstance(labels, dict) else labels
            Image.fromarray(draw.draw_detection(img, boxes, scores, classes, names, thr=0.3, ratio=True)).save(detected_image_file)
        return preds


class YOLOV3(_Detector):
    cfg, flags, header_ints, graph_max_output = "yolov3", FLAGS_V3, 5, 20


class YOLOV3Tiny(_Detector):
    cfg, flags, header_ints, graph_max_output = "yolov3-tiny", FLAGS_V3, 5, 20


class YOLOV2(_Detector):
    cfg, flags, header_ints, graph_max_output = "yolov2", FLAGS_V2, 4, 10
    select_mode = hip.SELECT_GE          # `filter_mask = box_class_scores >= threshold` (D2T V2 :314 / V2/postprocess.py:61)


class YOLOV2TinyVoc(_Detector):
    cfg, flags, header_ints, graph_max_output = "yolov2-tiny-voc", FLAGS_V2, 4, 10
    select_mode = hip.SELECT_GE
