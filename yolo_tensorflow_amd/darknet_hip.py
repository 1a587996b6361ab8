"""Counterpart of the reference's `darknet.py` (D2T/darknet.py:20-142), named darknet_hip to keep the two apart: the same ctypes
declarations (BOX, DETECTION, IMAGE, METADATA) and every name that file binds (`load_net`, `load_meta`, `load_image`,
`predict_image`, `get_network_boxes`, `make_network_boxes`, `do_nms_obj`, `do_nms_sort`, `free_detections`, `free_ptrs`,
`letterbox_image`, `rgbgr_image`, `set_gpu`, `reset_rnn`, `detect`), bound to `libdarknet_hip.so` (include/darknet_hip.h)
instead of `./libdarknet.so`.

Differences a caller sees: `load_image` decodes binary PPM / PGM only (the reference's stb JPEG decoding, D2T/darknet.py:105,
is outside the inference path), so `detect` also takes the image as an array (RGB, HWC uint8 or float 0..1) or a ready IMAGE, and
the class names as a METADATA or a plain list."""
import ctypes as C
import os
import numpy as np
from .hip import YoloError

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdarknet_hip.so")


class BOX(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("w", C.c_float), ("h", C.c_float)]


class DETECTION(C.Structure):
    _fields_ = [("bbox", BOX), ("classes", C.c_int), ("prob", C.POINTER(C.c_float)), ("mask", C.POINTER(C.c_float)),
                ("objectness", C.c_float), ("sort_class", C.c_int)]


class IMAGE(C.Structure):
    _fields_ = [("w", C.c_int), ("h", C.c_int), ("c", C.c_int), ("data", C.POINTER(C.c_float))]


class METADATA(C.Structure):
    _fields_ = [("classes", C.c_int), ("names", C.POINTER(C.c_char_p))]


_lib = None


# prototype table: name -> (restype, argtypes); the layouts are those of include/darknet_hip.h
_PROTOTYPES = {
    "load_network": (C.c_void_p, [C.c_char_p, C.c_char_p, C.c_int]),
    "free_network": (None, [C.c_void_p]),
    "network_width": (C.c_int, [C.c_void_p]),
    "network_height": (C.c_int, [C.c_void_p]),
    "network_predict": (C.POINTER(C.c_float), [C.c_void_p, C.POINTER(C.c_float)]),
    "network_predict_image": (C.POINTER(C.c_float), [C.c_void_p, IMAGE]),
    "get_network_boxes": (C.POINTER(DETECTION), [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int)]),
    "free_detections": (None, [C.POINTER(DETECTION), C.c_int]),
    "do_nms_obj": (None, [C.POINTER(DETECTION), C.c_int, C.c_int, C.c_float]),
    "do_nms_sort": (None, [C.POINTER(DETECTION), C.c_int, C.c_int, C.c_float]),
    "cuda_set_device": (None, [C.c_int]),
    "make_image": (IMAGE, [C.c_int, C.c_int, C.c_int]),
    "free_image": (None, [IMAGE]),
    "make_network_boxes": (C.POINTER(DETECTION), [C.c_void_p, C.c_float, C.POINTER(C.c_int)]),
    "free_ptrs": (None, [C.POINTER(C.c_void_p), C.c_int]),
    "reset_rnn": (None, [C.c_void_p]),
    "set_batch_network": (None, [C.c_void_p, C.c_int]),
    "letterbox_image": (IMAGE, [IMAGE, C.c_int, C.c_int]),
    "get_metadata": (METADATA, [C.c_char_p]),
    "load_image_color": (IMAGE, [C.c_char_p, C.c_int, C.c_int]),
    "rgbgr_image": (None, [IMAGE]),
}


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise YoloError("libdarknet_hip.so is not built (make -C yolo_tensorflow_amd/csrc); there is no CPU fallback")
        handle = C.CDLL(LIB_PATH)
        for name, (restype, argtypes) in _PROTOTYPES.items():
            fn = getattr(handle, name)
            fn.restype = restype; fn.argtypes = argtypes
        _lib = handle
    return _lib


def load_net(cfg, weights, clear=0):
    net = _load().load_network(os.fsencode(cfg), os.fsencode(weights) if weights else None, clear)
    if not net:
        raise YoloError("load_network failed for %s / %s" % (cfg, weights))
    return net


def free_net(net):
    _load().free_network(net)


def load_meta(path):
    """D2T/darknet.py:101-103."""
    return _load().get_metadata(os.fsencode(path))


def load_image(path, w=0, h=0):
    """D2T/darknet.py:105-107 (binary PPM / PGM files)."""
    im = _load().load_image_color(os.fsencode(path), w, h)
    if not im.data:
        raise YoloError("load_image_color failed for %s" % path)
    return im


def set_gpu(n):
    _load().cuda_set_device(n)


def array_to_image(arr):
    """HWC RGB uint8 (0..255) or float (0..1) -> darknet IMAGE (planar float 0..1); keeps the buffer alive on the result."""
    a = np.asarray(arr)
    a = a.astype(np.float32) / np.float32(255) if a.dtype == np.uint8 else a.astype(np.float32)
    chw = np.ascontiguousarray(a.transpose(2, 0, 1))
    im = IMAGE(chw.shape[2], chw.shape[1], chw.shape[0], chw.ctypes.data_as(C.POINTER(C.c_float)))
    im._keep = chw
    return im


def predict_image(net, im):
    out = _load().network_predict_image(net, im)
    if not out:
        raise YoloError("network_predict_image failed")
    return out


def detect(net, names, image, thresh=.5, hier_thresh=.5, nms=.45):
    """D2T/darknet.py:125-142; `image`: a PPM path (as the reference passes a file name), an array or an IMAGE; `names`: a
    METADATA (load_meta) or the class-name list."""
    l = _load()
    own = isinstance(image, (str, bytes))
    im = load_image(image, 0, 0) if own else image if isinstance(image, IMAGE) else array_to_image(image)
    if isinstance(names, METADATA):
        names = [names.names[i] for i in range(names.classes)]
    num = C.c_int(0)
    predict_image(net, im)
    dets = l.get_network_boxes(net, im.w, im.h, thresh, hier_thresh, None, 0, C.byref(num))
    n = num.value
    classes = len(names)
    if nms:
        l.do_nms_obj(dets, n, classes, nms)
    res = []
    for j in range(n):
        for i in range(classes):
            if dets[j].prob[i] > 0:
                b = dets[j].bbox
                res.append((names[i], dets[j].prob[i], (b.x, b.y, b.w, b.h)))
    res = sorted(res, key=lambda x: -x[1])
    if own:
        l.free_image(im)
    l.free_detections(dets, n)
    return res
