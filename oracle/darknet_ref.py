"""ctypes driver for the compiled reference (oracle/_ref/libdarknet_ref.so, built by oracle/Makefile
from /root/reference's own darknet C sources).

**TEST INFRASTRUCTURE ONLY** (same rule as yolo_ref.py).  Mirrors the reference's own binding
D2T/darknet.py:20-115 (struct BOX/DETECTION/IMAGE, load_network, network_predict, get_network_boxes,
do_nms_sort, free_detections) -- that file is Python-2 syntax and binds a CUDA build, so it cannot be
imported; this is a fresh Python-3 binding of the same C ABI plus the accessors of ref_shim.c.
"""
import ctypes as C
import os
import tempfile
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_ref", "libdarknet_ref.so")


class BOX(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("w", C.c_float), ("h", C.c_float)]


class DETECTION(C.Structure):
    _fields_ = [("bbox", BOX), ("classes", C.c_int), ("prob", C.POINTER(C.c_float)),
                ("mask", C.POINTER(C.c_float)), ("objectness", C.c_float), ("sort_class", C.c_int)]


_lib = None


def available():
    return os.path.exists(LIB_PATH)


def lib():
    global _lib
    if _lib is None:
        l = C.CDLL(LIB_PATH, C.RTLD_GLOBAL)
        l.load_network.argtypes = [C.c_char_p, C.c_char_p, C.c_int]; l.load_network.restype = C.c_void_p
        l.free_network.argtypes = [C.c_void_p]
        l.set_batch_network.argtypes = [C.c_void_p, C.c_int]
        l.network_predict.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        l.network_predict.restype = C.POINTER(C.c_float)
        l.get_network_boxes.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float,
                                        C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int)]
        l.get_network_boxes.restype = C.POINTER(DETECTION)
        l.free_detections.argtypes = [C.POINTER(DETECTION), C.c_int]
        l.do_nms_sort.argtypes = [C.POINTER(DETECTION), C.c_int, C.c_int, C.c_float]
        l.do_nms_obj.argtypes = [C.POINTER(DETECTION), C.c_int, C.c_int, C.c_float]
        for name in ("ref_num_layers", "ref_net_w", "ref_net_h"):
            getattr(l, name).argtypes = [C.c_void_p]; getattr(l, name).restype = C.c_int
        for name in ("ref_layer_outputs", "ref_layer_type"):
            getattr(l, name).argtypes = [C.c_void_p, C.c_int]; getattr(l, name).restype = C.c_int
        l.ref_layer_output.argtypes = [C.c_void_p, C.c_int]; l.ref_layer_output.restype = C.POINTER(C.c_float)
        l.ref_layer_dims.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        _lib = l
    return _lib


class _Quiet:
    """The fork's loader floods stdout/stderr (DN/parser.c:1176-1228): silence fds 1 and 2."""

    def __enter__(self):
        import sys
        sys.stdout.flush(); sys.stderr.flush()
        self.saved = [os.dup(1), os.dup(2)]
        nul = os.open(os.devnull, os.O_WRONLY)
        os.dup2(nul, 1); os.dup2(nul, 2); os.close(nul)

    def __exit__(self, *a):
        lib_c = C.CDLL(None)
        lib_c.fflush(None)
        os.dup2(self.saved[0], 1); os.dup2(self.saved[1], 2)
        os.close(self.saved[0]); os.close(self.saved[1])


class RefNet:
    """load_network(cfg, weights) -> predict -> per-layer outputs / boxes (batch 1, like the reference)."""

    def __init__(self, cfg_text, flat_weights=None, major=0, minor=2):
        self.l = lib()
        self.tmp = tempfile.mkdtemp(prefix="dnref_")
        cfg = os.path.join(self.tmp, "net.cfg")
        with open(cfg, "w") as f:
            f.write(cfg_text)
        wpath = None
        if flat_weights is not None:
            wpath = os.path.join(self.tmp, "net.weights")
            with open(wpath, "wb") as f:
                np.array([major, minor, 0], dtype=np.int32).tofile(f)
                (np.zeros(1, np.int64) if major * 10 + minor >= 2 else np.zeros(1, np.int32)).tofile(f)
                np.asarray(flat_weights, dtype=np.float32).tofile(f)
        with _Quiet():
            self.net = self.l.load_network(cfg.encode(), wpath.encode() if wpath else None, 0)
            self.l.set_batch_network(self.net, 1)
        self.w = self.l.ref_net_w(self.net); self.h = self.l.ref_net_h(self.net)
        self.n = self.l.ref_num_layers(self.net)

    def predict(self, img_hwc01):
        """img [H,W,3] float32 in 0..1 already at network size -> runs network_predict on CHW data."""
        chw = np.ascontiguousarray(np.transpose(np.asarray(img_hwc01, dtype=np.float32), (2, 0, 1)))
        self._keep = chw
        self.l.network_predict(self.net, chw.ctypes.data_as(C.POINTER(C.c_float)))

    def layer_output_nhwc(self, i):
        whc = (C.c_int * 3)()
        self.l.ref_layer_dims(self.net, i, whc)
        w, h, c = whc[0], whc[1], whc[2]
        n = self.l.ref_layer_outputs(self.net, i)
        buf = np.ctypeslib.as_array(self.l.ref_layer_output(self.net, i), shape=(n,)).copy()
        if w * h * c != n:
            return buf
        return np.ascontiguousarray(buf.reshape(c, h, w).transpose(1, 2, 0))[None]

    def boxes(self, thresh, nms=None, classes=80, relative=1):
        """get_network_boxes(net, netw, neth, thresh, .5, NULL, relative) [+ do_nms_sort] with w=h=net size
        (neutralises the letterbox correction, SURVEY 8a').  -> (bbox [n,4] cx,cy,w,h; obj [n]; prob [n,C])."""
        num = C.c_int(0)
        dets = self.l.get_network_boxes(self.net, self.w, self.h, thresh, .5, None, relative, C.byref(num))
        n = num.value
        if nms is not None and n:
            self.l.do_nms_sort(dets, n, classes, nms)
        bb = np.zeros((n, 4), np.float32); obj = np.zeros(n, np.float32); pr = np.zeros((n, classes), np.float32)
        for i in range(n):
            d = dets[i]
            bb[i] = (d.bbox.x, d.bbox.y, d.bbox.w, d.bbox.h); obj[i] = d.objectness
            pr[i] = np.ctypeslib.as_array(d.prob, shape=(classes,))
        self.l.free_detections(dets, n)
        return bb, obj, pr

    def close(self):
        if self.net:
            self.l.free_network(self.net); self.net = None
