"""ctypes shim over libyolo_hip.so (the C ABI in include/yolo_hip.h).

This is the only module that talks to the device.  It replaces the reference's `tf.Session.run`
boundary (V3/YOLO_V3_inference.py:106-107) and plays the role D2T/darknet.py:48-115 plays for
libdarknet.so.  There is deliberately **no CPU fallback**: if the library is missing or no HIP
device is visible every entry point raises.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# YOLO_HIP_LIB: another build of the SAME library (A/B probes, tools/probe/ab) -- selected, never copied over the in-tree file, whose
# sources buildinfo.source_hash() stamps the profiles with; a line measured through it says so (bench.py: "lib_override")
LIB_OVERRIDE = os.environ.get("YOLO_HIP_LIB") or None
LIB_PATH = LIB_OVERRIDE or os.path.join(_HERE, "libyolo_hip.so")

BF16, FP32, FP8, FP16, FP16X2 = 0, 1, 2, 3, 4
SEM_TF, SEM_DARKNET = 0, 1
DECODE_RATIO, DECODE_PIXEL = 0, 1
HOST, DEVICE = 0, 1
IMG_U8, IMG_F32, IMG_F32_CHW = 0, 1, 2
NMS_TF, NMS_PER_CLASS, NMS_DARKNET, NMS_NUMPY_V3, NMS_TF_V1 = 0, 1, 2, 3, 4
SELECT_GT, SELECT_GE = 0, 1

BOX_DTYPE = np.dtype([("x0", "<f4"), ("y0", "<f4"), ("x1", "<f4"), ("y1", "<f4"), ("score", "<f4"), ("cls", "<i4")])

EXPORTS = [
    "yolo_create", "yolo_destroy", "yolo_last_error", "yolo_load_darknet_weights", "yolo_set_weights",
    "yolo_weights_count", "yolo_set_act_scales", "yolo_export", "yolo_create_from_file", "yolo_input_size", "yolo_num_rows", "yolo_num_attrs", "yolo_num_layers", "yolo_head_geometry",
    "yolo_conv_flops", "yolo_conv_bytes", "yolo_forward", "yolo_forward_image_u8", "yolo_postprocess",
    "yolo_detect", "yolo_detect_graph", "yolo_synchronize", "yolo_layer_output", "yolo_time_forward", "yolo_time_layers",
    "yolo_autotune", "yolo_get_tile_configs", "yolo_set_tile_configs", "yolo_op_conv2d", "yolo_op_conv_num_cfgs", "yolo_op_upsample2x", "yolo_op_reorg",
    "yolo_darknet_boxes", "yolo_last_layer_size", "yolo_last_layer_output", "yolo_op_letterbox",
    "yolo_op_maxpool", "yolo_op_resize_u8", "yolo_op_detections_boxes", "yolo_op_nms_detections", "yolo_forward_letterbox_chw", "yolo_op_decode", "yolo_op_postprocess",
    "yolo_postprocess_rows", "yolo_op_postprocess_rows", "yolo_last_layer_output_batch", "yolo_head_raw", "yolo_calibrate", "yolo_calibrate_copy", "yolo_op_resize_cv2",
]
# include/yolo_dist.h: the image-sharded detect step
DIST_EXPORTS = ["yolo_shard_bounds", "yolo_dist_flat_words", "yolo_dist_split_records", "yolo_dist_unique_id", "yolo_dist_create",
                "yolo_dist_detect", "yolo_dist_detect_async", "yolo_dist_destroy"]


class YoloError(RuntimeError):
    pass


class _Config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("cfg_text", C.c_char_p), ("max_batch", C.c_int32),
                ("dtype", C.c_int32), ("semantics", C.c_int32), ("decode", C.c_int32), ("device", C.c_int32),
                ("keep_layers", C.c_int32), ("stream", C.c_void_p)]


_lib = None


def load_library():
    """dlopen libyolo_hip.so and declare every prototype; raises YoloError when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise YoloError("%s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(or `make -C yolo_tensorflow_amd/csrc`); there is no CPU fallback" % LIB_PATH)
    l = C.CDLL(LIB_PATH)
    P, I, F, D = C.c_void_p, C.c_int, C.c_float, C.c_double
    FP = C.POINTER(C.c_float)
    l.yolo_create.argtypes = [C.POINTER(_Config), C.c_char_p, C.c_size_t]; l.yolo_create.restype = P
    l.yolo_destroy.argtypes = [P]; l.yolo_destroy.restype = None
    l.yolo_last_error.argtypes = [P]; l.yolo_last_error.restype = C.c_char_p
    l.yolo_load_darknet_weights.argtypes = [P, C.c_char_p, I]
    l.yolo_set_weights.argtypes = [P, FP, C.c_size_t]
    l.yolo_weights_count.argtypes = [P]; l.yolo_weights_count.restype = C.c_size_t
    l.yolo_set_act_scales.argtypes = [P, P, I]
    l.yolo_head_geometry.argtypes = [P, I, P, P, P, P]
    l.yolo_export.argtypes = [P, C.c_char_p]
    l.yolo_create_from_file.argtypes = [C.c_char_p, I, I, P, I, C.c_char_p, C.c_size_t]; l.yolo_create_from_file.restype = P
    l.yolo_input_size.argtypes = [P, C.POINTER(I), C.POINTER(I), C.POINTER(I)]
    for n in ("yolo_num_rows", "yolo_num_attrs", "yolo_num_layers", "yolo_synchronize"):
        getattr(l, n).argtypes = [P]
    l.yolo_conv_flops.argtypes = [P]; l.yolo_conv_flops.restype = D
    l.yolo_conv_bytes.argtypes = [P, I]; l.yolo_conv_bytes.restype = D
    l.yolo_forward.argtypes = [P, P, I, I, I, F, P, I]
    l.yolo_forward_image_u8.argtypes = [P, P, I, I, I, P, I]
    l.yolo_postprocess.argtypes = [P, I, F, F, I, I, I, P, P, I]
    l.yolo_detect.argtypes = [P, P, I, I, I, F, F, F, I, I, I, P, P, I]
    l.yolo_detect_graph.argtypes = [P, P, I, I, F, F, F, I, I, I, P, P]
    l.yolo_layer_output.argtypes = [P, I, I, P, C.c_size_t, C.POINTER(I)]
    l.yolo_time_forward.argtypes = [P, I, I, FP, FP]
    l.yolo_time_layers.argtypes = [P, I, I, FP]
    l.yolo_autotune.argtypes = [P, I, I]
    l.yolo_get_tile_configs.argtypes = [P, P]
    l.yolo_set_tile_configs.argtypes = [P, P]
    l.yolo_op_conv2d.argtypes = [P, I, I, I, I, P, P, I, I, I, I, P, P, I, I, I]
    l.yolo_op_conv_num_cfgs.argtypes = []
    l.yolo_op_upsample2x.argtypes = [P, I, I, I, I, I, P, I]
    l.yolo_op_reorg.argtypes = [P, I, I, I, I, I, I, P, I]
    l.yolo_op_maxpool.argtypes = [P, I, I, I, I, I, I, P, I]
    l.yolo_op_resize_u8.argtypes = [P, I, I, I, F, P, I]
    l.yolo_op_detections_boxes.argtypes = [P, I, I, I, P, I]
    l.yolo_op_nms_detections.argtypes = [P, P, P, I, I, F, I, I]
    l.yolo_forward_letterbox_chw.argtypes = [P, P, I, I, I, P, I]
    l.yolo_op_decode.argtypes = [P, I, I, I, I, P, I, I, I, P, I]
    l.yolo_darknet_boxes.argtypes = [P, I, I, F, I, P, I, P]
    l.yolo_last_layer_size.argtypes = [P]; l.yolo_last_layer_size.restype = C.c_size_t
    l.yolo_last_layer_output.argtypes = [P, P, C.c_size_t]
    l.yolo_op_letterbox.argtypes = [P, I, I, I, I, I, P, I]
    l.yolo_op_postprocess.argtypes = [P, I, I, I, F, F, I, I, I, P, P, I]
    l.yolo_op_postprocess_rows.argtypes = [P, I, I, I, F, F, I, I, I, P, P, P, I]
    l.yolo_postprocess_rows.argtypes = [P, I, F, F, I, I, I, P, P, P, I]
    l.yolo_last_layer_output_batch.argtypes = [P, I, P, C.c_size_t]
    l.yolo_head_raw.argtypes = [P, I, I, P, C.c_size_t]
    l.yolo_calibrate.argtypes = [I, P, I, C.c_double, FP, FP]
    l.yolo_calibrate_copy.argtypes = [I, P, C.c_double, FP]
    l.yolo_op_resize_cv2.argtypes = [P, I, I, I, I, I, F, P, I]
    l.yolo_shard_bounds.argtypes = [I, I, I, C.POINTER(I), C.POINTER(I)]
    l.yolo_dist_flat_words.argtypes = [I, I]; l.yolo_dist_flat_words.restype = C.c_size_t
    l.yolo_dist_split_records.argtypes = [P, I, I, I, P, P]
    l.yolo_dist_unique_id.argtypes = [P]
    l.yolo_dist_create.argtypes = [P, I, I, P, P, I, I, C.c_char_p, C.c_size_t]; l.yolo_dist_create.restype = P
    l.yolo_dist_detect.argtypes = [P, P, I, F, F, F, I, I, P, P]
    l.yolo_dist_detect_async.argtypes = [P, P, I, F, F, F, I, I, C.POINTER(P)]
    l.yolo_dist_destroy.argtypes = [P]; l.yolo_dist_destroy.restype = None
    _lib = l
    return l


def _ptr(a):
    """host numpy array / torch tensor (host or device) / raw int pointer -> (void*, location)."""
    if a is None:
        return None, HOST
    if isinstance(a, int):
        return C.c_void_p(a), DEVICE
    if isinstance(a, np.ndarray):
        if not a.flags["C_CONTIGUOUS"]:
            raise YoloError("array must be C-contiguous")
        return C.c_void_p(a.ctypes.data), HOST
    if hasattr(a, "data_ptr"):   # torch tensor, no torch import needed
        if not a.is_contiguous():
            raise YoloError("tensor must be contiguous")
        return C.c_void_p(a.data_ptr()), (DEVICE if a.is_cuda else HOST)
    raise YoloError("unsupported buffer type %r" % type(a))


def _op_check(rc, what):
    if rc != 0:
        msg = load_library().yolo_last_error(None)
        raise YoloError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else ""))


_warned_legacy = False
HIP_STREAM_LEGACY = 1     # hipStreamLegacy: the explicit handle of the legacy NULL stream (hip_runtime_api.h)


def _stream_handle(stream):
    """None: the engine creates a stream of its own.  An integer handle: the caller's stream -- and a handle of 0, which is what
    torch.cuda.current_stream().cuda_stream reads on torch's DEFAULT stream, means that stream, not "none": it is passed on as
    hipStreamLegacy, so that everything stays stream-ordered (NULL at the C boundary would mean "create one").  The legacy stream cannot
    be captured: detect_graph then launches the step eagerly (same device time, ~0.17 ms of host time per step; a one-time
    RuntimeWarning says so).  The legacy stream orders itself against torch's default stream and every BLOCKING stream; a tensor produced
    on a NON-blocking side stream (torch.cuda.Stream() is one) must be ordered by the caller (event / synchronize) before it is handed
    over -- unlike stream=None, where the engine waits on the host for the producer stream."""
    if stream is None:
        return None
    if int(stream) == 0:
        global _warned_legacy
        if not _warned_legacy:
            _warned_legacy = True
            import warnings
            warnings.warn("stream handle 0 is torch's default (legacy NULL) stream: the engine runs on hipStreamLegacy, detect_graph launches "
                          "eagerly (the legacy stream cannot be captured) and tensors from non-blocking side streams are not ordered "
                          "against it; pass stream=None for an engine-owned stream or a created stream's handle", RuntimeWarning, stacklevel=3)
        return HIP_STREAM_LEGACY
    return int(stream)


class Engine:
    """One planned network on one GPU (a `yolo_ctx`)."""

    def __init__(self, cfg_text, max_batch=1, dtype=BF16, semantics=SEM_TF, decode=DECODE_RATIO, device=0,
                 keep_layers=False, stream=None):
        self.lib = load_library()
        self._cfg_bytes = cfg_text.encode()
        stream = _stream_handle(stream)
        conf = _Config(C.sizeof(_Config), self._cfg_bytes, max_batch, dtype, semantics, decode, device,
                       1 if keep_layers else 0, C.c_void_p(stream) if stream else None)
        err = C.create_string_buffer(512)
        self.ctx = self.lib.yolo_create(C.byref(conf), err, 512)
        if not self.ctx:
            raise YoloError("yolo_create: " + err.value.decode())
        h, w, ch = C.c_int(), C.c_int(), C.c_int()
        self.lib.yolo_input_size(self.ctx, C.byref(h), C.byref(w), C.byref(ch))
        self.size = h.value
        self.rows = self.lib.yolo_num_rows(self.ctx)
        self.attrs = self.lib.yolo_num_attrs(self.ctx)
        self.num_layers = self.lib.yolo_num_layers(self.ctx)
        self.max_batch = max_batch
        self.dtype = dtype
        self._own_stream = not stream

    @classmethod
    def from_file(cls, path, max_batch=1, device=0, keep_layers=False, stream=None):
        """Load an export artifact written by `export` (cfg + run configuration + packed parameters + tile plan)."""
        self = cls.__new__(cls)
        self.lib = load_library()
        stream = _stream_handle(stream)
        err = C.create_string_buffer(512)
        self.ctx = self.lib.yolo_create_from_file(os.fsencode(path), max_batch, device, C.c_void_p(stream) if stream else None,
                                                  1 if keep_layers else 0, err, 512)
        if not self.ctx:
            raise YoloError("yolo_create_from_file: " + err.value.decode())
        h, w, ch = C.c_int(), C.c_int(), C.c_int()
        self.lib.yolo_input_size(self.ctx, C.byref(h), C.byref(w), C.byref(ch))
        self.size = h.value
        self.rows = self.lib.yolo_num_rows(self.ctx); self.attrs = self.lib.yolo_num_attrs(self.ctx)
        self.num_layers = self.lib.yolo_num_layers(self.ctx); self.max_batch = max_batch; self.dtype = None
        self._own_stream = not stream
        return self

    def export(self, path):
        self._check(self.lib.yolo_export(self.ctx, os.fsencode(path)), "yolo_export")

    def _check(self, rc, what):
        if rc != 0:
            raise YoloError("%s failed (%d): %s" % (what, rc, self.lib.yolo_last_error(self.ctx).decode()))

    def _order_after_producer(self, *tensors):
        """Stream ordering for device tensors.  An engine created with stream=None runs on a stream of its own, which
        nothing orders against the torch stream that produced `tensors`: wait (on the host) for that stream first.  The
        engine's own device outputs are valid after `synchronize()`.  With an explicit stream (the `.cuda_stream` of a torch.cuda.Stream
        made current, what bench.py passes; or torch's default stream, handle 0, passed on as hipStreamLegacy) everything is
        stream-ordered and nothing is waited for."""
        if not self._own_stream:
            return
        for t in tensors:
            if hasattr(t, "is_cuda") and t.is_cuda:
                import torch
                torch.cuda.current_stream(t.device).synchronize()
                return

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.yolo_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- weights ----
    def weights_count(self):
        return self.lib.yolo_weights_count(self.ctx)

    def set_act_scales(self, scales):
        """FP8 engines: one activation scale per cfg layer (code = e4m3(value / scale)); call before loading weights."""
        sc = np.ascontiguousarray(scales, dtype=np.float32)
        self._check(self.lib.yolo_set_act_scales(self.ctx, sc.ctypes.data, sc.size), "yolo_set_act_scales")

    def load_weights(self, path, header_ints=0):
        self._check(self.lib.yolo_load_darknet_weights(self.ctx, os.fsencode(path), header_ints), "yolo_load_darknet_weights")

    def set_weights(self, flat):
        flat = np.ascontiguousarray(flat, dtype=np.float32)
        self._check(self.lib.yolo_set_weights(self.ctx, flat.ctypes.data_as(C.POINTER(C.c_float)), flat.size), "yolo_set_weights")

    # ---- hot path ----
    def forward(self, images, scale=1.0 / 255.0, want_detections=True, n=None, fmt=None, out=None):
        """images: uint8 or float32 [n,S,S,3] (numpy, torch host/device tensor, or raw device pointer with
        n and fmt given).  Returns the decoded tensor [n, rows, attrs] (numpy) unless out/want_detections say otherwise."""
        p, loc = _ptr(images)
        # a device uint8 image may be read IN PLACE by the fused stem, also by a later time_forward / time_layers / autotune pass
        # (yolo_hip.h, yolo_forward): keep it alive until the next image replaces it
        self._last_image = images if loc == DEVICE else None
        self._order_after_producer(images, out)
        if n is None:
            n = int(images.shape[0])
        if fmt is None:
            fmt = IMG_U8 if str(images.dtype).endswith("uint8") else IMG_F32
        det = None
        dp, dloc = None, HOST
        if out is not None:
            dp, dloc = _ptr(out)
        elif want_detections:
            det = np.empty((n, self.rows, self.attrs), dtype=np.float32)
            dp, dloc = _ptr(det)
        self._check(self.lib.yolo_forward(self.ctx, p, n, fmt, loc, scale, dp, dloc), "yolo_forward")
        return det

    def forward_image(self, image_u8):
        """One uint8 RGB image of any size: /255 + legacy bilinear stretch on the device, then forward."""
        image_u8 = np.ascontiguousarray(image_u8, dtype=np.uint8)
        det = np.empty((1, self.rows, self.attrs), dtype=np.float32)
        self._check(self.lib.yolo_forward_image_u8(self.ctx, image_u8.ctypes.data, image_u8.shape[0], image_u8.shape[1],
                                                   HOST, det.ctypes.data, HOST), "yolo_forward_image_u8")
        return det

    def postprocess(self, n, score_thr=0.5, iou_thr=0.5, max_out=20, nms_mode=NMS_TF, select_mode=SELECT_GT,
                    boxes_out=None, counts_out=None, rows_out=None, return_rows=False):
        """-> list of structured arrays (BOX_DTYPE) per image, or writes into the given device buffers.
        return_rows=True: -> (records, rows) where rows[i] holds, for every kept record of image i, the row of the decoded tensor
        it was formed from (yolo_postprocess_rows); rows_out: the same into a device int32 buffer [n * max_out]."""
        if boxes_out is not None:
            bp, bloc = _ptr(boxes_out); cp, _ = _ptr(counts_out); rp, _ = _ptr(rows_out)
            self._order_after_producer(boxes_out, counts_out, rows_out)
            self._check(self.lib.yolo_postprocess_rows(self.ctx, n, score_thr, iou_thr, max_out, nms_mode, select_mode, bp, cp, rp, bloc), "yolo_postprocess")
            return None
        boxes = np.zeros((n, max_out), dtype=BOX_DTYPE); counts = np.zeros(n, dtype=np.int32)
        rows = np.full((n, max_out), -1, dtype=np.int32) if return_rows else None
        self._check(self.lib.yolo_postprocess_rows(self.ctx, n, score_thr, iou_thr, max_out, nms_mode, select_mode,
                                                   boxes.ctypes.data, counts.ctypes.data, rows.ctypes.data if return_rows else None, HOST), "yolo_postprocess")
        recs = [boxes[i, :counts[i]].copy() for i in range(n)]
        if return_rows:
            return recs, [rows[i, :counts[i]].copy() for i in range(n)]
        return recs

    def head_raw(self, head, n):
        """Raw fp32 tensor [n, grid, grid, anchors * (5 + classes)] that detection head `head` decodes (yolo_head_raw)."""
        kind, grid, na, off = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        self._check(self.lib.yolo_head_geometry(self.ctx, head, C.byref(kind), C.byref(grid), C.byref(na), C.byref(off)), "yolo_head_geometry")
        out = np.empty((n, grid.value, grid.value, na.value * self.attrs), dtype=np.float32)
        self._check(self.lib.yolo_head_raw(self.ctx, head, n, out.ctypes.data, out.size), "yolo_head_raw")
        return out

    def last_layer_output(self, n=1):
        """darknet's net->output (DN/network.c:497-508) of images 0..n-1 of the last forward: [n, yolo_last_layer_size()] float32."""
        self.lib.yolo_last_layer_size.restype = C.c_size_t
        per = self.lib.yolo_last_layer_size(self.ctx)
        out = np.empty((n, per), dtype=np.float32)
        self._check(self.lib.yolo_last_layer_output_batch(self.ctx, n, out.ctypes.data, out.size), "yolo_last_layer_output_batch")
        return out

    def detect(self, images, scale=1.0 / 255.0, **kw):
        self.forward(images, scale=scale, want_detections=False)
        return self.postprocess(int(images.shape[0]), **kw)

    def detect_graph(self, images, boxes_out, counts_out, scale=1.0 / 255.0, score_thr=0.5, iou_thr=0.5, max_out=20,
                     nms_mode=NMS_TF, select_mode=SELECT_GT):
        """Device-resident detect replayed from a HIP graph (images / boxes_out / counts_out: device tensors)."""
        p, loc = _ptr(images); bp, bl = _ptr(boxes_out); cp, cl = _ptr(counts_out)
        if loc != DEVICE or bl != DEVICE or cl != DEVICE:
            raise YoloError("detect_graph needs device-resident buffers")
        self._last_image = images                   # (see forward)
        self._order_after_producer(images, boxes_out, counts_out)
        fmt = IMG_U8 if str(images.dtype).endswith("uint8") else IMG_F32
        self._check(self.lib.yolo_detect_graph(self.ctx, p, int(images.shape[0]), fmt, scale, score_thr, iou_thr, max_out,
                                               nms_mode, select_mode, bp, cp), "yolo_detect_graph")

    def synchronize(self):
        self._check(self.lib.yolo_synchronize(self.ctx), "yolo_synchronize")

    # ---- introspection / measurement ----
    def layer_output(self, index, n):
        dims = (C.c_int * 3)()
        self._check(self.lib.yolo_layer_output(self.ctx, index, n, None, 0, dims), "yolo_layer_output")
        out = np.empty((n, dims[0], dims[1], dims[2]), dtype=np.float32)
        self._check(self.lib.yolo_layer_output(self.ctx, index, n, out.ctypes.data, out.size, dims), "yolo_layer_output")
        return out

    def conv_flops(self):
        return self.lib.yolo_conv_flops(self.ctx)

    def conv_bytes(self, n):
        return self.lib.yolo_conv_bytes(self.ctx, n)

    def time_forward(self, n, iters, conv=True):
        t, c = C.c_float(0), C.c_float(0)
        self._check(self.lib.yolo_time_forward(self.ctx, n, iters, C.byref(t), C.byref(c) if conv else None), "yolo_time_forward")
        return t.value, (c.value if conv else None)

    def time_layers(self, n, iters):
        ms = np.zeros(self.num_layers, dtype=np.float32)
        self._check(self.lib.yolo_time_layers(self.ctx, n, iters, ms.ctypes.data_as(C.POINTER(C.c_float))), "yolo_time_layers")
        return ms

    def autotune(self, n, iters=3):
        self._check(self.lib.yolo_autotune(self.ctx, n, iters), "yolo_autotune")

    def get_tile_configs(self):
        cfgs = np.full(self.num_layers, -1, dtype=np.int32)
        self._check(self.lib.yolo_get_tile_configs(self.ctx, cfgs.ctypes.data), "yolo_get_tile_configs")
        return cfgs

    def set_tile_configs(self, cfgs):
        cfgs = np.ascontiguousarray(cfgs, dtype=np.int32)
        if cfgs.size != self.num_layers:
            raise YoloError("need %d tile configs, got %d" % (self.num_layers, cfgs.size))
        self._check(self.lib.yolo_set_tile_configs(self.ctx, cfgs.ctypes.data), "yolo_set_tile_configs")


# ---- single operators (host buffers in, host buffers out; production kernels underneath) ----------
def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def op_conv2d(x, w_hwio, bias=None, stride=1, act=0, residual=None, dtype=BF16, tile_cfg=-1, device=0):
    l = load_library()
    x = _f32(x); w = _f32(w_hwio)
    n, h, wd, cin = x.shape
    k, _, _, cout = w.shape
    pad = k // 2
    ho, wo = (h + 2 * pad - k) // stride + 1, (wd + 2 * pad - k) // stride + 1
    out = np.empty((n, ho, wo, cout), dtype=np.float32)
    b = _f32(bias) if bias is not None else None
    r = _f32(residual) if residual is not None else None
    rc = l.yolo_op_conv2d(x.ctypes.data, n, h, wd, cin, w.ctypes.data, b.ctypes.data if b is not None else None, k, stride,
                          cout, act, r.ctypes.data if r is not None else None, out.ctypes.data, dtype, tile_cfg, device)
    _op_check(rc, "yolo_op_conv2d")
    return out


def op_conv_num_cfgs():
    return load_library().yolo_op_conv_num_cfgs()


def calibrate(seconds=0.4, f16=False, device=0, stream=None):
    """yolo_calibrate: (TFLOP/s, GHz) this chip sustains on a register-resident MFMA loop (bf16, or fp16 with f16=True) -- the yardstick
    bench.py prints next to its roofline fraction so that lines measured on different boxes can be normalised."""
    t, g = C.c_float(0), C.c_float(0)
    _op_check(load_library().yolo_calibrate(device, _stream_handle(stream), 1 if f16 else 0, float(seconds), C.byref(t), C.byref(g)), "yolo_calibrate")
    return float(t.value), float(g.value)


def calibrate_copy(seconds=0.2, device=0, stream=None):
    """yolo_calibrate_copy: GB/s (read + written) of a streaming 1 GiB -> 1 GiB device copy -- the memory side of the box's yardstick."""
    g = C.c_float(0)
    _op_check(load_library().yolo_calibrate_copy(device, _stream_handle(stream), float(seconds), C.byref(g)), "yolo_calibrate_copy")
    return float(g.value)


def op_upsample2x(x, semantics=SEM_TF, device=0):
    x = _f32(x); n, h, w, c = x.shape
    out = np.empty((n, 2 * h, 2 * w, c), dtype=np.float32)
    _op_check(load_library().yolo_op_upsample2x(x.ctypes.data, n, h, w, c, semantics, out.ctypes.data, device), "yolo_op_upsample2x")
    return out


def op_reorg(x, stride=2, semantics=SEM_TF, device=0):
    x = _f32(x); n, h, w, c = x.shape
    out = np.empty((n, h // stride, w // stride, c * stride * stride), dtype=np.float32)
    _op_check(load_library().yolo_op_reorg(x.ctypes.data, n, h, w, c, stride, semantics, out.ctypes.data, device), "yolo_op_reorg")
    return out


def op_maxpool(x, size=2, stride=2, device=0):
    x = _f32(x); n, h, w, c = x.shape
    pad = (size - 1) // 2
    out = np.empty((n, (h + 2 * pad) // stride, (w + 2 * pad) // stride, c), dtype=np.float32)
    _op_check(load_library().yolo_op_maxpool(x.ctypes.data, n, h, w, c, size, stride, out.ctypes.data, device), "yolo_op_maxpool")
    return out


def op_resize_u8(img, size, post_scale=1.0, device=0):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    out = np.empty((size, size, 3), dtype=np.float32)
    _op_check(load_library().yolo_op_resize_u8(img.ctypes.data, img.shape[0], img.shape[1], size, post_scale, out.ctypes.data, device), "yolo_op_resize_u8")
    return out


def op_resize_cv2(img, out_hw, swap_rb=True, divisor=225.0, device=0):
    """yolo_op_resize_cv2: `cv2.resize(image.astype(float32)[, BGR -> RGB], (w, h)) / divisor` of one uint8 [h,w,3] image -> [oh,ow,3] float32."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    out = np.empty((int(out_hw[0]), int(out_hw[1]), 3), dtype=np.float32)
    _op_check(load_library().yolo_op_resize_cv2(img.ctypes.data, img.shape[0], img.shape[1], int(out_hw[0]), int(out_hw[1]), 1 if swap_rb else 0,
                                                float(divisor), out.ctypes.data, device), "yolo_op_resize_cv2")
    return out


def op_nms_detections(boxes_xywh, prob, objectness, thresh, by_objectness=False, device=0):
    """darknet do_nms_sort / do_nms_obj on arrays; returns (prob, objectness) after suppression."""
    b = _f32(boxes_xywh); p = _f32(prob).copy(); o = _f32(objectness).copy()
    n, classes = p.shape
    _op_check(load_library().yolo_op_nms_detections(b.ctypes.data, p.ctypes.data, o.ctypes.data, n, classes, thresh, 1 if by_objectness else 0, device), "yolo_op_nms_detections")
    return p, o


def op_detections_boxes(det, device=0):
    det = _f32(det); n, rows, attrs = det.shape
    out = np.empty_like(det)
    _op_check(load_library().yolo_op_detections_boxes(det.ctypes.data, n, rows, attrs, out.ctypes.data, device), "yolo_op_detections_boxes")
    return out


def op_decode(raw, anchors, classes, img_size, decode=DECODE_RATIO, region=False, device=0):
    raw = _f32(raw); n, g = raw.shape[0], raw.shape[1]
    anchors = _f32(anchors).reshape(-1)
    na = anchors.size // 2
    out = np.empty((n, g * g * na, 5 + classes), dtype=np.float32)
    _op_check(load_library().yolo_op_decode(raw.ctypes.data, n, g, na, classes, anchors.ctypes.data, img_size, decode,
                                            1 if region else 0, out.ctypes.data, device), "yolo_op_decode")
    return out


def op_postprocess(det, score_thr, iou_thr, max_out, nms_mode=NMS_TF, select_mode=SELECT_GT, image_hw=None, corners=False, device=0,
                   return_rows=False):
    """det [n,rows,5+C] fp32 rows (cx,cy,w,h,obj,cls..) -- or (x0,y0,x1,y1,obj,cls..) with corners=True.
    return_rows=True: -> (records, rows): the row of `det` every kept record was formed from."""
    det = _f32(det); n, rows, attrs = det.shape
    mode = nms_mode
    if corners:
        select_mode |= 0x100
    if image_hw is not None:
        mode |= (int(image_hw[0]) << 8) | (int(image_hw[1]) << 20)
    boxes = np.zeros((n, max_out), dtype=BOX_DTYPE); counts = np.zeros(n, dtype=np.int32)
    ridx = np.full((n, max_out), -1, dtype=np.int32) if return_rows else None
    _op_check(load_library().yolo_op_postprocess_rows(det.ctypes.data, n, rows, attrs, score_thr, iou_thr, max_out, mode, select_mode,
                                                      boxes.ctypes.data, counts.ctypes.data, ridx.ctypes.data if return_rows else None, device),
              "yolo_op_postprocess")
    recs = [boxes[i, :counts[i]].copy() for i in range(n)]
    if return_rows:
        return recs, [ridx[i, :counts[i]].copy() for i in range(n)]
    return recs


# ---- include/yolo_dist.h: the image-sharded detect step behind the C ABI ----
def shard_bounds(global_batch, world_size, rank):
    """yolo_shard_bounds: (lo, hi) of rank's contiguous slice of the global batch."""
    first, count = C.c_int(0), C.c_int(0)
    _op_check(load_library().yolo_shard_bounds(global_batch, world_size, rank, C.byref(first), C.byref(count)), "yolo_shard_bounds")
    return first.value, first.value + count.value


def dist_split_records(gathered, world_size, global_batch, max_out):
    """yolo_dist_split_records: gathered int32 [world, flat] (host) -> (boxes [global_batch, max_out] records, counts [global_batch])."""
    lib = load_library()
    per = -(-global_batch // world_size)
    g = np.ascontiguousarray(gathered, dtype=np.int32)
    if g.size != world_size * lib.yolo_dist_flat_words(per, max_out):
        raise YoloError("gathered buffer holds %d words, %d ranks x %d expected" % (g.size, world_size, lib.yolo_dist_flat_words(per, max_out)))
    boxes = np.zeros((global_batch, max_out), dtype=BOX_DTYPE)
    counts = np.zeros(global_batch, dtype=np.int32)
    _op_check(lib.yolo_dist_split_records(g.ctypes.data, world_size, global_batch, max_out, boxes.ctypes.data, counts.ctypes.data),
              "yolo_dist_split_records")
    return boxes, counts


def dist_unique_id():
    """yolo_dist_unique_id: the 128 bytes rank 0 hands to every rank's ShardedDetector."""
    buf = (C.c_uint8 * 128)()
    _op_check(load_library().yolo_dist_unique_id(buf), "yolo_dist_unique_id (RCCL)")
    return bytes(buf)


class ShardedDetector(object):
    """yolo_dist_*: this rank's Engine bound to a communicator of `world_size` ranks.  detect(images) takes THIS rank's slice
    (device tensor) and returns the whole batch's per-image record arrays, in global image order, on every rank."""

    def __init__(self, engine, world_size, rank, unique_id, global_batch, max_out=20):
        self.lib = load_library()
        self.engine = engine
        self.global_batch, self.max_out, self.world_size, self.rank = global_batch, max_out, world_size, rank
        self.lo, self.hi = shard_bounds(global_batch, world_size, rank)
        err = C.create_string_buffer(512)
        idbuf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        self.h = self.lib.yolo_dist_create(engine.ctx, world_size, rank, idbuf, None, global_batch, max_out, err, 512)
        if not self.h:
            raise YoloError("yolo_dist_create: " + err.value.decode())

    def detect(self, images, scale=1.0 / 255.0, score_thr=0.5, iou_thr=0.5, nms_mode=NMS_TF, select_mode=SELECT_GT):
        p, loc = _ptr(images)
        if self.hi > self.lo and (loc != DEVICE or int(images.shape[0]) != self.hi - self.lo):
            raise YoloError("rank %d serves images [%d, %d): a device tensor of that many images is needed" % (self.rank, self.lo, self.hi))
        self._images = images                       # the graph replays from this buffer: keep it alive
        fmt = IMG_U8
        if images is not None:                      # (a rank whose slice is empty -- more ranks than images -- only takes part in the gather)
            self.engine._order_after_producer(images)
            fmt = IMG_U8 if str(images.dtype).endswith("uint8") else IMG_F32
        boxes = np.zeros((self.global_batch, self.max_out), dtype=BOX_DTYPE)
        counts = np.zeros(self.global_batch, dtype=np.int32)
        self.engine._check(self.lib.yolo_dist_detect(self.h, p, fmt, scale, score_thr, iou_thr, nms_mode, select_mode,
                                                      boxes.ctypes.data, counts.ctypes.data), "yolo_dist_detect")
        return [boxes[i, :counts[i]].copy() for i in range(self.global_batch)]

    def close(self):
        """Destroy BEFORE the engine (include/yolo_dist.h): a later step would use a dead context.  (yolo_dist_destroy itself only needs the
        device number, which the handle keeps, so the order of two close() calls cannot crash.)"""
        if self.h:
            self.lib.yolo_dist_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:      # noqa: BLE001
            pass
