"""CPU restatement (numpy, fp32) of the reference's YOLO inference hot path.

**TEST INFRASTRUCTURE ONLY** -- this module is the parity oracle.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it; the product
(`yolo_tensorflow_amd/`) never does and fails loudly when its HIP library is missing.

Parity status: the reference's TF graph cannot run here (TensorFlow absent, its
kernels un-vendored, TF 1.5-1.8 per V1/README.md:3, V3/README.md:5), so the TF-only ops are
restated from their published closed forms; every function that *can* be pinned is pinned by
tests/test_oracle_*.py:
  * conv / BN / leaky / maxpool / shortcut / route / yolo+region decode / darknet NMS, and the whole
    network in `semantics="darknet"`: against the reference's own C code compiled into oracle/_ref
    (oracle/Makefile), fixtures in tests/golden/ (tools/make_golden.py);
  * numpy NMS `non_max_suppression`/`_iou` (V3/yolo_v3.py:350-420) and V2 `postprocess`/`bboxes_*`
    (V2/utils.py:30-187): bit-for-bit against the reference functions imported in the build
    container, golden vectors committed;
  * `resize_bilinear` (legacy), `_upsample`, `space_to_depth`, `tf.image.non_max_suppression`:
    **parity unpinned** by any reference artefact -- restated from the TF-1.x kernel semantics and
    checked against hand-derived closed forms only;
  * `to_fp8_e4m3`, `fp8_scheme_forward`, `fp8_calibrate_scales` (BASELINE config 5): **parity unpinned** --
    the reference has no reduced-precision path at all, so the quantisation scheme is defined by this repo
     and restated here; the e4m3 rounding itself is checked exhaustively against the OCP
    value table (tests/test_oracle_closed_forms.py);
  * `split_f16`, `forward_f16x2` (the device's split-fp16 configuration): **parity unpinned** for the same reason -- the
    scheme is this repo's; what anchors it is the fp32 forward above (pinned by the compiled reference): the scheme must, and does, stay
    within 1e-5 of it (tests/test_oracle_closed_forms.py), and the device's boxes are held to IoU >= 0.999 against that fp32 forward.

Layout convention: activations NHWC float32 (the reference's `data_format='NHWC'` path), conv
weights HWIO, exactly as the TF graph holds them after `load_weights`.
"""
import numpy as np

LEAKY = 0.1          # V3/yolo_v3.py:10
BN_EPS_TF = 1e-5     # V3/yolo_v3.py:9


# ----------------------------------------------------------------------------------------------
# numeric helpers
# ----------------------------------------------------------------------------------------------
def to_bf16(x):
    """Round float32 -> bfloat16 (round-to-nearest-even) and return as float32 (for emulating the
    device's storage precision; not part of the reference)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32)
    r = ((u >> 16) & 1) + np.uint32(0x7FFF)
    out = ((u + r) & np.uint32(0xFFFF0000)).view(np.float32)
    return np.where(np.isnan(x), x, out)


def to_f16(x):
    """Round float32 -> IEEE fp16 (RNE, saturating at +-65504 as the device's conversions do) -> float32: the YOLO_FP16 storage type."""
    return np.clip(np.asarray(x, np.float32), -65504.0, 65504.0).astype(np.float16).astype(np.float32)


def to_fp8_e4m3(x):
    """Round float32 -> OCP e4m3 (e4m3fn: 4 exponent bits, bias 7, 3 mantissa bits, no infinities, max 448),
    round-to-nearest-even, saturating at +-448, returned as float32 values on the e4m3 grid.  Emulates the
    device's fp8 storage (BASELINE config 5); not part of the reference, which has no reduced-precision path."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    a = np.minimum(np.abs(x), np.float32(448.0))
    with np.errstate(divide="ignore"):
        e = np.floor(np.log2(np.where(a > 0, a, np.float32(1.0)))).astype(np.int32)
    # log2 of a float32 just below a power of two can round up: fix with an exact comparison
    e = np.where(np.ldexp(np.float32(1.0), e) > a, e - 1, e)
    e = np.maximum(e, -6)                                  # subnormals share the quantum of the first binade (2^-9)
    quantum = np.ldexp(np.float32(1.0), e - 3).astype(np.float32)
    q = np.rint(a / quantum).astype(np.float32)            # np.rint rounds half to even; a / quantum is exact
    out = np.copysign(q * quantum, x).astype(np.float32)
    return np.where(np.isnan(x), x, out)


def fp8_scheme_forward(secs, params, x, scales=None, semantics="tf", teacher=None):
    """Emulation of the device's fp8 configuration (yolo_tensorflow_amd/csrc, DESIGN.md "fp8 scheme") in numpy:
      * every stored activation is an e4m3 code with a per-layer scale (value = code * scale; `scales[i]`, default 1);
        data-movement layers inherit their input's scale; the image and the first conv's filters are bf16;
      * conv i > 0: filters (BN folded) are multiplied by the scale of each input channel, then quantised per output
        channel c to e4m3 with osc[c] = max|w| / 448; acc = sum(code_w * code_x) in fp32; v = acc * osc[c] + bias;
        leaky; v rounded to bf16 (the device stages the tile through LDS in bf16); head convs keep v in fp32;
      * stored code = e4m3(v * (1 / scale)); shortcut = e4m3((a * s_a + b * s_b) * (1 / s_out)).
    x: [N,S,S,3] float32 in 0..1.  Returns (heads, outs) like forward(); outs hold real values (code * scale).
    teacher: optional per-layer list of real-valued tensors (the device's own layer outputs).  Where given, layer i's
    result is still computed and returned, but the layers after it consume teacher[i] instead -- every layer is then
    checked on identical inputs, so one-ulp rounding flips cannot compound through the depth of the network."""
    f32 = np.float32
    layers = secs[1:]
    NL = len(layers)
    user = np.ones(NL, np.float32) if scales is None else np.asarray(scales, np.float32)
    codes = [None] * NL; chs = [None] * NL       # e4m3 code values, per-channel scale vectors
    # mixed plans: a [convolutional] section with `yolo_store=bf16` keeps its output -- and what is derived from it without arithmetic --
    # in bf16 (scale 1; `codes` then holds the bf16 values themselves); a conv whose input is stored in bf16 runs as the first conv does
    # (bf16 filters, bf16 MFMA)
    is16 = [False] * NL
    outs, heads = [], []
    x = to_bf16(np.asarray(x, dtype=np.float32))
    ci = 0
    for i, s in enumerate(layers):
        t = s["type"]
        if t == "convolutional":
            p = params[ci]; ci += 1
            st = int(s.get("stride", 1))
            is_head = i + 1 < NL and layers[i + 1]["type"] in ("yolo", "region")
            w, b = fold_bn(p, mode="darknet" if semantics == "darknet" else "tf")
            if i == 0:
                y = conv2d_nhwc(x, to_bf16(w), st) + b
            elif is16[i - 1]:
                y = conv2d_nhwc(codes[i - 1], to_bf16(w), st) + b
            else:
                cx, sx = codes[i - 1], chs[i - 1]
                weff = (w * sx[None, None, :, None]).astype(np.float32)             # HWIO
                amax = np.abs(weff).max(axis=(0, 1, 2))
                osc = np.where(amax > 0, amax / f32(448.0), f32(1.0)).astype(np.float32)
                wq = to_fp8_e4m3(weff / osc[None, None, None, :])
                acc = conv2d_nhwc(cx, wq, st)
                y = (acc.astype(np.float64) * osc.astype(np.float64) + b.astype(np.float64)).astype(np.float32)   # device: one fma
            act = s.get("activation", "logistic")
            if act == "leaky":
                y = leaky_relu(y)
            elif act != "linear":
                raise ValueError(act)
            y = y.astype(np.float32)
            if is_head:
                outs.append(y); continue
            if s.get("yolo_store", "fp8") == "bf16":
                is16[i] = True
                codes[i] = to_bf16(y); chs[i] = np.ones(y.shape[-1], np.float32)
                outs.append(codes[i])
                if teacher is not None and teacher[i] is not None:
                    codes[i] = to_bf16(np.asarray(teacher[i], np.float32))
                continue
            inv = f32(1.0) / user[i]
            codes[i] = to_fp8_e4m3(to_bf16(y) * inv); chs[i] = np.full(codes[i].shape[-1], user[i], np.float32)
        elif t == "shortcut":
            f = int(s["from"]); f = f if f >= 0 else i + f
            if is16[i - 1] != is16[f]:
                raise ValueError("layer %d: shortcut operands stored in different types" % i)
            if is16[f]:
                is16[i] = True
                codes[i] = to_bf16(codes[i - 1] + codes[f]); chs[i] = np.ones(codes[i].shape[-1], np.float32)
                outs.append(codes[i])
                if teacher is not None and teacher[i] is not None:
                    codes[i] = to_bf16(np.asarray(teacher[i], np.float32))
                continue
            sa, sb = chs[i - 1][0], chs[f][0]
            inv = f32(1.0) / user[i]
            codes[i] = to_fp8_e4m3(((codes[i - 1] * sa).astype(np.float32) + (codes[f] * sb).astype(np.float32)).astype(np.float32) * inv)
            chs[i] = np.full(codes[i].shape[-1], user[i], np.float32)
        elif t == "route":
            ls = [int(v) for v in s["layers"].split(",")]
            ls = [l if l >= 0 else i + l for l in ls]
            if len({is16[l] for l in ls}) > 1:
                raise ValueError("layer %d: route inputs stored in different types" % i)
            is16[i] = is16[ls[0]]
            codes[i] = np.concatenate([codes[l] for l in ls], axis=-1) if len(ls) > 1 else codes[ls[0]]
            chs[i] = np.concatenate([chs[l] for l in ls]) if len(ls) > 1 else chs[ls[0]]
        elif t == "upsample":
            is16[i] = is16[i - 1]
            rnd = to_bf16 if is16[i] else to_fp8_e4m3
            codes[i] = rnd(upsample_tf(codes[i - 1])) if semantics == "tf" else upsample_nearest(codes[i - 1], int(s.get("stride", 2)))
            chs[i] = chs[i - 1]
        elif t == "maxpool":
            st = int(s.get("stride", 1)); k = int(s.get("size", st))
            codes[i] = max_pool(codes[i - 1], k, st, int(s.get("padding", (k - 1) // 2))); chs[i] = chs[i - 1]; is16[i] = is16[i - 1]
        elif t == "reorg":
            st = int(s.get("stride", 1))
            codes[i] = space_to_depth(codes[i - 1], st) if semantics == "tf" else reorg_darknet(codes[i - 1], st)
            chs[i] = np.full(codes[i].shape[-1], chs[i - 1][0], np.float32); is16[i] = is16[i - 1]
        elif t in ("yolo", "region"):
            heads.append((s, outs[i - 1])); outs.append(None); continue
        else:
            raise ValueError(t)
        outs.append((codes[i] * chs[i]).astype(np.float32))
        if teacher is not None and teacher[i] is not None and not (t == "route" and len(ls) > 1):
            codes[i] = to_bf16(np.asarray(teacher[i], np.float32)) if is16[i] else to_fp8_e4m3(np.asarray(teacher[i], np.float32) / chs[i])
    return heads, outs


def fp8_calibrate_scales(secs, outs_fp32, headroom=2.0):
    """Per-layer power-of-two activation scales from a full-precision run's per-layer outputs (forward(collect=True)):
    the largest |value| of conv / shortcut outputs maps to at most 448 / headroom.  Other layers get 1 (ignored)."""
    layers = secs[1:]
    sc = np.ones(len(layers), np.float32)
    for i, s in enumerate(layers):
        if s["type"] in ("convolutional", "shortcut") and outs_fp32[i] is not None:
            m = float(np.abs(outs_fp32[i]).max())
            if m > 0:
                sc[i] = np.float32(2.0 ** np.ceil(np.log2(m * headroom / 448.0)))
    return sc


def sigmoid(x):
    x = np.asarray(x, dtype=np.float32)
    return (np.float32(1) / (np.float32(1) + np.exp(-x, dtype=np.float32))).astype(np.float32)


def leaky_relu(x, alpha=LEAKY):
    # tf.nn.leaky_relu = max(alpha*x, x)  (V3/yolo_v3.py:229; V2/model_darknet19_slim.py:111-116)
    return np.maximum(np.float32(alpha) * x, x)


# ----------------------------------------------------------------------------------------------
# row L: darknet .weights <-> tensors          (V3/yolo_v3.py:270-326, DN/parser.c:1163-1239)
# ----------------------------------------------------------------------------------------------
def parse_cfg(text):
    """Darknet INI-style topology -> list of {'type':..., key: str}.  DN/parser.c:730-875."""
    secs = []
    for line in text.splitlines():
        line = line.strip()
        if not line or line[0] in "#;":
            continue
        if line.startswith("["):
            secs.append({"type": line.strip("[]").strip()})
        else:
            k, v = line.split("=", 1)
            secs[-1][k.strip()] = v.strip()
    return secs


def conv_layers(secs):
    """(index_in_layers, filters, size, stride, bn, act, cin) for every conv in cfg order."""
    out = []
    for i, s, shp in _walk_shapes(secs):
        if s["type"] == "convolutional":
            out.append(dict(idx=i, filters=int(s["filters"]), size=int(s["size"]),
                            stride=int(s.get("stride", 1)), bn=int(s.get("batch_normalize", 0)),
                            act=s.get("activation", "logistic"), cin=shp["cin"]))
        elif s["type"] == "connected":       # parameters are stored like a bias conv's: biases, then weights [output][inputs]
            out.append(dict(idx=i, filters=int(s["output"]), size=1, stride=1, bn=0,
                            act=s.get("activation", "logistic"), cin=shp["cin"]))
        elif s["type"] == "local":           # DN/parser.c:1315-1320: biases [filters][out_h][out_w], then weights [location][filter][c][kh][kw]
            out.append(dict(idx=i, filters=int(s["filters"]), size=int(s["size"]), stride=int(s.get("stride", 1)), bn=0,
                            act=s.get("activation", "logistic"), cin=shp["cin"], local=True, locations=shp["H"] * shp["W"]))
    return out


def _walk_shapes(secs):
    net = secs[0]
    H, W, C = int(net["height"]), int(net["width"]), int(net["channels"])
    shapes = []
    for i, s in enumerate(secs[1:]):
        t = s["type"]
        cin = C
        if t == "convolutional":
            k, st = int(s["size"]), int(s.get("stride", 1))
            pad = k // 2 if int(s.get("pad", 0)) else int(s.get("padding", 0))
            H, W, C = (H + 2 * pad - k) // st + 1, (W + 2 * pad - k) // st + 1, int(s["filters"])
        elif t == "maxpool":
            st = int(s.get("stride", 1)); k = int(s.get("size", st))
            pad = int(s.get("padding", (k - 1) // 2))
            H, W = (H + 2 * pad) // st, (W + 2 * pad) // st
        elif t == "upsample":
            st = int(s.get("stride", 2)); H, W = H * st, W * st
        elif t == "reorg":
            st = int(s.get("stride", 1)); H, W, C = H // st, W // st, C * st * st
        elif t == "route":
            ls = [int(x) for x in s["layers"].split(",")]
            ls = [l if l >= 0 else i + l for l in ls]
            H, W = shapes[ls[0]][0], shapes[ls[0]][1]
            C = sum(shapes[l][2] for l in ls)
        elif t == "connected":               # DN/connected_layer.c:151 / slim.flatten + fully_connected (V1/YOLO_V1_Inference.py:198-206)
            cin = H * W * C
            H, W, C = 1, 1, int(s["output"])
        elif t == "local":                   # DN/local_layer.c:10-24: `pad` is a flag here (and the im2col pad amount, DN/local_layer.c:103)
            k, st, pad = int(s["size"]), int(s.get("stride", 1)), int(s.get("pad", 0))
            H, W, C = ((H - 1) if pad else (H - k)) // st + 1, ((W - 1) if pad else (W - k)) // st + 1, int(s["filters"])
        elif t in ("shortcut", "dropout"):
            pass
        elif t in ("yolo", "region", "detection"):
            pass
        else:
            raise ValueError("unsupported layer type " + t)
        shapes.append((H, W, C))
        yield i, s, dict(cin=cin, H=H, W=W, C=C)


def read_darknet_weights(path, secs, header_ints=None):
    """Return per-conv list of dicts {bias|beta,gamma,mean,var, w_hwio}.

    header: 5 x int32 for v3 / v3-tiny (V3/yolo_v3.py:278), 4 x int32 for v2 / tiny-voc / v1-tiny
    (D2T/YOLO_V2_convert...py:351); when `header_ints` is None apply darknet's own rule
    (DN/parser.c:1259-1265): major*10+minor >= 2 -> 64-bit `seen`."""
    raw = np.fromfile(path, dtype=np.int32, count=3)
    if header_ints is None:
        header_ints = 5 if (raw[0] * 10 + raw[1]) >= 2 else 4
    with open(path, "rb") as f:
        np.fromfile(f, dtype=np.int32, count=header_ints)
        flat = np.fromfile(f, dtype=np.float32)
    return unflatten_weights(flat, secs)


def unflatten_weights(flat, secs):
    ptr = 0
    out = []
    for c in conv_layers(secs):
        n, k, cin = c["filters"], c["size"], c["cin"]
        p = {}
        if c.get("local"):
            loc = c["locations"]
            p["bias_fl"] = flat[ptr:ptr + n * loc].reshape(n, loc).copy(); ptr += n * loc            # [filter][location]
            cnt = loc * n * cin * k * k
            p["w_local"] = flat[ptr:ptr + cnt].reshape(loc, n, cin, k, k).copy(); ptr += cnt         # [location][filter][c][kh][kw]
            out.append(p)
            continue
        if c["bn"]:
            # file order: biases(beta), scales(gamma), rolling_mean, rolling_variance
            for name in ("beta", "gamma", "mean", "var"):
                p[name] = flat[ptr:ptr + n].copy(); ptr += n
        else:
            p["bias"] = flat[ptr:ptr + n].copy(); ptr += n
        cnt = n * cin * k * k
        w = flat[ptr:ptr + cnt].reshape(n, cin, k, k); ptr += cnt       # OIHW
        p["w_hwio"] = np.ascontiguousarray(np.transpose(w, (2, 3, 1, 0)))  # V3/yolo_v3.py:319-321
        out.append(p)
    if ptr != flat.size:
        raise ValueError(f"weights stream has {flat.size} floats, topology consumes {ptr}")
    return out


def fold_bn(p, eps=BN_EPS_TF, mode="tf"):
    """W' = W*gamma/sqrt(var+eps), b' = beta - mean*gamma/sqrt(var+eps)  (SURVEY 8a row C).
    fp32 arithmetic -- this is the load-time transformation the product applies.  mode 'darknet': the scale of the reference's CPU
    normalize, gamma / (sqrt(var) + 1e-6) (DN/blas.c:147-158), which the product uses under its darknet semantics."""
    if "bias" in p:
        return p["w_hwio"].astype(np.float32), p["bias"].astype(np.float32)
    if mode == "darknet":
        s = (p["gamma"] / (np.sqrt(p["var"]) + np.float32(.000001))).astype(np.float32)
    else:
        s = (p["gamma"] / np.sqrt(p["var"] + np.float32(eps))).astype(np.float32)
    return (p["w_hwio"] * s[None, None, None, :]).astype(np.float32), \
           (p["beta"] - p["mean"] * s).astype(np.float32)


# ----------------------------------------------------------------------------------------------
# row P: input processing
# ----------------------------------------------------------------------------------------------
def resize_bilinear_legacy(img, out_h, out_w):
    """TF-1.x `tf.image.resize_images` / `resize_bilinear(align_corners=False)`: src = dst*(in/out),
    no half-pixel offset, upper index clamped; value = top + (bottom-top)*y_lerp with
    top = tl + (tr-tl)*x_lerp (resize_bilinear_op.cc compute_lerp).  img [H,W,C] or [N,H,W,C] f32."""
    img = np.asarray(img, dtype=np.float32)
    squeeze = img.ndim == 3
    if squeeze:
        img = img[None]
    N, H, W, C = img.shape
    hs = np.float32(H) / np.float32(out_h)
    ws = np.float32(W) / np.float32(out_w)
    ys = (np.arange(out_h, dtype=np.float32) * hs).astype(np.float32)
    xs = (np.arange(out_w, dtype=np.float32) * ws).astype(np.float32)
    y0 = np.floor(ys).astype(np.int64); y1 = np.minimum(y0 + 1, H - 1); yl = (ys - y0).astype(np.float32)
    x0 = np.floor(xs).astype(np.int64); x1 = np.minimum(x0 + 1, W - 1); xl = (xs - x0).astype(np.float32)
    tl = img[:, y0][:, :, x0]; tr = img[:, y0][:, :, x1]
    bl = img[:, y1][:, :, x0]; br = img[:, y1][:, :, x1]
    xl = xl[None, None, :, None]; yl = yl[None, :, None, None]
    top = tl + (tr - tl) * xl
    bot = bl + (br - bl) * xl
    out = (top + (bot - top) * yl).astype(np.float32)
    return out[0] if squeeze else out


def resize_cv2_linear(img, out_h, out_w):
    """`cv2.resize(img_float32, (out_w, out_h))`, INTER_LINEAR, as V2/utils.py:19 calls it -- restated from OpenCV's published CV_32F linear
    resize (imgproc/resize.cpp): half-pixel centres, fx = float32((dx + .5) * scale - .5) with scale = 1 / (dst / src) in double,
    sx = floor(fx), fx -= sx; sx < 0 -> (0, 0); sx >= src - 1 -> (src - 1, 0); a horizontal pass S[sx] * (1 - fx) + S[sx + 1] * fx, then the
    vertical pass R0 * (1 - fy) + R1 * fy, float32 operations.  cv2 is not installed: PARITY UNPINNED (closed forms in
    tests/test_oracle_closed_forms.py).  img [H,W,C] float32."""
    img = np.asarray(img, dtype=np.float32); H, W = img.shape[:2]

    def taps(n_out, n_in):
        scale = 1.0 / (np.float64(n_out) / np.float64(n_in))
        f = ((np.arange(n_out, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64); f = (f - s.astype(np.float32)).astype(np.float32)
        lo = s < 0; f[lo] = 0; s[lo] = 0
        hi = s >= n_in - 1; f[hi] = 0; s[hi] = n_in - 1
        return s, np.minimum(s + 1, n_in - 1), (np.float32(1) - f).astype(np.float32), f
    sx, sx1, a0, a1 = taps(out_w, W); sy, sy1, b0, b1 = taps(out_h, H)
    a0 = a0[None, :, None]; a1 = a1[None, :, None]
    rows = (img[:, sx] * a0 + img[:, sx1] * a1).astype(np.float32)                   # horizontal pass on every source row
    return (rows[sy] * b0[:, None, None] + rows[sy1] * b1[:, None, None]).astype(np.float32)


def v2_preprocess_image(image_bgr_u8, image_size=(416, 416)):
    """V2/utils.py:13-27: float32 copy, BGR -> RGB, cv2.resize to image_size = (width, height), / 225.0 (sic), batch axis."""
    x = np.asarray(image_bgr_u8).astype(np.float32)[:, :, ::-1]
    return (resize_cv2_linear(x, image_size[1], image_size[0]) / np.float32(225.0))[None]


def input_process(image_u8, size):
    """D2T/YOLO_V3_convert...py:106-111: uint8 HWC -> float /255.0 -> bilinear stretch -> [1,S,S,3]."""
    x = image_u8.astype(np.float32) / np.float32(255.0)
    return resize_bilinear_legacy(x, size, size)[None]


# ----------------------------------------------------------------------------------------------
# rows C, Cb, M, U, R, Rt, B: operators
# ----------------------------------------------------------------------------------------------
def conv2d_nhwc(x, w_hwio, stride=1, pad=None):
    """Cross-correlation, zero padding `pad` each side (k//2 by default == TF SAME at stride 1 and
    `_fixed_padding`+VALID at stride 2, V3/yolo_v3.py:47-51,63-90; == darknet pad=1,
    DN/parser.c:186-187).  im2col + sgemm, fp32."""
    x = np.asarray(x, dtype=np.float32)
    k = w_hwio.shape[0]
    if pad is None:
        pad = k // 2
    N, H, W, C = x.shape
    O = w_hwio.shape[3]
    Ho = (H + 2 * pad - k) // stride + 1
    Wo = (W + 2 * pad - k) // stride + 1
    wm = w_hwio.reshape(k * k * C, O).astype(np.float32)
    out = np.empty((N, Ho, Wo, O), dtype=np.float32)
    for n in range(N):
        xp = np.pad(x[n], ((pad, pad), (pad, pad), (0, 0))) if pad else x[n]
        if k == 1 and stride == 1:
            cols = xp.reshape(-1, C)
        else:
            win = np.lib.stride_tricks.sliding_window_view(xp, (k, k), axis=(0, 1))  # [H',W',C,k,k]
            win = win[::stride, ::stride][:Ho, :Wo]
            cols = np.ascontiguousarray(win.transpose(0, 1, 3, 4, 2)).reshape(Ho * Wo, k * k * C)
        out[n] = (cols @ wm).reshape(Ho, Wo, O)
    return out


def batch_norm(x, p, mode="tf"):
    """mode 'tf': slim.batch_norm inference, gamma*(x-mean)/sqrt(var+1e-5)+beta (V3/yolo_v3.py:218-224).
    mode 'darknet_cpu': (x-mean)/(sqrt(var)+1e-6)*gamma+beta (DN/blas.c:147-158, batchnorm_layer.c:135-155)."""
    if mode == "tf":
        return ((x - p["mean"]) / np.sqrt(p["var"] + np.float32(BN_EPS_TF)) * p["gamma"] + p["beta"]).astype(np.float32)
    return ((x - p["mean"]) / (np.sqrt(p["var"]) + np.float32(.000001)) * p["gamma"] + p["beta"]).astype(np.float32)


def max_pool(x, size, stride, pad=0):
    """slim.max_pool2d(2,'VALID') and (2, stride=1, 'SAME') (V2/model_darknet19_slim.py:141; D2T V3_Tiny
    :445) == DN/maxpool_layer.c:79-111 with this fork's geometry out=(in+2*pad)/stride, window
    origin -pad, out-of-range = -inf."""
    N, H, W, C = x.shape
    Ho, Wo = (H + 2 * pad) // stride, (W + 2 * pad) // stride
    need_h = (Ho - 1) * stride + size - pad
    need_w = (Wo - 1) * stride + size - pad
    xp = np.pad(x, ((0, 0), (pad, max(0, need_h - H)), (pad, max(0, need_w - W)), (0, 0)),
                constant_values=-np.inf)
    out = np.full((N, Ho, Wo, C), -np.inf, dtype=np.float32)
    for dy in range(size):
        for dx in range(size):
            out = np.maximum(out, xp[:, dy:dy + Ho * stride:stride, dx:dx + Wo * stride:stride, :])
    return out


def upsample_tf(x):
    """`_upsample` (V3/yolo_v3.py:162-192): SYMMETRIC pad 1 -> legacy resize_bilinear to (2H+4,2W+4)
    -> crop [2:-2].  Evaluated literally (pad, resize, crop), not through the closed form."""
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1), (0, 0)), mode="symmetric")
    N, H, W, C = x.shape
    r = resize_bilinear_legacy(xp, 2 * H + 4, 2 * W + 4)
    return np.ascontiguousarray(r[:, 2:-2, 2:-2, :])


def upsample_tf_closed_form(x):
    """SURVEY 8a row U closed form: out[2i]=in[i]; out[2i+1]=in[i]+(in[min(i+1,H-1)]-in[i])*0.5,
    separably (x first, then y) -- what the device kernel evaluates."""
    def up1(a, axis):
        a = np.moveaxis(a, axis, 0)
        nxt = np.concatenate([a[1:], a[-1:]], axis=0)
        odd = a + (nxt - a) * np.float32(0.5)
        out = np.empty((2 * a.shape[0],) + a.shape[1:], dtype=np.float32)
        out[0::2] = a; out[1::2] = odd
        return np.moveaxis(out, 0, axis)
    return up1(up1(x.astype(np.float32), 2), 1)


def upsample_nearest(x, stride=2):
    """darknet upsample_cpu (DN/blas.c:334-350), scale 1."""
    return np.repeat(np.repeat(x, stride, axis=1), stride, axis=2)


def space_to_depth(x, s=2):
    """tf.space_to_depth NHWC (V2/model_darknet19_slim.py:44-45): out[n,h,w,(dy*s+dx)*C+c]=in[n,h*s+dy,w*s+dx,c]."""
    N, H, W, C = x.shape
    x = x.reshape(N, H // s, s, W // s, s, C).transpose(0, 1, 3, 2, 4, 5)
    return np.ascontiguousarray(x.reshape(N, H // s, W // s, s * s * C))


def reorg_darknet(x, s=2):
    """darknet forward_reorg_layer -> reorg_cpu(forward=0) (DN/reorg_layer.c:91-111, DN/blas.c:9-30),
    restated on NCHW flat buffers then returned NHWC."""
    N, H, W, C = x.shape
    xin = np.ascontiguousarray(x.transpose(0, 3, 1, 2)).reshape(N, -1)
    out = np.empty_like(xin)
    out_c = C // (s * s)
    k, j, i = np.meshgrid(np.arange(C), np.arange(H), np.arange(W), indexing="ij")
    in_index = i + W * (j + H * k)
    c2 = k % out_c; off = k // out_c
    w2 = i * s + off % s; h2 = j * s + off // s
    out_index = w2 + W * s * (h2 + H * s * c2)
    out[:, in_index.ravel()] = xin[:, out_index.ravel()]
    return np.ascontiguousarray(out.reshape(N, C * s * s, H // s, W // s).transpose(0, 2, 3, 1))


# ----------------------------------------------------------------------------------------------
# rows D3, X, S, D2, D1: decode
# ----------------------------------------------------------------------------------------------
def detection_layer_pixel(pred, anchors, img_size):
    """`_detection_layer` V3/yolo_v3.py:111-159 (NHWC branch).  pred [N,g,g,A*(5+C)] raw conv output.
    Returns [N, g*g*A, 5+C] = (cx,cy,w,h in input pixels, obj, cls...).  True division at :131."""
    N, gy, gx, ch = pred.shape
    A = len(anchors); attrs = ch // A
    p = pred.reshape(N, gy * gx * A, attrs).astype(np.float32)
    stride = (img_size[0] // gy, img_size[1] // gx)
    anc = np.array([(a[0] / stride[0], a[1] / stride[1]) for a in anchors], dtype=np.float32)
    centers = sigmoid(p[..., 0:2]); conf = sigmoid(p[..., 4:5])
    a, b = np.meshgrid(np.arange(gy, dtype=np.float32), np.arange(gx, dtype=np.float32))
    xy = np.concatenate([a.reshape(-1, 1), b.reshape(-1, 1)], axis=-1)
    xy = np.tile(xy, [1, A]).reshape(1, -1, 2)
    centers = (centers + xy) * np.array(stride, dtype=np.float32)
    sizes = np.exp(p[..., 2:4]) * np.tile(anc, [gy * gx, 1]) * np.array(stride, dtype=np.float32)
    cls = sigmoid(p[..., 5:])
    return np.concatenate([centers, sizes, conf, cls], axis=-1).astype(np.float32)


def detection_layer_ratio(pred, anchors, img_size):
    """`_ratio_detection_layer` V3/YOLOV3.py:168-238 == D2T `_detection_layer` :363-433: normalised
    (cx,cy,w,h) = (sigmoid+cell)/g, exp*anchor/stride/g."""
    N, gy, gx, ch = pred.shape
    A = len(anchors); attrs = ch // A
    p = pred.reshape(N, gy * gx * A, attrs).astype(np.float32)
    a, b = np.meshgrid(np.arange(gy, dtype=np.float32), np.arange(gx, dtype=np.float32))
    xy = np.concatenate([a.reshape(-1, 1), b.reshape(-1, 1)], axis=-1)
    xy = np.tile(xy, [1, A]).reshape(1, -1, 2)
    grid = np.array([gy, gx], dtype=np.float32)
    centers = (sigmoid(p[..., 0:2]) + xy) / grid
    stride = (img_size[0] // gy, img_size[1] // gx)
    anc = np.array([(1.0 * a_[0] / stride[0], 1.0 * a_[1] / stride[1]) for a_ in anchors], dtype=np.float32)
    sizes = np.exp(p[..., 2:4]) * np.tile(anc, [gy * gx, 1]) / grid
    return np.concatenate([centers, sizes, sigmoid(p[..., 4:5]), sigmoid(p[..., 5:])], axis=-1).astype(np.float32)


def detections_boxes(det):
    """V3/yolo_v3.py:329-347: (cx,cy,w,h) -> (x0,y0,x1,y1), w/2 form."""
    cx, cy, w, h = det[..., 0:1], det[..., 1:2], det[..., 2:3], det[..., 3:4]
    w2 = w / np.float32(2); h2 = h / np.float32(2)
    return np.concatenate([cx - w2, cy - h2, cx + w2, cy + h2, det[..., 4:]], axis=-1).astype(np.float32)


def select_threshold(det, score_threshold):
    """V3/YOLOV3.py:347-362 for ONE image ([rows,5+C]): corners via w*0.5, score=obj*cls, argmax/max,
    strict `>`; order-preserving boolean_mask.  Returns boxes[x0,y0,x1,y1], scores, classes, row idx."""
    cx, cy, w, h = det[:, 0], det[:, 1], det[:, 2], det[:, 3]
    w2 = w * np.float32(0.5); h2 = h * np.float32(0.5)
    boxes = np.stack([cx - w2, cy - h2, cx + w2, cy + h2], axis=-1).astype(np.float32)
    sc = (det[:, 4:5] * det[:, 5:]).astype(np.float32)
    label = np.argmax(sc, axis=-1).astype(np.int32)
    smax = np.max(sc, axis=-1)
    m = smax > np.float32(score_threshold)
    return boxes[m], smax[m], label[m], np.nonzero(m)[0]


def region_decode(pred, anchors, num_class):
    """V2 `decode` V2/decode.py:13-47: [N,H,W,A*(5+C)] -> bboxes [N,HW,A,4] (x0,y0,x1,y1 normalised),
    obj [N,HW,A], class_probs [N,HW,A,C] (softmax)."""
    N, H, W, ch = pred.shape
    A = len(anchors)
    d = pred.reshape(N, H * W, A, num_class + 5).astype(np.float32)
    xy = sigmoid(d[..., 0:2]); wh = np.exp(d[..., 2:4]); obj = sigmoid(d[..., 4])
    z = d[..., 5:]
    e = np.exp(z - z.max(axis=-1, keepdims=True))
    cls = (e / e.sum(axis=-1, keepdims=True)).astype(np.float32)
    xc, yc = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32))
    xc = xc.reshape(1, -1, 1); yc = yc.reshape(1, -1, 1)
    anc = np.asarray(anchors, dtype=np.float32)
    bx = (xc + xy[..., 0]) / np.float32(W); by = (yc + xy[..., 1]) / np.float32(H)
    bw = (anc[:, 0] * wh[..., 0]) / np.float32(W); bh = (anc[:, 1] * wh[..., 1]) / np.float32(H)
    two = np.float32(2)
    boxes = np.stack([bx - bw / two, by - bh / two, bx + bw / two, by + bh / two], axis=3).astype(np.float32)
    return boxes, obj.astype(np.float32), cls


def region_select(boxes, obj, cls, threshold):
    """V2/postprocess.py:49-64 for one image: score = obj*cls, argmax/max, `>=` threshold."""
    C = cls.shape[-1]
    sc = (obj[..., None] * cls).reshape(-1, C)
    b = boxes.reshape(-1, 4)
    label = np.argmax(sc, axis=1).astype(np.int32); smax = np.max(sc, axis=1)
    m = smax >= np.float32(threshold)
    return b[m], smax[m], label[m], np.nonzero(m)[0]


def v1_decode(predicts, S=7, B=2, C=20, threshold=0.2):
    """V1 `_build_detector` V1/YOLO_V1_Inference.py:213-258 (before NMS), batch element 0 only:
    layout [cls S*S*C | conf S*S*B | boxes S*S*B*4]; x=(bx+col)/S, w=bw^2; score=conf*cls; `>=`."""
    p = np.asarray(predicts, dtype=np.float32)
    idx1 = S * S * C; idx2 = idx1 + S * S * B
    cls = p[0, :idx1].reshape(S, S, C)
    conf = p[0, idx1:idx2].reshape(S, S, B)
    boxes = p[0, idx2:].reshape(S, S, B, 4)
    x_off = np.transpose(np.reshape(np.array([np.arange(S)] * S * B, dtype=np.float32), [B, S, S]), [1, 2, 0])
    y_off = np.transpose(x_off, [1, 0, 2])
    boxes = np.stack([(boxes[..., 0] + x_off) / np.float32(S), (boxes[..., 1] + y_off) / np.float32(S),
                      np.square(boxes[..., 2]), np.square(boxes[..., 3])], axis=3)
    scores = conf[..., None] * cls[:, :, None, :]
    scores = scores.reshape(-1, C); boxes = boxes.reshape(-1, 4)
    label = np.argmax(scores, axis=1).astype(np.int32); smax = np.max(scores, axis=1)
    m = smax >= np.float32(threshold)
    return boxes[m].astype(np.float32), smax[m], label[m]


def v1_rows(predicts, S=7, B=2, C=20, sqr=True):
    """The same decode as rows [S*S*B, 5+C] = (cx, cy, w, h, conf, cls...) of EVERY box, batch element `predicts` [P]: the layout the
    device keeps (cell-major, box inside the cell)."""
    p = np.asarray(predicts, dtype=np.float32).reshape(-1)
    idx1 = S * S * C; idx2 = idx1 + S * S * B
    cls = p[:idx1].reshape(S * S, C); conf = p[idx1:idx2].reshape(S * S, B); box = p[idx2:].reshape(S * S, B, 4)
    col = (np.arange(S * S) % S).astype(np.float32)[:, None]; row = (np.arange(S * S) // S).astype(np.float32)[:, None]
    out = np.zeros((S * S, B, 5 + C), np.float32)
    out[..., 0] = (box[..., 0] + col) / np.float32(S); out[..., 1] = (box[..., 1] + row) / np.float32(S)
    out[..., 2] = np.square(box[..., 2]) if sqr else box[..., 2]; out[..., 3] = np.square(box[..., 3]) if sqr else box[..., 3]
    out[..., 4] = conf; out[..., 5:] = cls[:, None, :]
    return out.reshape(S * S * B, 5 + C)


def detect_v1_tf(predicts, S=7, B=2, C=20, threshold=0.2, iou_threshold=0.4, max_output_size=10):
    """V1 `_build_detector` complete (V1/YOLO_V1_Inference.py:213-270): decode, `>=` threshold, then
    tf.image.non_max_suppression on `_boxes` = [y - w/2, x - h/2, y + w/2, x + h/2] -- the reference builds the vertical
    extent from the width and the horizontal one from the height (:259-262); kept verbatim.  -> (boxes cxcywh, scores, classes)."""
    boxes, scores, classes = v1_decode(np.asarray(predicts, np.float32).reshape(1, -1), S, B, C, threshold)
    f = np.float32
    _b = np.stack([boxes[:, 1] - f(0.5) * boxes[:, 2], boxes[:, 0] - f(0.5) * boxes[:, 3],
                   boxes[:, 1] + f(0.5) * boxes[:, 2], boxes[:, 0] + f(0.5) * boxes[:, 3]], axis=1).astype(np.float32)
    sel = tf_nms(_b, scores, max_output_size, iou_threshold)
    return boxes[sel], scores[sel], classes[sel]


# ----------------------------------------------------------------------------------------------
# rows N1, N2, N3 (+ darknet do_nms_sort): NMS flavours
# ----------------------------------------------------------------------------------------------
def _tf_iou(b, i, j):
    """TF NonMaxSuppression IOU (non_max_suppression_op.cc): min/max-normalised corners, 0 if an
    area <= 0.  b rows are [y0,x0,y1,x1]."""
    f = np.float32
    ymin_i = min(b[i, 0], b[i, 2]); xmin_i = min(b[i, 1], b[i, 3])
    ymax_i = max(b[i, 0], b[i, 2]); xmax_i = max(b[i, 1], b[i, 3])
    ymin_j = min(b[j, 0], b[j, 2]); xmin_j = min(b[j, 1], b[j, 3])
    ymax_j = max(b[j, 0], b[j, 2]); xmax_j = max(b[j, 1], b[j, 3])
    area_i = f(f(ymax_i - ymin_i) * f(xmax_i - xmin_i))
    area_j = f(f(ymax_j - ymin_j) * f(xmax_j - xmin_j))
    if area_i <= 0 or area_j <= 0:
        return f(0)
    iy0 = max(ymin_i, ymin_j); ix0 = max(xmin_i, xmin_j)
    iy1 = min(ymax_i, ymax_j); ix1 = min(xmax_i, xmax_j)
    inter = f(max(f(iy1 - iy0), f(0)) * max(f(ix1 - ix0), f(0)))
    return f(inter / f(f(area_i + area_j) - inter))


def tf_nms(boxes_yxyx, scores, max_output_size, iou_threshold):
    """`tf.image.non_max_suppression` (TF-1.x): class-agnostic greedy; candidates by score descending
    (ties: lower index first -- the reference kernel's heap order is unspecified, this is our stated
    rule); a candidate is dropped when IoU with any selected box is `> iou_threshold`; stops at
    max_output_size.  Returns selected indices (into the input) in selection order."""
    b = np.asarray(boxes_yxyx, dtype=np.float32).reshape(-1, 4)
    s = np.asarray(scores, dtype=np.float32)
    order = np.argsort(-s, kind="stable")
    sel = []
    for i in order:
        if len(sel) >= max_output_size:
            break
        ok = True
        for j in reversed(sel):
            if _tf_iou(b, i, j) > np.float32(iou_threshold):
                ok = False
                break
        if ok:
            sel.append(int(i))
    return np.array(sel, dtype=np.int32)


def np_iou_v3(box1, box2):
    """`_iou` V3/yolo_v3.py:350-373 incl. its quirks (no clamp of negative overlap, +1e-05)."""
    b1_x0, b1_y0, b1_x1, b1_y1 = box1
    b2_x0, b2_y0, b2_x1, b2_y1 = box2
    int_x0 = max(b1_x0, b2_x0); int_y0 = max(b1_y0, b2_y0)
    int_x1 = min(b1_x1, b2_x1); int_y1 = min(b1_y1, b2_y1)
    int_area = (int_x1 - int_x0) * (int_y1 - int_y0)
    b1_area = (b1_x1 - b1_x0) * (b1_y1 - b1_y0)
    b2_area = (b2_x1 - b2_x0) * (b2_y1 - b2_y0)
    return int_area / (b1_area + b2_area - int_area + 1e-05)


def np_nms_v3(predictions_with_boxes, confidence_threshold, iou_threshold=0.4):
    """`non_max_suppression` V3/yolo_v3.py:376-420, restated with its behaviours kept: objectness-only
    gate (:385), class = argmax cls (:397), per class sort by objectness desc (:404, numpy default
    argsort reversed), greedy keep `iou < thr` (:416), result dict shared across the batch (:388),
    scores indexed through the mask built on cls_boxes[1:] (:414-418, the off-by-one)."""
    p = np.asarray(predictions_with_boxes)
    conf_mask = np.expand_dims(p[:, :, 4] > confidence_threshold, -1)
    predictions = p * conf_mask
    result = {}
    for image_pred in predictions:
        shape = image_pred.shape
        nz = np.nonzero(image_pred)
        image_pred = image_pred[nz].reshape(-1, shape[-1])
        bbox_attrs = image_pred[:, :5]
        classes = np.argmax(image_pred[:, 5:], axis=-1)
        for cls in list(set(classes.reshape(-1))):
            cls_boxes = bbox_attrs[np.nonzero(classes == cls)]
            cls_boxes = cls_boxes[cls_boxes[:, -1].argsort()[::-1]]
            cls_scores = cls_boxes[:, -1]
            cls_boxes = cls_boxes[:, :-1]
            while len(cls_boxes) > 0:
                box = cls_boxes[0]; score = cls_scores[0]
                result.setdefault(cls, []).append((box, score))
                cls_boxes = cls_boxes[1:]
                ious = np.array([np_iou_v3(box, x) for x in cls_boxes])
                keep = np.nonzero(ious < iou_threshold)
                cls_boxes = cls_boxes[keep]
                cls_scores = cls_scores[keep]
    return result


def v2_bboxes_iou(b1, b2):
    """V2/utils.py:155-174 (clamped overlap; division by zero possible for degenerate int boxes)."""
    b1 = np.transpose(b1); b2 = np.transpose(b2)
    int_ymin = np.maximum(b1[0], b2[0]); int_xmin = np.maximum(b1[1], b2[1])
    int_ymax = np.minimum(b1[2], b2[2]); int_xmax = np.minimum(b1[3], b2[3])
    int_h = np.maximum(int_ymax - int_ymin, 0.); int_w = np.maximum(int_xmax - int_xmin, 0.)
    int_vol = int_h * int_w
    vol1 = (b1[2] - b1[0]) * (b1[3] - b1[1]); vol2 = (b2[2] - b2[0]) * (b2[3] - b2[1])
    with np.errstate(divide="ignore", invalid="ignore"):
        return int_vol / (vol1 + vol2 - int_vol)


def v2_postprocess(bboxes, obj_probs, class_probs, image_shape=(416, 416), threshold=0.5,
                   top_k=400, nms_threshold=0.5):
    """`postprocess` V2/utils.py:30-62 -> bboxes_cut :133-143, bboxes_sort :146-151, bboxes_nms :176-187."""
    bboxes = np.reshape(np.array(bboxes, dtype=np.float32), [-1, 4])
    bboxes[:, 0:1] *= float(image_shape[1]); bboxes[:, 1:2] *= float(image_shape[0])
    bboxes[:, 2:3] *= float(image_shape[1]); bboxes[:, 3:4] *= float(image_shape[0])
    bboxes = bboxes.astype(np.int32)
    mm = [0, 0, image_shape[1] - 1, image_shape[0] - 1]
    bboxes = np.stack([np.maximum(bboxes[:, 0], mm[0]), np.maximum(bboxes[:, 1], mm[1]),
                       np.minimum(bboxes[:, 2], mm[2]), np.minimum(bboxes[:, 3], mm[3])], axis=1).astype(np.int32)
    obj = np.reshape(obj_probs, [-1])
    cls = np.reshape(class_probs, [len(obj), -1])
    cmax = np.argmax(cls, axis=1)
    scores = obj * cls[np.arange(len(obj)), cmax]
    keep = scores > threshold
    cmax, scores, bboxes = cmax[keep], scores[keep], bboxes[keep]
    index = np.argsort(-scores)
    cmax, scores, bboxes = cmax[index][:top_k], scores[index][:top_k], bboxes[index][:top_k]
    kb = np.ones(scores.shape, dtype=bool)
    for i in range(scores.size - 1):
        if kb[i]:
            overlap = v2_bboxes_iou(bboxes[i], bboxes[(i + 1):])
            keep_overlap = np.logical_or(overlap < nms_threshold, cmax[(i + 1):] != cmax[i])
            kb[(i + 1):] = np.logical_and(kb[(i + 1):], keep_overlap)
    idx = np.where(kb)
    return bboxes[idx], scores[idx], cmax[idx]


def _dn_overlap(x1, w1, x2, w2):
    f = np.float32
    l1 = f(x1 - f(w1 / f(2))); l2 = f(x2 - f(w2 / f(2)))
    r1 = f(x1 + f(w1 / f(2))); r2 = f(x2 + f(w2 / f(2)))
    return f(min(r1, r2) - max(l1, l2))


def dn_box_iou(a, b):
    """darknet box_iou (DN/box.c:152-182) on (cx,cy,w,h) boxes, fp32 step by step."""
    f = np.float32
    w = _dn_overlap(a[0], a[2], b[0], b[2]); h = _dn_overlap(a[1], a[3], b[1], b[3])
    inter = f(0) if (w < 0 or h < 0) else f(w * h)
    union = f(f(f(a[2] * a[3]) + f(b[2] * b[3])) - inter)
    return f(inter / union)


def dn_nms_sort(boxes_cxcywh, probs, thresh):
    """darknet do_nms_sort (DN/box.c:58-89): per class k, sort by prob[k] desc, zero prob[k] of any later
    box with iou > thresh.  probs [n,classes] is modified copy-returned.  (The objectness==0 pre-filter
    at :63-71 only reorders; survivors are identical.)"""
    b = np.asarray(boxes_cxcywh, dtype=np.float32)
    p = np.array(probs, dtype=np.float32)
    n, C = p.shape
    for k in range(C):
        order = np.argsort(-p[:, k], kind="stable")
        for ii in range(n):
            i = order[ii]
            if p[i, k] == 0:
                continue
            for jj in range(ii + 1, n):
                j = order[jj]
                if p[j, k] != 0 and dn_box_iou(b[i], b[j]) > np.float32(thresh):
                    p[j, k] = 0
    return p


# ----------------------------------------------------------------------------------------------
# whole network (rows B, Y, Net): cfg-driven forward
# ----------------------------------------------------------------------------------------------
def flatten_weights(params, secs):
    """Inverse of unflatten_weights: the per-conv dicts back into darknet's flat stream (file order, filters OIHW)."""
    parts = []
    for c, p in zip(conv_layers(secs), params):
        if c.get("local"):
            parts += [p["bias_fl"], p["w_local"]]
            continue
        if c["bn"]:
            parts += [p["beta"], p["gamma"], p["mean"], p["var"]]
        else:
            parts.append(p["bias"])
        parts.append(np.transpose(p["w_hwio"], (3, 2, 0, 1)).reshape(-1))            # HWIO -> OIHW
    return np.concatenate([np.asarray(q, dtype=np.float32).reshape(-1) for q in parts])


def calibrate_bn_statistics(secs, params, images01, seed=0, head_std=(1.0, 0.35, 1.5), keep_var=False):
    """Make a synthetic parameter set behave like a TRAINED file on the given images: one forward pass in which every batch-normalised
    conv's filters are rescaled per output channel to a target variance drawn from the ranges of the reference's dump of real files
    (D2T/log.txt: 2e-3 .. 0.3 on the first two layers, 0.6 .. 19 after, a few per cent down to 8e-4) and its rolling mean / variance
    are set to what those filters produce on `images01` ([N,S,S,3] in 0..1) -- training's running averages.  gamma / beta are left as
    drawn (darknet_io.synth_weights(stats="log"): gamma to 4.7, some negative, beta to -11).  Head convs are rescaled so that their raw
    outputs have standard deviation `head_std` = (centre logits, log-size offsets, objectness / class logits) around their biases -- a
    trained detector's size offsets stay within about +-1 (boxes 0.3 .. 3 anchors), not the +-5 of an unscaled random filter.  Without this the analytic statistics of the generator hold for
    white-noise inputs only: on natural images neighbouring taps add coherently and the activations grow layer by layer.
    keep_var=True (darknet_io.synth_weights(stats="real"): beta, gamma and rolling variance are the reference's real vectors,
    D2T/log.txt:224-949): the target variance of a channel is the rolling variance it already carries, which stays as it is; only the
    filters are rescaled and the rolling mean set.
    Test infrastructure (no reference counterpart: the reference ships no weights).  Returns the calibrated params (modified in place)."""
    rng = np.random.default_rng(seed)

    def hook(i, p, y):
        n = y.shape[-1]
        m = y.mean((0, 1, 2), dtype=np.float64); v = y.var((0, 1, 2), dtype=np.float64)
        if "bias" in p:
            attrs = None
            for h in secs[1:][i + 1:i + 2]:
                if h["type"] in ("yolo", "region"):
                    attrs = 5 + int(h.get("classes", 20))
            k = np.arange(n) % attrs if attrs and n % attrs == 0 else np.full(n, 4)
            want = np.where(k < 2, head_std[0], np.where(k < 4, head_std[1], head_std[2]))
            sc = want / np.sqrt(np.maximum(v, 1e-20))
        elif keep_var:
            target = np.maximum(p["var"].astype(np.float64), 1e-30)
            sc = np.sqrt(target / np.maximum(v, 1e-30))
            p["mean"] = (m * sc).astype(np.float32)
        else:
            lo, hi = (2e-3, 0.3) if i < 2 else (0.6, 19.0)
            target = np.exp(rng.uniform(np.log(lo), np.log(hi), n))
            tiny = rng.random(n) < 0.03
            target[tiny] = np.exp(rng.uniform(np.log(8e-4), np.log(1e-2), int(tiny.sum())))
            sc = np.sqrt(target / np.maximum(v, 1e-20))
            p["mean"] = (m * sc).astype(np.float32); p["var"] = (target * rng.uniform(0.9, 1.1, n)).astype(np.float32)
        p["w_hwio"] = (p["w_hwio"] * sc[None, None, None, :]).astype(np.float32)
        return (y * sc.astype(np.float32)).astype(np.float32)

    forward(secs, params, images01, calibrate=hook)
    return params


def forward(secs, params, x, semantics="tf", bn_mode="tf", emulate_bf16=False, collect=None, calibrate=None, storage=None):
    """Run the cfg (yolov3 == V3/yolo_v3.py:195-267; yolov2 == V2/model_darknet19_slim.py:119-200; ...)
    on x [N,S,S,3] float32 **already scaled to 0..1** (the /255 of V3/yolo_v3.py:215 is applied by the
    caller, as D2T `_input_process` does).

    semantics 'tf'      : upsample = `_upsample` bilinear, reorg = tf.space_to_depth   (parity target)
    semantics 'darknet' : upsample = nearest, reorg = darknet reorg_cpu                (pins vs oracle/_ref)
    bn_mode             : see batch_norm()
    emulate_bf16        : fold BN, round folded weights and every stored activation to bf16 -- the
                          device's storage precision; head conv outputs stay fp32 (device keeps them fp32).
    Returns (heads, outs): heads = list of (section, raw head tensor [N,g,g,ch] fp32) in cfg order;
    outs = list of per-layer NHWC outputs (None for yolo/region) when collect is not None."""
    # storage: "bf16" (== emulate_bf16=True) or "f16": the device's 16-bit storage types (folded filters and stored activations rounded)
    if storage is not None:
        emulate_bf16 = True
    q = (to_f16 if storage == "f16" else to_bf16) if emulate_bf16 else (lambda a: a)
    x = q(np.asarray(x, dtype=np.float32))
    layers = secs[1:]
    outs = []
    heads = []
    ci = 0
    for i, s in enumerate(layers):
        t = s["type"]
        if t == "connected":
            # Net1's head (V1/YOLO_V1_Inference.py:196-206): NHWC -> NCHW transpose, flatten, x @ W + b; darknet's [connected]
            # (DN/connected_layer.c:151-166) is the same product on its CHW tensors.  Stored here as [N,1,1,output].
            p = params[ci]; ci += 1
            is_head = i + 1 < len(layers) and layers[i + 1]["type"] == "detection"
            w = p["w_hwio"].reshape(-1, p["w_hwio"].shape[-1])            # [inputs (CHW order), output]
            flat = np.ascontiguousarray(np.transpose(x, (0, 3, 1, 2))).reshape(x.shape[0], -1)
            y = (flat @ (q(w) if emulate_bf16 else w) + p["bias"]).astype(np.float32)
            act = s.get("activation", "logistic")
            if act == "leaky":
                y = leaky_relu(y)
            elif act != "linear":
                raise ValueError(act)
            y = y.reshape(x.shape[0], 1, 1, -1)
            x = y if is_head else q(y)
        elif t == "local":
            # DN/local_layer.c:91-120: out[f][loc] = bias[f][loc] + sum_k W[loc][f][k] * col[k][loc], k = (c, kh, kw); then the activation
            p = params[ci]; ci += 1
            k, st, pad = int(s["size"]), int(s.get("stride", 1)), int(s.get("pad", 0))
            wl = to_bf16(p["w_local"]) if emulate_bf16 else p["w_local"]
            if emulate_bf16:
                wl = q(p["w_local"])
            N_, H_, W_, C_ = x.shape
            Ho = ((H_ - 1) if pad else (H_ - k)) // st + 1; Wo = ((W_ - 1) if pad else (W_ - k)) // st + 1
            xp = np.pad(x, ((0, 0), (pad, pad + k), (pad, pad + k), (0, 0)))
            y = np.zeros((N_, Ho, Wo, wl.shape[1]), np.float32)
            for oy in range(Ho):
                for ox in range(Wo):
                    patch = xp[:, oy * st:oy * st + k, ox * st:ox * st + k, :]                              # [N, kh, kw, c]
                    wloc = np.transpose(wl[oy * Wo + ox], (0, 2, 3, 1))                                       # [f, kh, kw, c]
                    y[:, oy, ox, :] = np.einsum("nhwc,fhwc->nf", patch, wloc, dtype=np.float32) + p["bias_fl"][:, oy * Wo + ox]
            act = s.get("activation", "logistic")
            if act == "leaky":
                y = leaky_relu(y)
            elif act != "linear":
                raise ValueError(act)
            x = q(y.astype(np.float32))
        elif t == "dropout":
            pass                                                          # inference: identity (is_training=False, :203-204)
        elif t == "detection":
            heads.append((s, outs[i - 1]))
            outs.append(None)
            continue
        elif t == "convolutional":
            p = params[ci]; ci += 1
            k, st = int(s["size"]), int(s.get("stride", 1))
            is_head = i + 1 < len(layers) and layers[i + 1]["type"] in ("yolo", "region")
            if emulate_bf16:
                w, b = fold_bn(p, mode="darknet" if semantics == "darknet" else "tf")
                y = conv2d_nhwc(x, q(w), st) + b
            elif calibrate is not None:          # (calibrate_bn_statistics: the hook sees the raw conv output and rewrites p)
                y = calibrate(i, p, conv2d_nhwc(x, p["w_hwio"], st))
                y = y + p["bias"] if "bias" in p else batch_norm(y, p, bn_mode)
            elif "bias" in p:
                y = conv2d_nhwc(x, p["w_hwio"], st) + p["bias"]
            else:
                y = batch_norm(conv2d_nhwc(x, p["w_hwio"], st), p, bn_mode)
            act = s.get("activation", "logistic")
            if act == "leaky":
                y = leaky_relu(y)
            elif act != "linear":
                raise ValueError(act)
            x = y.astype(np.float32) if is_head else q(y.astype(np.float32))
        elif t == "shortcut":
            f = int(s["from"]); f = f if f >= 0 else i + f
            x = q(outs[i - 1] + outs[f])
        elif t == "route":
            ls = [int(v) for v in s["layers"].split(",")]
            ls = [l if l >= 0 else i + l for l in ls]
            x = np.concatenate([outs[l] for l in ls], axis=-1) if len(ls) > 1 else outs[ls[0]]
        elif t == "upsample":
            if semantics == "tf":
                x = q(upsample_tf(x))
            else:
                x = upsample_nearest(x, int(s.get("stride", 2)))
        elif t == "maxpool":
            st = int(s.get("stride", 1)); k = int(s.get("size", st))
            x = max_pool(x, k, st, int(s.get("padding", (k - 1) // 2)))
        elif t == "reorg":
            st = int(s.get("stride", 1))
            x = space_to_depth(x, st) if semantics == "tf" else reorg_darknet(x, st)
        elif t in ("yolo", "region"):
            heads.append((s, outs[i - 1]))
            outs.append(None)
            continue
        else:
            raise ValueError(t)
        outs.append(x)
    return heads, outs



def split_f16(x):
    """Split fp16 storage of the device's YOLO_FP16X2 configuration (no reference counterpart -- the reference has no
    reduced-precision path): v -> (hi, lo) with hi = f16(v) (round to nearest even, saturating at +-65504 as the device's conversions do),
    lo = f16(v - hi); both returned as float32 arrays holding fp16 values."""
    x = np.asarray(x, dtype=np.float32)
    hi = to_f16(x)
    return hi, to_f16(x - hi)


def forward_f16x2(secs, params, x, semantics="tf", collect=None, pair=None):
    """Emulation of the device's split-fp16 configuration: every stored activation and folded filter is a pair (hi, lo) of fp16 numbers;
    a conv is W_hi x_hi + W_hi x_lo + W_lo x_hi (+ fp32 bias) in fp32 -- W_lo x_lo is dropped, as on the device --, the shortcut and the
    layers that move or interpolate values work on hi + lo and split again; head convs stay fp32.  `x` is [N,S,S,3] float32 in 0..1.
    `pair` (round 5, mixed plans: darknet_io.pair_closure): {layer index: bool, -1 = the image}; a tensor whose entry is False is PLAIN fp16
    (hi only, lo = 0), a conv that reads a plain tensor multiplies it by fp16-rounded filters (one product), a shortcut of plain operands
    adds and rounds once; None = pairs everywhere.
    Returns (heads, outs) like forward(); outs[i] is the JOINED value hi + lo of layer i's tensor.  Test infrastructure."""
    layers = secs[1:]
    outs, heads = [], []
    ci = 0
    is_pair = (lambda i: True) if pair is None else (lambda i: bool(pair.get(i, True)))
    def store(v, i):
        v = np.asarray(v, dtype=np.float32)
        if is_pair(i):
            return split_f16(v)
        h = to_f16(v)
        return h, np.zeros_like(h)
    xs = store(x, -1)
    for i, s in enumerate(layers):
        t = s["type"]
        if t == "convolutional":
            p = params[ci]; ci += 1
            st = int(s.get("stride", 1))
            is_head = i + 1 < len(layers) and layers[i + 1]["type"] in ("yolo", "region")
            w, b = fold_bn(p, mode="darknet" if semantics == "darknet" else "tf")
            wh, wl = split_f16(w); xh, xl = xs
            if is_pair(i - 1):
                y = (conv2d_nhwc(xh, wh, st) + conv2d_nhwc(xl, wh, st) + conv2d_nhwc(xh, wl, st) + b).astype(np.float32)
            else:
                y = (conv2d_nhwc(xh, wh, st) + b).astype(np.float32)
            act = s.get("activation", "logistic")
            if act == "leaky":
                y = leaky_relu(y)
            elif act != "linear":
                raise ValueError(act)
            y = y.astype(np.float32)
            xs = (y, np.zeros_like(y)) if is_head else store(y, i)
        elif t == "shortcut":
            f = int(s["from"]); f = f if f >= 0 else i + f
            a, b2 = outs[i - 1], outs[f]
            xs = store((a[0] + a[1]) + (b2[0] + b2[1]), i)
        elif t == "route":
            ls = [int(v) for v in s["layers"].split(",")]
            ls = [l if l >= 0 else i + l for l in ls]
            xs = tuple(np.concatenate([outs[l][k] for l in ls], axis=-1) for k in (0, 1)) if len(ls) > 1 else outs[ls[0]]
        elif t == "upsample":
            v = xs[0] + xs[1]
            xs = store(upsample_tf(v) if semantics == "tf" else upsample_nearest(v, int(s.get("stride", 2))), i)
        elif t == "maxpool":
            st = int(s.get("stride", 1)); k = int(s.get("size", st))
            xs = store(max_pool(xs[0] + xs[1], k, st, int(s.get("padding", (k - 1) // 2))), i)
        elif t == "reorg":
            st = int(s.get("stride", 1)); v = xs[0] + xs[1]
            xs = store(space_to_depth(v, st) if semantics == "tf" else reorg_darknet(v, st), i)
        elif t in ("yolo", "region"):
            heads.append((s, outs[i - 1][0]))
            outs.append(None)
            continue
        else:
            raise ValueError("%s: not served in the split-fp16 configuration" % t)
        outs.append(xs)
    joined = [None if o is None else (o[0] + o[1]) for o in outs] if collect is not None else None
    return heads, joined

def yolo_anchors(sec):
    a = [float(v) for v in sec["anchors"].split(",")]
    pairs = [(a[2 * i], a[2 * i + 1]) for i in range(len(a) // 2)]
    if "mask" in sec:
        pairs = [pairs[int(m)] for m in sec["mask"].split(",")]
    return pairs


def yolo_v3_detections(heads, img_size, ratio=False):
    """concat of the per-scale detection layers (V3/yolo_v3.py:266): [N, sum(g*g*3), 5+C]."""
    fn = detection_layer_ratio if ratio else detection_layer_pixel
    return np.concatenate([fn(raw, yolo_anchors(s), (img_size, img_size)) for s, raw in heads], axis=1)


def detect_v3_tf(det_ratio_one_image, score_threshold, iou_threshold, max_output_size):
    """D2T YOLOV3 tail (YOLO_V3_convert...py:515-545) on one image's [rows,5+C] ratio detections:
    -> (boxes [K,4] x0y0x1y1, scores [K], classes [K])."""
    boxes, scores, classes, _ = select_threshold(det_ratio_one_image, score_threshold)
    sel = tf_nms(boxes[:, [1, 0, 3, 2]], scores, max_output_size, iou_threshold)
    return boxes[sel], scores[sel], classes[sel]
