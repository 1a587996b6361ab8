"""Darknet topology (`.cfg`) and `.weights` stream handling for the host side.

Counterpart of the reference's `load_weights` (V3/yolo_v3.py:270-326; D2T/YOLO_V3_convert...py:113-216)
and of darknet's writer/reader (DN/parser.c:992-1067 save, :1241-1345 load): the file is
`int32 major, minor, revision`, then `seen` (int64 when major*10+minor >= 2, else int32;
DN/parser.c:1259-1265), then for every convolutional layer in cfg order
`[beta|bias][gamma][mean][var]` (BN) or `[bias]`, then the filters `OIHW`.

The HIP library consumes the flat float stream directly (`yolo_set_weights`); nothing here touches
the device.  Also holds the seeded synthetic-weights generator used by bench.py and the tests
(SURVEY.md 8d "Synthetic inputs") because no real `.weights` ships with the reference.
"""
import os
import numpy as np

CFG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cfg")


def cfg_text(name):
    """Text of a shipped topology ('yolov3', 'yolov3-608', 'yolov3-tiny', 'yolov2', 'yolov2-tiny-voc', 'yolov1', 'yolov1-tiny')
    or of a cfg file path."""
    path = name if os.path.exists(name) else os.path.join(CFG_DIR, name + ".cfg")
    with open(path) as f:
        return f.read()


def with_input_size(text, size):
    """Same topology at another square input size (kernels are shape-generic in H,W; darknet
    `resize_network`, DN/network.c:358-438)."""
    out = []
    for line in text.splitlines():
        key = line.split("=")[0].strip()
        out.append(f"{key}={size}" if key in ("width", "height") else line)
    return "\n".join(out)


def with_layer_store(text, layers, store="bf16"):
    """cfg text with `yolo_store=<store>` added to the [convolutional] sections whose layer index (0-based, [net] not counted) is in
    `layers`: in an fp8 network those layers' outputs -- and what is derived from them without arithmetic (shortcut, route, upsample,
    pooling) -- are stored in bf16, and the convs that read them run on the bf16 MFMA (mixed-precision plans).  Operands of a shortcut
    or of a concatenating route must end up in one type; `store_closure` adds what that takes."""
    want = set(int(l) for l in layers)
    out, idx = [], -2
    for line in text.splitlines():
        t = line.strip()
        if t.startswith("["):
            idx += 1
            out.append(line)
            if idx in want:
                if t[1:t.index("]")].strip() != "convolutional":
                    raise ValueError("layer %d is not a [convolutional] section" % idx)
                out.append("yolo_store=%s" % store)
            continue
        if t.split("=")[0].strip() == "yolo_store":
            continue
        out.append(line)
    return "\n".join(out) + "\n"


def store_closure(secs, layers):
    """Smallest superset of the conv layers `layers` whose bf16 storage is consistent: both operands of every shortcut and all inputs
    of every concatenating route are stored in one type (the residual stream of a stage is all-or-nothing)."""
    L = secs[1:]
    S = set(int(l) for l in layers)

    def inputs(i):
        s = L[i]; t = s["type"]
        if t == "shortcut":
            f = int(s["from"]); return [i - 1, f if f >= 0 else i + f]
        if t == "route":
            return [int(x) if int(x) >= 0 else i + int(x) for x in s["layers"].split(",")]
        return [i - 1]

    def is16(i, S):
        t = L[i]["type"]
        if t == "convolutional":
            return i in S
        if t in ("yolo", "region", "detection"):
            return False
        return any(is16(j, S) for j in inputs(i))

    def force(i, S):                      # make layer i's tensor bf16
        t = L[i]["type"]
        if t == "convolutional":
            S.add(i)
        else:
            for j in inputs(i):
                force(j, S)

    changed = True
    while changed:
        changed = False
        for i, s in enumerate(L):
            if s["type"] == "shortcut" or (s["type"] == "route" and "," in s["layers"]):
                ins = inputs(i)
                flags = [is16(j, S) for j in ins]
                if any(flags) and not all(flags):
                    before = len(S)
                    for j in ins:
                        force(j, S)
                    changed = changed or len(S) != before
    return sorted(S)


def bf16_flop_share(secs):
    """Share of the conv FLOPs (2 k k Cin Cout Ho Wo) of a mixed plan that runs on the bf16 MFMA: the convs whose INPUT tensor is stored
    in bf16 (cfg key yolo_store on the producers, inherited through shortcut / route / upsample / pooling) and the first conv."""
    L = secs[1:]
    shapes = layer_shapes(secs)

    def is16(i):
        t = L[i]["type"]
        if t == "convolutional":
            return L[i].get("yolo_store") == "bf16"
        if t in ("yolo", "region", "detection"):
            return False
        if t == "route":
            ls = [int(v) if int(v) >= 0 else i + int(v) for v in L[i]["layers"].split(",")]
            return is16(ls[0])
        return is16(i - 1)

    tot = b16 = 0.0
    for i, s in enumerate(L):
        if s["type"] != "convolutional":
            continue
        f = 2.0 * int(s["size"]) ** 2 * shapes[i][4] * shapes[i][3] * shapes[i][1] * shapes[i][2]
        tot += f
        if i == 0 or is16(i - 1):
            b16 += f
    return b16 / tot


def parse_cfg(text):
    secs = []
    for line in text.splitlines():
        line = line.strip()
        if not line or line[0] in "#;":
            continue
        if line[0] == "[":
            secs.append({"type": line[1:line.index("]")].strip()})
        else:
            k, v = line.split("=", 1)
            secs[-1][k.strip()] = v.strip()
    return secs


def layer_shapes(secs):
    """[(type, H, W, C_out, C_in)] per layer, darknet shape rules (DN/parser.c:730-875)."""
    net = secs[0]
    H, W, C = int(net["height"]), int(net["width"]), int(net["channels"])
    shapes = []
    for i, s in enumerate(secs[1:]):
        t, cin = s["type"], C
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
            H, W = shapes[ls[0]][1], shapes[ls[0]][2]
            C = sum(shapes[l][3] for l in ls)
        elif t == "connected":               # flattens its producer (CHW order, DN/connected_layer.c:151) to `output` values
            cin = H * W * C
            H, W, C = 1, 1, int(s["output"])
        elif t == "local":                   # locally connected: unshared filters per output location (DN/local_layer.c:10-24)
            k, st, pad = int(s["size"]), int(s.get("stride", 1)), int(s.get("pad", 0))
            H, W, C = ((H - 1) if pad else (H - k)) // st + 1, ((W - 1) if pad else (W - k)) // st + 1, int(s["filters"])
        elif t == "detection":
            H = W = int(s.get("side", 7))
        elif t not in ("shortcut", "yolo", "region", "dropout"):
            raise ValueError("unsupported layer type [%s]" % t)
        shapes.append((t, H, W, C, cin))
    return shapes


def conv_specs(secs):
    """Per parameterised layer in file order ([convolutional], and [connected] as a 1x1 conv over the flattened producer):
    dict(filters, size, cin, bn, head, index)."""
    shapes = layer_shapes(secs)
    layers = secs[1:]
    out = []
    for i, s in enumerate(layers):
        head = i + 1 < len(layers) and layers[i + 1]["type"] in ("yolo", "region", "detection")
        if s["type"] == "convolutional":
            out.append(dict(filters=int(s["filters"]), size=int(s["size"]), cin=shapes[i][4],
                            bn=int(s.get("batch_normalize", 0)), head=head, index=i))
        elif s["type"] == "connected":
            out.append(dict(filters=int(s["output"]), size=1, cin=shapes[i][4], bn=0, head=head, index=i))
        elif s["type"] == "local":           # biases [filters * locations], weights [locations][filters][cin][k][k] (DN/parser.c:1315-1320)
            out.append(dict(filters=int(s["filters"]), size=int(s["size"]), cin=shapes[i][4], bn=0, head=False, index=i, locations=shapes[i][1] * shapes[i][2]))
    return out


def weights_count(secs):
    n = 0
    for c in conv_specs(secs):
        loc = c.get("locations", 1)
        n += c["filters"] * loc * (4 if c["bn"] else 1) + loc * c["filters"] * c["cin"] * c["size"] ** 2
    return n


def read_weights_file(path, header_ints=None):
    """-> (flat float32 stream, (major, minor, revision, seen)).  `header_ints` forces the
    reference's fixed counts (5: V3/yolo_v3.py:278; 4: D2T/YOLO_V2_convert...py:351)."""
    with open(path, "rb") as f:
        ver = np.fromfile(f, dtype=np.int32, count=3)
        if ver.size < 3:
            raise ValueError("truncated .weights header: " + path)
        if header_ints is None:
            header_ints = 5 if int(ver[0]) * 10 + int(ver[1]) >= 2 else 4
        seen = np.fromfile(f, dtype=np.int64 if header_ints == 5 else np.int32, count=1)
        flat = np.fromfile(f, dtype=np.float32)
    return flat, (int(ver[0]), int(ver[1]), int(ver[2]), int(seen[0]) if seen.size else 0)


def write_weights_file(path, flat, major=0, minor=2, revision=0, seen=0):
    """Writer matching darknet `save_weights_upto` (DN/parser.c:992-1009)."""
    with open(path, "wb") as f:
        np.array([major, minor, revision], dtype=np.int32).tofile(f)
        (np.array([seen], dtype=np.int64) if major * 10 + minor >= 2
         else np.array([seen], dtype=np.int32)).tofile(f)
        np.asarray(flat, dtype=np.float32).tofile(f)


def synth_weights(secs, seed=0, obj_bias=-0.75, stats="benign"):
    """Seeded synthetic parameter stream, valid for any supported topology.
    stats="log": batch-norm statistics in the ranges of the reference's dump of real files (see synth_weights_log).
    stats="real": the reference's REAL batch-norm vectors (see synth_weights_real; yolov3 / yolov2 topologies only).

    Filters W ~ N(0, 2/(k*k*Cin)) (darknet's own init, DN/convolutional_layer.c:205-209); gamma ~ U(.8,1.2),
    beta ~ N(0,.1), rolling_mean ~ N(0,.1).  rolling_variance is set to the *expected* variance of the
    conv output (tracked analytically through conv / shortcut / route), times U(.8,1.25) -- what a trained
    network's statistics look like -- so activations stay O(1) through 75 layers instead of growing
    ~1.1x per layer (the reference's real files span var 2e-3..16, D2T/log.txt).  Head convs: small
    filters, class/box biases N(0,1)/N(0,.5), objectness bias `obj_bias` (a few % of candidates pass 0.5)."""
    if stats == "log":
        return synth_weights_log(secs, seed, obj_bias)
    if stats == "real":
        return synth_weights_real(secs, seed, obj_bias)
    rng = np.random.default_rng(seed)
    layers = secs[1:]
    shapes = layer_shapes(secs)
    m2 = []                  # expected second moment of each layer's output
    cur = 1.0 / 3.0          # uniform [0,1) pixels: E[x^2]
    parts = []
    for i, s in enumerate(layers):
        t = s["type"]
        if t == "convolutional":
            n, k, cin = int(s["filters"]), int(s["size"]), shapes[i][4]
            head = i + 1 < len(layers) and layers[i + 1]["type"] in ("yolo", "region")
            if int(s.get("batch_normalize", 0)):
                pre_var = 2.0 * cur                                   # k*k*cin * (2/(k*k*cin)) * E[x^2]
                gamma = rng.uniform(.8, 1.2, n)
                parts += [rng.normal(0, .1, n), gamma, rng.normal(0, .1 * np.sqrt(pre_var), n), pre_var * rng.uniform(.8, 1.25, n)]
                parts.append(rng.normal(0, np.sqrt(2.0 / (k * k * cin)), n * cin * k * k))
                post = 1.0 + 0.01                                     # gamma^2 + beta^2 on average
                cur = post * (0.505 if s.get("activation", "linear") == "leaky" else 1.0)
            else:
                b = rng.normal(0, 1.0, n)
                if head:
                    h = layers[i + 1]
                    classes = int(h.get("classes", 20))
                    na = len(h["mask"].split(",")) if "mask" in h else int(h.get("num", 1))
                    attrs = 5 + classes
                    if na * attrs == n:
                        b = b.reshape(na, attrs)
                        b[:, 0:4] = rng.normal(0, .5, (na, 4))
                        b[:, 4] = obj_bias + rng.normal(0, .5, na)
                        b = b.reshape(-1)
                parts.append(b)
                parts.append(rng.normal(0, np.sqrt(1.0 / (k * k * cin * max(cur, 1e-6))), n * cin * k * k))
                cur = 1.0 + 1.0
        elif t == "local":
            n, k, cin = int(s["filters"]), int(s["size"]), shapes[i][4]
            loc = shapes[i][1] * shapes[i][2]
            parts += [rng.normal(0, .1, n * loc), rng.normal(0, np.sqrt(2.0 / (k * k * cin * max(cur, 1e-6))), loc * n * cin * k * k)]
            cur = 1.01 * (0.505 if s.get("activation", "linear") == "leaky" else 1.0)
        elif t == "connected":
            n, cin = int(s["output"]), shapes[i][4]
            if i + 1 < len(layers) and layers[i + 1]["type"] == "detection":
                # a [detection] head reads its inputs as probabilities / box numbers directly (no sigmoid): small filters around biases
                # that look like them -- classes U(0,.9), confidences U(.05,.95), centres U(.1,.9), sqrt-sizes U(.3,.8)
                h = layers[i + 1]; S, B, C = int(h.get("side", 7)), int(h.get("num", 1)), int(h.get("classes", 1))
                b = np.concatenate([rng.uniform(0, .9, S * S * C), rng.uniform(.05, .95, S * S * B),
                                    np.stack([rng.uniform(.1, .9, S * S * B), rng.uniform(.1, .9, S * S * B),
                                              rng.uniform(.3, .8, S * S * B), rng.uniform(.3, .8, S * S * B)], -1).reshape(-1)])
                parts += [b, rng.normal(0, 0.05 * np.sqrt(1.0 / (cin * max(cur, 1e-6))), n * cin)]
            else:
                parts += [rng.normal(0, .1, n), rng.normal(0, np.sqrt(2.0 / (cin * max(cur, 1e-6))), n * cin)]
            cur = 1.0 + 0.01 if s.get("activation", "logistic") != "leaky" else 1.01
        elif t == "shortcut":
            f = int(s["from"]); f = f if f >= 0 else i + f
            # leaky outputs have a positive mean, so residual branches add coherently: use the fully
            # correlated bound (slightly over-normalises, never explodes)
            cur = (np.sqrt(m2[i - 1]) + np.sqrt(m2[f])) ** 2
        elif t == "route":
            ls = [int(x) for x in s["layers"].split(",")]
            ls = [l if l >= 0 else i + l for l in ls]
            tot = sum(shapes[l][3] for l in ls)
            cur = sum(m2[l] * shapes[l][3] for l in ls) / tot
        m2.append(cur)
    return np.concatenate(parts).astype(np.float32)


def _leaky_moments(mean, std, slope):
    """(E[f(y)], E[f(y)^2]) for y ~ N(mean, std^2) and f = leaky ReLU with `slope` (slope 1: identity); arrays per channel."""
    from math import pi
    std = np.maximum(std, 1e-12)
    z = mean / std
    cdf = 0.5 * (1.0 + _erf(z / np.sqrt(2.0))); pdf = np.exp(-0.5 * z * z) / np.sqrt(2.0 * pi)
    m_pos = mean * cdf + std * pdf                                   # E[y 1(y > 0)]
    s_pos = (mean * mean + std * std) * cdf + mean * std * pdf       # E[y^2 1(y > 0)]
    m1 = m_pos + slope * (mean - m_pos)
    m2 = s_pos + slope * slope * (mean * mean + std * std - s_pos)
    return m1, m2


def _erf(x):
    # Abramowitz-Stegun 7.1.26, |err| < 1.5e-7: plenty for the moment bookkeeping (numpy has no erf)
    sign = np.sign(x); x = np.abs(x)
    t = 1.0 / (1.0 + 0.3275911 * x)
    y = 1.0 - (((((1.061405429 * t - 1.453152027) * t) + 1.421413741) * t - 0.284496736) * t + 0.254829592) * t * np.exp(-x * x)
    return sign * y


def bn_real_vectors(secs, path=None):
    """The batch-norm vectors of a TRAINED yolov3.weights / yolov2.weights as the reference itself printed them (D2T/log.txt:224-949 /
    :1-222 through DN/parser.c:1176-1228; parsed into tests/golden/yolov{3,2}_bn_real.npz by tools/make_golden.py): one dict
    {beta, gamma, mean, var, w_first} per batch-normalised conv of `secs` in file order (None for the plain head convs).  The
    topology is recognised by its filter counts; anything else raises."""
    import os
    convs = [s for s in secs[1:] if s["type"] == "convolutional"]
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
    names = [path] if path else [os.path.join(golden, "yolov3_bn_real.npz"), os.path.join(golden, "yolov2_bn_real.npz")]
    missing = [n for n in names if not os.path.exists(n)]
    if len(missing) == len(names):      # a test-only generator: the fixtures live in the source tree's tests/golden/, not in an installed package
        raise FileNotFoundError("the real batch-norm vectors are test fixtures (%s): not found -- run from the source tree or pass path=" % ", ".join(missing))
    for name in names:
        if not os.path.exists(name):
            continue
        z = np.load(name)
        if int(z["n_conv"]) == len(convs) and all(int(c["filters"]) == int(f) for c, f in zip(convs, z["filters"])):
            return [({q: z["%s_%d" % (q, i)] for q in ("beta", "gamma", "mean", "var", "w_first")} if z["bn"][i] else None) for i in range(len(convs))]
    raise ValueError("no real batch-norm vectors for this topology (%d convs): the reference's dump covers yolov3 and yolov2 only" % len(convs))


def synth_weights_real(secs, seed=0, obj_bias=-0.75, path=None):
    """The trained-file stand-in built from the reference's REAL vectors: beta, gamma and rolling variance of every
    batch-normalised conv are the file's own, per channel and PAIRED as trained (e.g. the first conv's beta -4.3 goes with gamma 2.6,
    D2T/log.txt:225-227 -- that pairing decides which channels are dead after the leaky ReLU); the filters, which the dump does not hold,
    are random directions scaled per output channel so that the conv output has the file's variance on inputs with the analytically
    tracked moments, and the rolling mean is what those filters produce.  oracle.calibrate_bn_statistics(..., keep_var=True) then
    re-scales filters and means on actual images, leaving beta / gamma / variance as the file has them."""
    return synth_weights_log(secs, seed, obj_bias, real=bn_real_vectors(secs, path))


def synth_weights_log(secs, seed=0, obj_bias=-0.75, real=None):
    """Seeded synthetic parameters with the batch-norm statistics of a TRAINED darknet file -- the ranges the reference's own dump of
    yolov3.weights / yolov2.weights documents (D2T/log.txt:1-949): gamma 0.0017 .. 4.7 and a few per cent of them negative, beta out to
    -11, rolling means out to +-11, rolling variances from 8e-4 (first layers) to 19 -- instead of the benign gamma ~ U(.8, 1.2),
    var ~ 1 of `synth_weights`.  The network must still behave like a trained one (activations O(1) through 75 layers), so the rolling
    statistics are not drawn freely: filters are scaled per output channel to hit a log-uniform target variance, and rolling mean /
    variance are then what those filters PRODUCE on inputs with the per-channel mean / variance tracked analytically through conv,
    BN, leaky ReLU, shortcut, route, upsample and pooling (independence between channels and taps assumed) -- what training's running
    averages would have recorded, within ten or twenty per cent."""
    rng = np.random.default_rng(seed)
    layers = secs[1:]
    shapes = layer_shapes(secs)
    cin0 = int(secs[0].get("channels", 3))
    mean = [None] * len(layers); var = [None] * len(layers)            # per-channel first / second central moment of every layer's output
    m_in, v_in = np.full(cin0, 0.5), np.full(cin0, 1.0 / 12.0)         # uniform [0, 1) pixels
    parts = []
    ci = -1
    for i, s in enumerate(layers):
        t = s["type"]
        pm, pv = (mean[i - 1], var[i - 1]) if i else (m_in, v_in)
        if t == "convolutional":
            ci += 1
            n, k, cin = int(s["filters"]), int(s["size"]), shapes[i][4]
            head = i + 1 < len(layers) and layers[i + 1]["type"] in ("yolo", "region")
            w = rng.normal(0, 1.0, (n, cin, k * k))
            ex2 = pv + pm * pm
            if int(s.get("batch_normalize", 0)) and real is not None:
                r = real[ci]
                target = np.maximum(r["var"].astype(np.float64), 1e-30)
                unit_var = (w * w * pv[None, :, None]).sum((1, 2))
                w *= np.sqrt(target / np.maximum(unit_var, 1e-30))[:, None, None]
                rmean = (w * pm[None, :, None]).sum((1, 2))
                gamma, beta = r["gamma"].astype(np.float64), r["beta"].astype(np.float64)
                parts += [beta, gamma, rmean, target, w.reshape(-1)]
                std = np.abs(gamma) * np.sqrt(target / (target + 1e-5))
                m1, m2 = _leaky_moments(beta, std, 0.1 if s.get("activation", "linear") == "leaky" else 1.0)
            elif int(s.get("batch_normalize", 0)):
                # target rolling variance: first two convs (pixel inputs, small filters) 2e-3 .. 0.3, later layers 0.6 .. 19, 3 % tiny
                lo, hi = (2e-3, 0.3) if i < 2 else (0.6, 19.0)
                target = np.exp(rng.uniform(np.log(lo), np.log(hi), n))
                tiny = rng.random(n) < 0.03
                target[tiny] = np.exp(rng.uniform(np.log(8e-4), np.log(1e-2), int(tiny.sum())))
                unit_var = (w * w * pv[None, :, None]).sum((1, 2))
                w *= np.sqrt(target / np.maximum(unit_var, 1e-30))[:, None, None]
                rmean = (w * pm[None, :, None]).sum((1, 2))                                  # what the conv produces on average
                rvar = target * rng.uniform(0.9, 1.1, n)                                     # running average vs this batch
                gamma = np.exp(rng.normal(0.25, 0.45, n)); gamma = np.clip(gamma, 0.0017, 4.7)
                gamma *= np.where(rng.random(n) < 0.06, -1.0, 1.0)
                beta = np.where(rng.random(n) < 0.12, rng.normal(0, 4.0, n), rng.normal(0, 1.0, n)).clip(-11.2, 4.5)
                parts += [beta, gamma, rmean, rvar, w.reshape(-1)]
                std = np.abs(gamma) * np.sqrt(target / (rvar + 1e-5))
                m1, m2 = _leaky_moments(beta, std, 0.1 if s.get("activation", "linear") == "leaky" else 1.0)
            else:
                b = rng.normal(0, 1.0, n)
                if head:
                    h = layers[i + 1]
                    classes = int(h.get("classes", 20))
                    na = len(h["mask"].split(",")) if "mask" in h else int(h.get("num", 1))
                    attrs = 5 + classes
                    if na * attrs == n:
                        b = b.reshape(na, attrs)
                        b[:, 0:4] = rng.normal(0, .5, (na, 4))
                        b[:, 4] = obj_bias + rng.normal(0, .5, na)
                        b = b.reshape(-1)
                w *= np.sqrt(1.0 / np.maximum((ex2[None, :, None] * np.ones((1, 1, k * k))).sum(), 1e-30))
                parts += [b, w.reshape(-1)]
                m1, m2 = b + (w * pm[None, :, None]).sum((1, 2)), None
                v_out = (w * w * pv[None, :, None]).sum((1, 2))
                m2 = v_out + m1 * m1
            mean[i] = m1; var[i] = np.maximum(m2 - m1 * m1, 1e-12)
        elif t == "shortcut":
            f = int(s["from"]); f = f if f >= 0 else i + f
            mean[i] = mean[i - 1] + mean[f]; var[i] = var[i - 1] + var[f]
        elif t == "route":
            ls = [int(x) for x in s["layers"].split(",")]
            ls = [l if l >= 0 else i + l for l in ls]
            mean[i] = np.concatenate([mean[l] for l in ls]); var[i] = np.concatenate([var[l] for l in ls])
        elif t == "reorg":
            st = int(s.get("stride", 1)); mean[i] = np.tile(pm, st * st); var[i] = np.tile(pv, st * st)
        elif t == "maxpool":
            mean[i] = pm + 0.8 * np.sqrt(pv); var[i] = 0.6 * pv            # rough: max of a few correlated samples
        elif t in ("upsample", "dropout"):
            mean[i] = pm; var[i] = pv
        elif t in ("yolo", "region", "detection"):
            mean[i] = pm; var[i] = pv
        else:
            raise ValueError("synth_weights_log: unsupported layer type [%s]" % t)
    return np.concatenate([np.asarray(q, dtype=np.float64).reshape(-1) for q in parts]).astype(np.float32)


def pair_closure(secs, want):
    """Mixed fp16 / split-fp16 plans: `want` is a set of layer indices (-1 = the network input) whose tensors should be
    stored as split-fp16 pairs.  Only [convolutional] sections (and the image) carry the choice -- layers that move data (shortcut, route,
    upsample, pooling, reorg) inherit their operands' form, as on the device (csrc/yolo_plan.cpp) --, and both operands of a shortcut / all
    inputs of a concatenation must share one form: the set is grown until it is closed.  Heads ([yolo] producers) are fp32, never pairs.
    Returns {layer index: bool} for every layer and -1."""
    L = secs[1:]
    S = set(int(l) for l in want if int(l) < 0 or L[int(l)]["type"] == "convolutional")

    def inputs(i):
        s = L[i]; t = s["type"]
        if t == "shortcut":
            f = int(s["from"]); return [i - 1, f if f >= 0 else i + f]
        if t == "route":
            return [int(x) if int(x) >= 0 else i + int(x) for x in s["layers"].split(",")]
        return [i - 1]

    def head(i):
        return i >= 0 and i + 1 < len(L) and L[i + 1]["type"] in ("yolo", "region", "detection")

    def is_pair(i):
        if i < 0:
            return -1 in S
        t = L[i]["type"]
        if t in ("convolutional", "connected", "local"):
            return i in S and not head(i)
        if t in ("yolo", "region", "detection"):
            return False
        return is_pair(inputs(i)[0])

    def force(i):
        if i < 0 or L[i]["type"] == "convolutional":
            S.add(i)
        elif L[i]["type"] not in ("yolo", "region", "detection"):
            for j in inputs(i):
                force(j)

    for l in want:                        # a mover asked for: its producers
        if int(l) >= 0 and L[int(l)]["type"] != "convolutional":
            force(int(l))
    changed = True
    while changed:
        changed = False
        for i, s in enumerate(L):
            if s["type"] == "shortcut" or (s["type"] == "route" and "," in s["layers"]):
                flags = [is_pair(j) for j in inputs(i)]
                if any(flags) and not all(flags):
                    for j in inputs(i):
                        force(j)
                    changed = True
    return {i: is_pair(i) for i in range(-1, len(L))}


def with_layer_pairs(text, pair):
    """cfg text of a split-fp16 (YOLO_FP16X2) network with the storage form of every tensor written out: `yolo_pair=0` on the [convolutional]
    sections whose output is PLAIN fp16 and `yolo_pair_input=0` in [net] when the image is, per `pair` = pair_closure(secs, want)
    ({layer index: bool}); sections not mentioned stay pairs (the device's default).  The convs that read a plain tensor then run one MFMA
    product per algorithmic one instead of three, and the fused kernels of the fp16 configuration serve the plain stretches."""
    out, idx = [], -2
    for line in text.splitlines():
        t = line.strip()
        if t.split("=")[0].strip() in ("yolo_pair", "yolo_pair_input"):
            continue
        out.append(line)
        if t.startswith("["):
            idx += 1
            name = t[1:t.index("]")].strip()
            if idx == -1 and not pair.get(-1, True):
                out.append("yolo_pair_input=0")
            elif idx >= 0 and name == "convolutional" and not pair.get(idx, True):
                out.append("yolo_pair=0")
    return "\n".join(out) + "\n"


def pair_flop_share(secs, pair):
    """Share of the conv FLOPs whose conv reads a tensor stored as pairs (three MFMA products per algorithmic one)."""
    shapes = layer_shapes(secs)
    tot = three = 0.0
    for i, s in enumerate(secs[1:]):
        if s["type"] == "convolutional":
            f = 2.0 * int(s["size"]) ** 2 * shapes[i][4] * int(s["filters"]) * shapes[i][1] * shapes[i][2]
            tot += f; three += f if pair.get(i - 1, False) else 0.0
    return three / tot


def default_header(secs):
    """(major, minor) the reference's files carry: v3 family -> (0,2) 64-bit seen; region (v2) -> (0,1)."""
    return (0, 2) if any(s["type"] == "yolo" for s in secs) else (0, 1)


def v1_classes():
    """Pascal VOC class names in the order of V1/YOLO_V1_Inference.py:40-44."""
    return ["aeroplane", "bicycle", "bird", "boat", "bottle", "bus", "car", "cat", "chair", "cow", "diningtable", "dog", "horse",
            "motorbike", "person", "pottedplant", "sheep", "sofa", "train", "tvmonitor"]
