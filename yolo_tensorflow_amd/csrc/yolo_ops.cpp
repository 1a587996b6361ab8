// Single-operator entry points of libyolo_hip.so (yolo_op_*): one kernel on caller-provided host tensors, for the parity tests.
#include "yolo_ctx.h"

namespace yolo_impl {

// a throw-away context for the single-operator entry points
struct OpScope {
    hipStream_t s = nullptr; std::vector<void *> bufs; int rc = YOLO_OK; std::string err;
    explicit OpScope(int device) { if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&s) != hipSuccess) rc = YOLO_ERR_HIP; }
    ~OpScope() { for (void *p : bufs) hipFree(p); if (s) hipStreamDestroy(s); }
    void *alloc(size_t bytes) { void *p = nullptr; if (hipMalloc(&p, bytes + 256) != hipSuccess) { rc = YOLO_ERR_NOMEM; return nullptr; } hipMemsetAsync(p, 0, bytes + 256, s); bufs.push_back(p); return p; }
    void *upload(const void *h, size_t bytes) { void *p = alloc(bytes); if (p && hipMemcpyAsync(p, h, bytes, hipMemcpyHostToDevice, s) != hipSuccess) rc = YOLO_ERR_HIP; return p; }
    int download(void *h, const void *d, size_t bytes) { if (hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) rc = YOLO_ERR_HIP; return rc; }
    bool ok(hipError_t e) { if (e != hipSuccess) { rc = YOLO_ERR_HIP; err = hipGetErrorString(e); } return e == hipSuccess; }
};
thread_local std::string g_op_err;

TView make_view(void *p, int n, int h, int w, int c, int stride, int dt) { TView v; v.ptr = p; v.n = n; v.h = h; v.w = w; v.c = c; v.stride = stride; v.dt = dt; return v; }


}  // namespace yolo_impl

extern "C" {

// ---- single operators -----------------------------------------------------------------------
// yolo_op_conv2d for dtype YOLO_FP16X2: x (and the residual) enter as interleaved split-fp16 pairs (32 hi | 32 lo per 32-channel group), the
// filters as W_hi 32 | W_lo 32 rows, the conv runs the pair K loop with the SPLIT epilogue, the shortcut (if any) folded into it or
// (YOLO_SPLIT_UNFUSED) as the separate k_add_split launch of that configuration; the result is joined to fp32
static int op_conv2d_split(const float *x, int n, int h, int w, int cin, const float *w_hwio, const float *bias, int k, int stride,
                           int cout, int act, const float *residual, float *out, int tile_cfg, int device)
{
    if (cin % 8 || cout % 8) { g_op_err = "conv2d (fp16x2): channel counts must be multiples of 8"; return YOLO_ERR_INVALID; }
    OpScope S(device); if (S.rc) { g_op_err = "conv2d: no HIP device"; return S.rc; }
    const int cpi = roundup(cin, 32), cpo = roundup(cout, 32);
    Layer L; L.type = L_CONV; L.filters = cout; L.size = k; L.stride = stride; L.pad = k / 2; L.bn = 0; L.act = act; L.in_dt = DT_F16;
    L.cin = cin; L.cin_pad = 2 * cpi; L.kpad = k * k * L.cin_pad; L.cout_pad = roundup(cout, 256);
    const int ho = (h + 2 * L.pad - k) / stride + 1, wo = (w + 2 * L.pad - k) / stride + 1;
    std::vector<float> oihw((size_t)cout * cin * k * k), b0(cout, 0.f);
    for (int kh = 0; kh < k; ++kh) for (int kw = 0; kw < k; ++kw) for (int ci = 0; ci < cin; ++ci) for (int o = 0; o < cout; ++o)
        oihw[(((size_t)o * cin + ci) * k + kh) * k + kw] = w_hwio[(((size_t)kh * k + kw) * cin + ci) * cout + o];
    if (bias) memcpy(b0.data(), bias, (size_t)cout * 4);
    std::vector<uint8_t> wbuf; std::vector<float> bv, osc; pack_conv(L, b0.data(), oihw.data(), DT_F16, nullptr, wbuf, bv, osc, YOLO_SEM_TF, 2);
    void *d_w = S.upload(wbuf.data(), wbuf.size()); float *d_b = (float *)S.upload(bv.data(), bv.size() * 4);
    const size_t pin = (size_t)n * h * w, pout = (size_t)n * ho * wo;
    // fp32 images padded to whole 32-channel groups (zeros): the split kernels read whole groups
    auto upload_padded = [&](const float *src, size_t npix, int c, int cp) -> float * {
        float *d = (float *)S.alloc(npix * cp * 4);
        if (d && hipMemcpy2DAsync(d, (size_t)cp * 4, src, (size_t)c * 4, (size_t)c * 4, npix, hipMemcpyHostToDevice, S.s) != hipSuccess) S.rc = YOLO_ERR_HIP;
        return d;
    };
    float *d_x32 = upload_padded(x, pin, cin, cpi);
    void *d_x = S.alloc(pin * 2 * cpi * 2), *d_o = S.alloc(pout * 2 * cpo * 2), *d_z = S.alloc(4096);
    float *d_o32 = (float *)S.alloc(pout * cpo * 4);
    if (S.rc) { g_op_err = "conv2d: allocation failed"; return S.rc; }
    if (!S.ok(launch_split_from_f32(d_x32, cpi, d_x, 2 * cpi, cpi, pin, S.s))) { g_op_err = S.err; return S.rc; }
    ConvArgs a; memset(&a, 0, sizeof a);
    a.in = d_x; a.in_stride = 2 * cpi; a.wt = d_w; a.bias = d_b; a.out = d_o; a.out_stride = 2 * cpo; a.out_dt = DT_F16; a.in_dt = DT_F16;
    a.split = 1; a.pairk = 1; a.out_inv_scale = a.res_scale = a.mid_scale = a.mid_inv_scale = 1.f;
    a.N = n; a.H = h; a.W = w; a.Cin_pad = L.cin_pad; a.Ho = ho; a.Wo = wo; a.Cout = cout;
    a.ksize = k; a.stride = stride; a.pad = L.pad; a.Kpad = L.kpad; a.kchunk = conv_kchunk(L.cin_pad, DT_F16); a.act = act; a.zeros = d_z;
    conv_finalize(a);
    int cfg = tile_cfg >= 0 ? tile_cfg : 6;
    if (!conv_cfg_pairk_ok(cfg, true) || (conv_cfg_is_halo(cfg) && !conv_halo_cfg_ok(a, cfg))) { g_op_err = "conv2d (fp16x2): tile config not instantiated for split storage / not applicable to this shape"; return YOLO_ERR_UNSUPPORTED; }
    // the shortcut: fused into the conv's epilogue, or (YOLO_SPLIT_UNFUSED, the parity tests' A/B) as the separate launch a keep_layers plan makes
    void *d_r = nullptr;
    if (residual) {
        float *d_r32 = upload_padded(residual, pout, cout, cpo); d_r = S.alloc(pout * 2 * cpo * 2);
        if (S.rc) return S.rc;
        if (!S.ok(launch_split_from_f32(d_r32, cpo, d_r, 2 * cpo, cpo, pout, S.s))) { g_op_err = S.err; return S.rc; }
        if (!getenv("YOLO_SPLIT_UNFUSED")) { a.res = d_r; a.res_stride = 2 * cpo; }
    }
    if (!S.ok(launch_conv_bf16(a, cfg, S.s))) { g_op_err = "conv2d launch: " + S.err; return S.rc; }
    if (residual && !a.res && !S.ok(launch_add_split(d_o, 2 * cpo, d_r, 2 * cpo, d_o, 2 * cpo, cpo, pout, S.s))) { g_op_err = S.err; return S.rc; }
    if (!S.ok(launch_split_to_f32(d_o, 2 * cpo, cpo, d_o32, cpo, pout, S.s))) { g_op_err = S.err; return S.rc; }
    if (hipMemcpy2DAsync(out, (size_t)cout * 4, d_o32, (size_t)cpo * 4, (size_t)cout * 4, pout, hipMemcpyDeviceToHost, S.s) != hipSuccess || hipStreamSynchronize(S.s) != hipSuccess) S.rc = YOLO_ERR_HIP;
    if (S.rc) g_op_err = "conv2d: " + std::string(hipGetErrorString(hipGetLastError()));
    return S.rc;
}

int yolo_op_conv_num_cfgs(void) { return conv_num_cfgs(); }

int yolo_op_conv2d(const float *x, int n, int h, int w, int cin, const float *w_hwio, const float *bias, int k, int stride,
                   int cout, int act, const float *residual, float *out, int dtype, int tile_cfg, int device)
{
    if (!x || !w_hwio || !out || n < 1 || (k != 1 && k != 3) || stride < 1) { g_op_err = "conv2d: bad arguments"; return YOLO_ERR_INVALID; }
    if (dtype == YOLO_FP16X2) return op_conv2d_split(x, n, h, w, cin, w_hwio, bias, k, stride, cout, act, residual, out, tile_cfg, device);
    OpScope S(device); if (S.rc) { g_op_err = "conv2d: no HIP device"; return S.rc; }
    // dtype YOLO_FP8: x, residual and the result are e4m3 tensors of scale 1 (the inputs are quantised here first)
    const bool f32 = dtype == YOLO_FP32; const int dt = f32 ? DT_F32 : dtype == YOLO_FP8 ? DT_FP8 : dtype == YOLO_FP16 ? DT_F16 : DT_BF16; const size_t es = dt_size(dt);
    const int gr = dt == DT_FP8 ? 16 : 8;
    Layer L; L.type = L_CONV; L.filters = cout; L.size = k; L.stride = stride; L.pad = k / 2; L.bn = 0; L.act = act; L.in_dt = dt;
    L.cin = cin; L.cin_pad = roundup(cin, gr); L.kpad = roundup(k * k * L.cin_pad, dt == DT_FP8 ? 128 : 64); L.cout_pad = roundup(cout, 256);
    const int ho = (h + 2 * L.pad - k) / stride + 1, wo = (w + 2 * L.pad - k) / stride + 1;
    // HWIO -> OIHW for the common packer
    std::vector<float> oihw((size_t)cout * cin * k * k), b0(cout, 0.f);
    for (int kh = 0; kh < k; ++kh) for (int kw = 0; kw < k; ++kw) for (int ci = 0; ci < cin; ++ci) for (int o = 0; o < cout; ++o)
        oihw[(((size_t)o * cin + ci) * k + kh) * k + kw] = w_hwio[(((size_t)kh * k + kw) * cin + ci) * cout + o];
    if (bias) memcpy(b0.data(), bias, (size_t)cout * 4);
    std::vector<uint8_t> wbuf; std::vector<float> bv, osc; pack_conv(L, b0.data(), oihw.data(), dt, nullptr, wbuf, bv, osc);
    void *d_w = S.upload(wbuf.data(), wbuf.size()); float *d_b = (float *)S.upload(bv.data(), bv.size() * 4);
    float *d_sc = dt == DT_FP8 ? (float *)S.upload(osc.data(), osc.size() * 4) : nullptr;
    float *d_x32 = (float *)S.upload(x, (size_t)n * h * w * cin * 4);
    void *d_x = S.alloc((size_t)n * h * w * L.cin_pad * es);
    const int cstride = roundup(cout, gr);
    void *d_o = S.alloc((size_t)n * ho * wo * cstride * es); float *d_o32 = (float *)S.alloc((size_t)n * ho * wo * cout * 4);
    void *d_r = nullptr;
    void *d_z = S.alloc(4096);
    if (S.rc) { g_op_err = "conv2d: allocation failed"; return S.rc; }
    TView vx = make_view(d_x, n, h, w, cin, L.cin_pad, dt);
    if (!S.ok(launch_from_f32(d_x32, vx, S.s))) { g_op_err = S.err; return S.rc; }
    if (residual) {
        float *d_r32 = (float *)S.upload(residual, (size_t)n * ho * wo * cout * 4); d_r = S.alloc((size_t)n * ho * wo * cstride * es);
        if (S.rc) return S.rc;
        if (!S.ok(launch_from_f32(d_r32, make_view(d_r, n, ho, wo, cout, cstride, dt), S.s))) { g_op_err = S.err; return S.rc; }
    }
    ConvArgs a; memset(&a, 0, sizeof a);
    a.in = d_x; a.in_stride = L.cin_pad; a.wt = d_w; a.bias = d_b; a.out = d_o; a.out_stride = cstride; a.out_dt = dt; a.in_dt = dt;
    a.oscale = d_sc; a.out_inv_scale = a.res_scale = a.mid_scale = a.mid_inv_scale = 1.f;
    a.res = d_r; a.res_stride = cstride; a.N = n; a.H = h; a.W = w; a.Cin_pad = L.cin_pad; a.Ho = ho; a.Wo = wo; a.Cout = cout;
    a.ksize = k; a.stride = stride; a.pad = L.pad; a.Kpad = L.kpad; a.kchunk = conv_kchunk(L.cin_pad, dt); a.act = act; a.zeros = d_z;
    conv_finalize(a);
    if (conv_cfg_is_halo(tile_cfg) && (f32 || !conv_halo_cfg_ok(a, tile_cfg))) { g_op_err = "conv2d: tile config not applicable to this shape (halo-staged form: 3x3, stride 1, size a multiple of 13, whole channel chunks)"; return YOLO_ERR_UNSUPPORTED; }
    hipError_t e;
    if (dt != DT_F32 && dt != DT_F16 && getenv("YOLO_CONV_DIAG") && a.Cin_pad % (dt == DT_FP8 ? 128 : 64) == 0) {
        // developer diagnostic: phase cycle sums of the stamped p176c128_s2 build (YOLO_CONV_DIAG=free: of the free-running halo form
        // f176c256), printed to stderr
        const bool dwide = !strcmp(getenv("YOLO_CONV_DIAG"), "free4");        // four waves of 176 x 64
        const bool dfree = (dwide || !strcmp(getenv("YOLO_CONV_DIAG"), "free")) && conv_halo13_ok(a) && dt == DT_BF16;
        const int wv = dfree && !dwide ? 8 : 4;
        const long tiles = dfree ? (long)n * ((h + 12) / 13) * ((w + 12) / 13) * ((cout + 255) / 256) : (((long)n * ho * wo + 175) / 176) * ((cout + 127) / 128);
        a.dbg = (unsigned long long *)S.alloc((size_t)tiles * wv * 16 * 8);
        a.dbg_light = getenv("YOLO_CONV_DIAG_LIGHT") ? 1 : 0;      // stamps around the K loop only (the clock measurement)
        if (dfree && !dwide && getenv("YOLO_CONV_DIAG_TAIL") && cout == 256) {
            // time the fused 1x1 tail too: any 128 x 256 filter block will do (the main filters' first rows), output to scratch
            a.w2 = a.wt; a.w2f = a.wt; a.K2pad = a.Kpad; a.b2 = a.bias; a.act2 = ACT_LEAKY; a.out2_stride = 128;
            a.out2 = S.alloc((size_t)n * ho * wo * 128 * 2);
        }
        // long enough for the clock to settle under load: the CDNA4 guide asks for >= 2 s of back-to-back launches before the stamps are read
        // (YOLO_CONV_DIAG_REPS; the stamps of the LAST launch are what is downloaded)
        const int reps = getenv("YOLO_CONV_DIAG_REPS") ? atoi(getenv("YOLO_CONV_DIAG_REPS")) : 200;
        for (int rep = 0; rep < reps; ++rep) e = dfree ? launch_conv_halo13_diag(a, S.s, dwide ? 1 : 0) : launch_conv_diag(a, S.s);
        std::vector<unsigned long long> hd((size_t)tiles * wv * 16);
        S.download(hd.data(), a.dbg, hd.size() * 8);
        double sum[16] = {0}; size_t cnt = hd.size() / 16;
        const unsigned long long kt = hd[5] >> 40;
        for (size_t i = 0; i < cnt; ++i)
            for (int q = 0; q < 16; ++q) sum[q] += q == 5 ? (double)(hd[i * 16 + 5] & 0xffffffffffull) : (double)hd[i * 16 + q];
        for (double &v : sum) v /= cnt;
        std::vector<unsigned long long> kclk(cnt); for (size_t i = 0; i < cnt; ++i) kclk[i] = hd[i * 16 + 15];
        std::nth_element(kclk.begin(), kclk.begin() + cnt / 2, kclk.end()); const double kmed = cnt ? (double)kclk[cnt / 2] : 0.0;
        fprintf(stderr, "diag: waves %zu KT %llu | per K-step cycles: wait+barrier %.0f  issue %.0f  ds_read+mfma %.0f  (loop total/KT %.0f) | epilogue %.0f cycles | shader clock %.0f MHz (K loop alone %.0f MHz, median over waves %.0f)\n",
                cnt, kt, sum[0] / kt, sum[1] / kt, sum[2] / kt, sum[3] / kt, sum[4], sum[5], sum[15], kmed);
        fprintf(stderr, "diag: setup (first instruction -> prologue issued) %.0f | first wait (prologue data + barrier) %.0f | epilogue: barrier %.0f  acc->LDS + barrier %.0f  shortcut add + store issue %.0f  store drain %.0f\n",
                sum[6], sum[7], sum[8], sum[9], sum[10], sum[11]);
        if (a.w2) fprintf(stderr, "diag: fused 1x1 tail: barrier %.0f  fragments + MFMA + pack %.0f  barrier %.0f (then the tail's stores, in `store drain`)\n", sum[12], sum[13], sum[14]);
    } else
        e = f32 ? launch_conv_f32(a, S.s)
                : dt == DT_FP8 ? launch_conv_fp8(a, tile_cfg >= 0 ? tile_cfg : conv_pick_cfg(a), S.s)
                               : launch_conv_bf16(a, tile_cfg >= 0 ? tile_cfg : conv_pick_cfg(a), S.s);
    if (!S.ok(e)) { g_op_err = "conv2d launch: " + S.err; return S.rc; }
    if (!S.ok(launch_to_f32(make_view(d_o, n, ho, wo, cout, cstride, dt), d_o32, S.s))) { g_op_err = S.err; return S.rc; }
    S.download(out, d_o32, (size_t)n * ho * wo * cout * 4);
    if (S.rc) g_op_err = "conv2d: " + std::string(hipGetErrorString(hipGetLastError()));
    return S.rc;
}

static int ew_op(int kind, const float *x, int n, int h, int w, int c, int p0, int p1, int p2, float *out, int device)
{
    if (!x || !out || c % 8) { g_op_err = "op: bad arguments (channels must be a multiple of 8)"; return YOLO_ERR_INVALID; }
    OpScope S(device); if (S.rc) { g_op_err = "op: no HIP device"; return S.rc; }
    int ho = h, wo = w, co = c;
    if (kind == 0) { ho = 2 * h; wo = 2 * w; }
    else if (kind == 1) { ho = h / p0; wo = w / p0; co = c * p0 * p0; }
    else { int pad = (p0 - 1) / 2; ho = (h + 2 * pad) / p1; wo = (w + 2 * pad) / p1; }
    float *d_x32 = (float *)S.upload(x, (size_t)n * h * w * c * 4);
    void *d_x = S.alloc((size_t)n * h * w * c * 2), *d_o = S.alloc((size_t)n * ho * wo * co * 2);
    float *d_o32 = (float *)S.alloc((size_t)n * ho * wo * co * 4);
    if (S.rc) return S.rc;
    TView vi = make_view(d_x, n, h, w, c, c, 0), vo = make_view(d_o, n, ho, wo, co, co, 0);
    bool ok = S.ok(launch_from_f32(d_x32, vi, S.s));
    if (ok && kind == 0) ok = S.ok(launch_upsample2x(vi, vo, p0 == YOLO_SEM_TF, S.s));
    if (ok && kind == 1) ok = S.ok(launch_reorg(vi, vo, p0, p1 == YOLO_SEM_DARKNET, S.s));
    if (ok && kind == 2) ok = S.ok(launch_maxpool(vi, vo, p0, p1, (p0 - 1) / 2, S.s));
    if (ok) ok = S.ok(launch_to_f32(vo, d_o32, S.s));
    if (!ok) { g_op_err = S.err; return S.rc; }
    return S.download(out, d_o32, (size_t)n * ho * wo * co * 4);
}
int yolo_op_upsample2x(const float *x, int n, int h, int w, int c, int semantics, float *out, int device) { return ew_op(0, x, n, h, w, c, semantics, 0, 0, out, device); }
int yolo_op_reorg(const float *x, int n, int h, int w, int c, int stride, int semantics, float *out, int device) { return ew_op(1, x, n, h, w, c, stride, semantics, 0, out, device); }
int yolo_op_maxpool(const float *x, int n, int h, int w, int c, int size, int stride, float *out, int device) { return ew_op(2, x, n, h, w, c, size, stride, 0, out, device); }

int yolo_op_resize_u8(const uint8_t *img, int h, int w, int s, float post_scale, float *out, int device)
{
    if (!img || !out || h < 1 || w < 1 || s < 1) return YOLO_ERR_INVALID;
    OpScope S(device); if (S.rc) return S.rc;
    uint8_t *d_i = (uint8_t *)S.upload(img, (size_t)h * w * 3); float *d_o = (float *)S.alloc((size_t)s * s * 3 * 4);
    if (S.rc) return S.rc;
    if (!S.ok(launch_resize_u8(d_i, h, w, s, d_o, 1, 3, 3, S.s, post_scale))) { g_op_err = S.err; return S.rc; }
    return S.download(out, d_o, (size_t)s * s * 3 * 4);
}

int yolo_op_resize_cv2(const uint8_t *img, int h, int w, int oh, int ow, int swap_rb, float divisor, float *out, int device)
{
    if (!img || !out || h < 1 || w < 1 || oh < 1 || ow < 1 || !(divisor != 0.f)) { g_op_err = "resize_cv2: bad arguments"; return YOLO_ERR_INVALID; }
    OpScope S(device); if (S.rc) return S.rc;
    uint8_t *d_i = (uint8_t *)S.upload(img, (size_t)h * w * 3); float *d_o = (float *)S.alloc((size_t)oh * ow * 3 * 4);
    if (S.rc) return S.rc;
    if (!S.ok(launch_resize_cv2_u8(d_i, h, w, oh, ow, swap_rb, divisor, d_o, S.s))) { g_op_err = S.err; return S.rc; }
    return S.download(out, d_o, (size_t)oh * ow * 3 * 4);
}

int yolo_op_letterbox(const float *image_chw, int iw, int ih, int w, int h, int embed, float *out_chw, int device)
{
    if (!image_chw || !out_chw || iw < 1 || ih < 1 || w < 1 || h < 1) { g_op_err = "letterbox: bad arguments"; return YOLO_ERR_INVALID; }
    OpScope S(device); if (S.rc) { g_op_err = "letterbox: no HIP device"; return S.rc; }
    float *d_i = (float *)S.upload(image_chw, (size_t)iw * ih * 3 * 4), *d_o = (float *)S.alloc((size_t)w * h * 3 * 4);
    if (S.rc) return S.rc;
    if (!S.ok(launch_letterbox_planar(d_i, iw, ih, w, h, embed, d_o, S.s))) { g_op_err = S.err; return S.rc; }
    return S.download(out_chw, d_o, (size_t)w * h * 3 * 4);
}

int yolo_op_decode(const float *raw, int n, int g, int na, int classes, const float *anchors_wh, int img_size, int decode,
                   int region, float *out, int device)
{
    if (!raw || !out || !anchors_wh || na < 1 || na > 16) return YOLO_ERR_INVALID;
    OpScope S(device); if (S.rc) return S.rc;
    const int attrs = 5 + classes; const size_t cnt = (size_t)n * g * g * na * attrs;
    float *d_r = (float *)S.upload(raw, cnt * 4), *d_o = (float *)S.alloc(cnt * 4);
    if (S.rc) return S.rc;
    DecodeArgs d; memset(&d, 0, sizeof d);
    d.raw = d_r; d.raw_stride = na * attrs; d.n = n; d.g = g; d.na = na; d.classes = classes; d.img_size = img_size; d.mode = decode; d.region = region;
    const int stride = img_size / g;
    for (int k = 0; k < 2 * na; ++k) d.anchors[k] = region ? anchors_wh[k] : (float)(1.0 * (double)anchors_wh[k] / (double)stride);
    d.det = d_o; d.rows_total = g * g * na; d.row_off = 0; d.reject_below = -INFINITY;
    if (!S.ok(launch_decode(d, nullptr, nullptr, S.s))) { g_op_err = S.err; return S.rc; }
    return S.download(out, d_o, cnt * 4);
}

int yolo_op_detections_boxes(const float *det, int n, int rows, int attrs, float *out, int device)
{
    if (!det || !out || n < 1 || rows < 1 || attrs < 5) return YOLO_ERR_INVALID;
    OpScope S(device); if (S.rc) return S.rc;
    const size_t cnt = (size_t)n * rows * attrs;
    float *d_i = (float *)S.upload(det, cnt * 4), *d_o = (float *)S.alloc(cnt * 4);
    if (S.rc) return S.rc;
    if (!S.ok(launch_boxes_to_corners(d_i, d_o, (size_t)n * rows, attrs, S.s))) { g_op_err = S.err; return S.rc; }
    return S.download(out, d_o, cnt * 4);
}

int yolo_op_nms_detections(const float *boxes_xywh, float *prob, float *objectness, int n, int classes, float thresh, int by_objectness, int device)
{
    if (n == 0) return YOLO_OK;
    if (!boxes_xywh || !prob || !objectness || n < 0 || classes < 1) { g_op_err = "nms_detections: bad arguments"; return YOLO_ERR_INVALID; }
    if (n > 4096) { g_op_err = "nms_detections: more than 4096 detections"; return YOLO_ERR_UNSUPPORTED; }
    OpScope S(device); if (S.rc) { g_op_err = "nms_detections: no HIP device"; return S.rc; }
    float4 *d_b = (float4 *)S.upload(boxes_xywh, (size_t)n * 16);
    float *d_p = (float *)S.upload(prob, (size_t)n * classes * 4), *d_o = (float *)S.upload(objectness, (size_t)n * 4);
    if (S.rc) return S.rc;
    if (!S.ok(launch_nms_dets(d_b, d_p, d_o, n, classes, thresh, by_objectness ? 1 : 0, S.s))) { g_op_err = S.err; return S.rc; }
    S.download(prob, d_p, (size_t)n * classes * 4);
    return S.download(objectness, d_o, (size_t)n * 4);
}

int yolo_op_postprocess_rows(const float *det, int n, int rows, int attrs, float score_thr, float iou_thr, int max_out, int nms_mode,
                             int select_mode, yolo_box *boxes_out, int32_t *counts_out, int32_t *rows_out, int device)
{
    if (!det || !boxes_out || !counts_out || n < 1 || rows < 1 || rows > 32768 || attrs < 6 || max_out < 1) return YOLO_ERR_INVALID;
    OpScope S(device); if (S.rc) return S.rc;
    size_t nr = (size_t)n * rows; int p2 = 1; while (p2 < rows) p2 <<= 1;
    PostArgs p; memset(&p, 0, sizeof p);
    p.det = (const float *)S.upload(det, nr * attrs * 4); p.n = n; p.rows = rows; p.attrs = attrs; p.score_thr = score_thr; p.iou_thr = iou_thr;
    p.max_out = max_out; p.nms_mode = nms_mode & 0xff; p.select_mode = select_mode & 0xff; p.corners_in = (select_mode >> 8) & 1;
    // bits 8.. of nms_mode carry the image size for the V2 numpy flavour: (h << 8) | (w << 20)
    p.img_h = (nms_mode >> 8) & 0xfff; p.img_w = (nms_mode >> 20) & 0xfff;
    p.scores = (float *)S.alloc(nr * 4); p.labels = (int *)S.alloc(nr * 4); p.cand = (int *)S.alloc(nr * 4);
    p.keys = (unsigned long long *)S.alloc((size_t)n * p2 * 8); p.rows_pow2 = p2;
    p.sbox = (float4 *)S.alloc(nr * 16); p.slabel = (int *)S.alloc(nr * 4); p.sscore = (float *)S.alloc(nr * 4);
    p.boxes_out = S.alloc((size_t)n * max_out * sizeof(yolo_box)); p.counts_out = (int *)S.alloc((size_t)n * 4);
    if (rows_out) { p.srow = (int *)S.alloc(nr * 4); p.rows_out = (int *)S.alloc((size_t)n * max_out * 4); }
    if (S.rc) return S.rc;
    if (!S.ok(launch_postprocess(p, S.s))) { g_op_err = S.err; return S.rc; }
    S.download(boxes_out, p.boxes_out, (size_t)n * max_out * sizeof(yolo_box));
    if (rows_out) S.download(rows_out, p.rows_out, (size_t)n * max_out * 4);
    return S.download(counts_out, p.counts_out, (size_t)n * 4);
}

int yolo_op_postprocess(const float *det, int n, int rows, int attrs, float score_thr, float iou_thr, int max_out, int nms_mode,
                        int select_mode, yolo_box *boxes_out, int32_t *counts_out, int device)
{
    return yolo_op_postprocess_rows(det, n, rows, attrs, score_thr, iou_thr, max_out, nms_mode, select_mode, boxes_out, counts_out, nullptr, device);
}

}  // extern "C"
