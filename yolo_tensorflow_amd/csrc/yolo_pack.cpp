// Parameters of libyolo_hip.so: batch-norm fold, filter packing (bf16 / fp16 / e4m3 / fp32 layouts), fp8 activation scales, the Darknet
// weight stream and the export artifact (8f-3).
#include "yolo_ctx.h"

namespace yolo_impl {

uint16_t f2bf(float f)
{
    uint32_t u; memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// float -> IEEE binary16 bits: round to nearest even, saturating at +-65504 (as the device's conversions do), NaN stays NaN
uint16_t f2h(float f)
{
    uint32_t u; memcpy(&u, &f, 4);
    const uint16_t sign = (uint16_t)((u >> 16) & 0x8000u);
    if (f != f) return (uint16_t)(sign | 0x7e00u);
    float a = fabsf(f);
    if (a >= 65504.f) return (uint16_t)(sign | 0x7bffu);
    if (a < ldexpf(1.f, -24) * 0.5f) return sign;                       // below half the smallest subnormal (ties to even: zero)
    int e; frexpf(a, &e); e -= 1;                                         // a in [2^e, 2^(e+1))
    if (e < -14) e = -14;                                                 // subnormals share the first binade's quantum
    const float quantum = ldexpf(1.f, e - 10);
    const float q = nearbyintf(a / quantum);                              // exact division, RNE under the default rounding mode
    const float v = q * quantum;
    if (v < ldexpf(1.f, -14)) return (uint16_t)(sign | (uint16_t)q);      // subnormal: q in 0..1023
    int e2; frexpf(v, &e2); e2 -= 1;
    const int m = (int)((v / ldexpf(1.f, e2) - 1.f) * 1024.f);
    return (uint16_t)(sign | ((e2 + 15) << 10) | m);
}

// float -> OCP e4m3 (e4m3fn) code: round to nearest even, saturate at +-448, NaN -> 0x7f
uint8_t f2e4m3(float f)
{
    uint32_t u; memcpy(&u, &f, 4);
    const uint8_t sign = (uint8_t)((u >> 31) << 7);
    if (f != f) return (uint8_t)(sign | 0x7f);
    float a = fabsf(f);
    if (a >= 448.f) return (uint8_t)(sign | 0x7e);
    int e; frexpf(a, &e); e -= 1;                          // a in [2^e, 2^(e+1))
    if (a == 0.f || e < -6) e = -6;                        // subnormal range shares the quantum of the first binade
    const float quantum = ldexpf(1.f, e - 3);
    const float q = nearbyintf(a / quantum);               // RNE under the default rounding mode; exact division
    const float v = q * quantum;
    if (v < ldexpf(1.f, -6)) return (uint8_t)(sign | (uint8_t)q);          // q in 0..7 (q == 8 is the first normal)
    int e2; frexpf(v, &e2); e2 -= 1;
    const int m = (int)((v / ldexpf(1.f, e2) - 1.f) * 8.f);
    return (uint8_t)(sign | ((e2 + 7) << 3) | m);
}

// fold + pack one conv's parameters (host).  w_oihw: [cout][cin][k][k].  wdt: element type of the packed filters.
// fp8: `in_scale` (per input channel, or null = 1) is folded into the filters first, then every output channel c is
// scaled so that its largest |w| maps to 448: code = e4m3(w * in_scale / osc[c]), osc[c] = max|w * in_scale| / 448.
// IEEE binary16 bits -> float (exact)
float h2f(uint16_t h)
{
    const int e = (h >> 10) & 31, m = h & 1023;
    float v = e == 0 ? ldexpf((float)m, -24) : e == 31 ? (m ? NAN : INFINITY) : ldexpf((float)(m | 1024), e - 25);
    return (h & 0x8000) ? -v : v;
}

void pack_conv(const Layer &L, const float *bn_or_bias, const float *w_oihw, int wdt, const float *in_scale,
               std::vector<uint8_t> &wbuf, std::vector<float> &bias, std::vector<float> &osc, int semantics, int split)
{
    const int n = L.filters, k = L.size, cin = L.cin;
    bias.assign(L.cout_pad, 0.f); osc.assign(L.cout_pad, 1.f);
    std::vector<float> scale(n, 1.f);
    if (L.bn) {
        const float *beta = bn_or_bias, *gamma = beta + n, *mean = gamma + n, *var = mean + n;
        for (int o = 0; o < n; ++o) {
            // TF: epsilon inside the sqrt (V3/yolo_v3.py:9).  darknet semantics follow the reference's CPU normalize
            // (DN/blas.c:154: (x - mean) / (sqrt(var) + .000001f)), the code oracle/_ref is compiled from
            float s = semantics == YOLO_SEM_DARKNET ? gamma[o] / (sqrtf(var[o]) + 1e-6f) : gamma[o] / sqrtf(var[o] + 1e-5f);
            scale[o] = s; bias[o] = beta[o] - mean[o] * s;
        }
    } else {
        for (int o = 0; o < n; ++o) bias[o] = bn_or_bias[o];
    }
    const size_t es = dt_size(wdt);
    wbuf.assign((size_t)L.cout_pad * L.kpad * es, 0);
    std::vector<float> row((size_t)cin * k * k);
    for (int o = 0; o < n; ++o) {
        float amax = 0.f;
        for (int ci = 0; ci < cin; ++ci)
            for (int t = 0; t < k * k; ++t) {
                float v = w_oihw[((size_t)o * cin + ci) * k * k + t] * scale[o];
                if (wdt == DT_FP8 && in_scale) v *= in_scale[ci];
                row[(size_t)ci * k * k + t] = v; amax = std::max(amax, fabsf(v));
            }
        if (wdt == DT_FP8) osc[o] = amax > 0.f ? amax / FP8_MAX : 1.f;
        const int kc = conv_kchunk(L.cin_pad, wdt);                              // K order: see conv_igemm.hip `stage`
        for (int ci = 0; ci < cin; ++ci)
            for (int t = 0; t < k * k; ++t) {
                const float v = row[(size_t)ci * k * k + t];
                if (split == 2) {
                    // split fp16 (YOLO_FP16X2), the input is an interleaved pair tensor (pair K loop): per 32-channel group and tap one 64-element
                    // K-step row W_hi 32 | W_lo 32 -- k = ((ci / 32) * k * k + t) * 64 + ci % 32 (+ 32 for the low half)
                    const uint16_t h = f2h(v); const uint16_t l = f2h(v - h2f(h));
                    const size_t idx = (size_t)o * L.kpad + ((size_t)(ci / 32) * k * k + t) * 64 + ci % 32;
                    memcpy(&wbuf[idx * 2], &h, 2); memcpy(&wbuf[(idx + 32) * 2], &l, 2);
                    continue;
                }
                if (split) {
                    // ... the input is the network image, three blocks hi | lo | hi of its 8 padded channels (plain K loop over 3 * Cp 'channels'):
                    // the filter row holds W_hi | W_hi | W_lo
                    const int Cp = L.cin_pad / 3;
                    const uint16_t h = f2h(v); const uint16_t l = f2h(v - h2f(h));
                    for (int blk = 0; blk < 3; ++blk) {
                        const int cv = blk * Cp + ci;
                        const size_t idx = (size_t)o * L.kpad + ((size_t)(cv / kc) * k * k + t) * kc + cv % kc;
                        memcpy(&wbuf[idx * 2], blk < 2 ? &h : &l, 2);
                    }
                    continue;
                }
                const size_t idx = (size_t)o * L.kpad + ((size_t)(ci / kc) * k * k + t) * kc + ci % kc;     // t = kh * k + kw
                if (wdt == DT_F32) memcpy(&wbuf[idx * 4], &v, 4);
                else if (wdt == DT_FP8) wbuf[idx] = f2e4m3(v / osc[o]);
                else { uint16_t b = wdt == DT_F16 ? f2h(v) : f2bf(v); memcpy(&wbuf[idx * 2], &b, 2); }
            }
    }
}

// fp8: scale of the tensor each layer's view holds, and per-input-channel scales of a conv
void resolve_scales(yolo_ctx *c)
{
    const int NL = (int)c->layers.size();
    if ((int)c->user_scale.size() != NL) c->user_scale.assign(NL, 1.f);
    c->eff_scale.assign(NL, 1.f);
    for (int i = 0; i < NL; ++i) {
        const Layer &L = c->layers[i];
        switch (L.type) {
        case L_CONV: c->eff_scale[i] = L.store_dt != DT_FP8 ? 1.f : (L.residual_from >= -1 && i + 1 < NL) ? c->user_scale[i + 1] : c->user_scale[i]; break;
        case L_SHORTCUT: c->eff_scale[i] = L.store_dt != DT_FP8 ? 1.f : c->user_scale[i]; break;
        case L_ROUTE: c->eff_scale[i] = L.in.size() == 1 ? c->eff_scale[L.in[0]] : NAN; break;
        case L_UPSAMPLE: case L_MAXPOOL: case L_REORG: c->eff_scale[i] = c->eff_scale[L.in[0]]; break;
        default: break;
        }
    }
}
// scale of every logical channel of layer idx's output (multi-input routes concatenate their sources)
void channel_scales(const yolo_ctx *c, int idx, std::vector<float> &out)
{
    const Layer &L = c->layers[idx];
    if (L.type == L_ROUTE && L.in.size() > 1) { for (int j : L.in) channel_scales(c, j, out); return; }
    if (L.type == L_REORG) {             // channel order is scrambled but every source channel has the same scale
        std::vector<float> src; channel_scales(c, L.in[0], src);
        for (int k = 0; k < L.C; ++k) out.push_back(src[0]);
        return;
    }
    if (L.type == L_ROUTE || L.type == L_UPSAMPLE || L.type == L_MAXPOOL) { channel_scales(c, L.in[0], out); return; }
    for (int k = 0; k < L.C; ++k) out.push_back(c->eff_scale[idx]);
}

// The fused 1x1 tail (conv_igemm_kernel.h) reads its filters as MFMA A fragments straight from global memory: lane (l15, lq) of a wave
// takes the 16 bytes at k = (kk * 4 + lq) * 8 of row ct2 * 16 + l15.  From the [row][K] image one such wave-load touches sixteen
// 64-byte pieces 2 * K bytes apart, and every workgroup of the layer asks for the same 64 KB at the same moment; a copy in fragment
// order -- [channel tile][K step][lane][8 bf16] -- makes each wave-load one contiguous KiB.  Built from the packed filters already on
// the device, so both ways of loading parameters (weight stream, export artifact) share it.
int tail_fragments(yolo_ctx *c)
{
    std::vector<uint16_t> src, dst;
    for (auto &T : c->layers) {
        if (T.type != L_CONV || T.fused_into < 0) continue;
        if (T.in_dt != DT_BF16 && T.in_dt != DT_F16) continue;          // (16-bit tails only; also the bf16 islands of a mixed e4m3 plan)
        // K == the producer's channel count, a multiple of 32.  A HEAD tail is read by all eight waves, 32 rows each, whatever its filter count
        // (conv_igemm_kernel.h, head mode: t2g = wave_id): its fragment image is always the padded 256 rows, the rows past the filters zero
        // (18 filters of a 1-class head, 75 of a VOC head: ADVICE r05 -- the short image was read past its end).
        const int C2 = T.head ? T.cout_pad : roundup(T.filters, 16), K = T.kpad;
        if (C2 > T.cout_pad || K % 32 || (T.head && T.cout_pad < 256)) continue;
        src.resize((size_t)T.cout_pad * K); dst.assign((size_t)C2 * K, 0);
        HIPCK(c, hipMemcpy(src.data(), T.d_w, src.size() * 2, hipMemcpyDeviceToHost));
        const int K2S = K / 32;
        for (int ct2 = 0; ct2 < C2 / 16; ++ct2)
            for (int kk = 0; kk < K2S; ++kk)
                for (int lane = 0; lane < 64; ++lane)
                    memcpy(&dst[(((size_t)ct2 * K2S + kk) * 64 + lane) * 8], &src[(size_t)(ct2 * 16 + (lane & 15)) * K + (kk * 4 + (lane >> 4)) * 8], 16);
        if (!T.d_wf) HIPCK(c, hipMalloc(&T.d_wf, dst.size() * 2));
        HIPCK(c, hipMemcpy(T.d_wf, dst.data(), dst.size() * 2, hipMemcpyHostToDevice));
    }
    return YOLO_OK;
}

}  // namespace yolo_impl

extern "C" {

int yolo_set_act_scales(yolo_ctx *c, const float *scales, int n)
{
    if (!c) return YOLO_ERR_INVALID;
    if (c->dtype != YOLO_FP8) return fail(c, YOLO_ERR_STATE, "activation scales only exist in the fp8 configuration");
    if (!scales || n != (int)c->layers.size()) return fail(c, YOLO_ERR_INVALID, "need one scale per layer (%zu)", c->layers.size());
    for (int i = 0; i < n; ++i) if (!(scales[i] > 0.f) || !std::isfinite(scales[i])) return fail(c, YOLO_ERR_INVALID, "layer %d: scale must be finite and > 0", i);
    c->user_scale.assign(scales, scales + n);
    resolve_scales(c);
    c->weights_loaded = false;           // filters absorb the input scales: they have to be packed again
    if (c->gexec) { hipGraphExecDestroy(c->gexec); c->gexec = nullptr; } if (c->gstate > 0) c->gstate = 0;
    return YOLO_OK;
}

int yolo_set_weights(yolo_ctx *c, const float *flat, size_t n)
{
    if (!c) return YOLO_ERR_INVALID;
    if (!flat) return fail(c, YOLO_ERR_INVALID, "weights == NULL");
    if (n != c->weights_count) return fail(c, YOLO_ERR_IO, "weights stream has %zu floats, topology needs %zu", n, c->weights_count);
    HIPCK(c, hipSetDevice(c->device));
    const float *p = flat;
    std::vector<uint8_t> wbuf; std::vector<float> bias, osc;
    resolve_scales(c);
    for (auto &L : c->layers) {
        if (L.type == L_LOCAL) {
            // file: biases [filter][location], weights [location][filter][c][kh][kw] (DN/parser.c:1315-1320) -> device: bias [location][filter],
            // weights [location][filter][kh][kw][c] in the activations' type
            const int loc = L.H * L.W, F = L.filters, k = L.size, C = L.cin;
            const float *bfile = p; p += (size_t)F * loc;
            const float *wfile = p; p += (size_t)loc * F * C * k * k;
            std::vector<float> b((size_t)loc * F);
            for (int f = 0; f < F; ++f) for (int l = 0; l < loc; ++l) b[(size_t)l * F + f] = bfile[(size_t)f * loc + l];
            const size_t es = dt_size(L.in_dt);
            std::vector<uint8_t> wb((size_t)loc * F * k * k * C * es);
            for (size_t lf = 0; lf < (size_t)loc * F; ++lf)
                for (int ch = 0; ch < C; ++ch)
                    for (int t = 0; t < k * k; ++t) {
                        const float v = wfile[(lf * C + ch) * k * k + t];
                        const size_t idx = (lf * k * k + t) * C + ch;
                        if (L.in_dt == DT_F32) memcpy(&wb[idx * 4], &v, 4);
                        else { uint16_t h = L.in_dt == DT_F16 ? f2h(v) : f2bf(v); memcpy(&wb[idx * 2], &h, 2); }
                    }
            HIPCK(c, hipMemcpy(L.d_w, wb.data(), wb.size(), hipMemcpyHostToDevice));
            HIPCK(c, hipMemcpy(L.d_b, b.data(), b.size() * 4, hipMemcpyHostToDevice));
            continue;
        }
        if (L.type != L_CONV) continue;
        const float *params = p; p += (size_t)L.filters * (L.bn ? 4 : 1);
        const float *w = p; p += (size_t)L.filters * L.cin * L.size * L.size;
        std::vector<float> in_sc;
        if (L.in_dt == DT_FP8) {
            channel_scales(c, L.in[0], in_sc);
            if ((int)in_sc.size() != L.cin) return fail(c, YOLO_ERR_STATE, "internal: scale vector of %zu for %d channels", in_sc.size(), L.cin);
        }
        std::vector<float> wperm;
        const Layer *PL = &L; Layer tmp;
        if (L.fc && L.fc_h * L.fc_w > 1) {
            // darknet / the transposed TF graph flatten the producer CHW (V1/YOLO_V1_Inference.py:196-198); the tensor here is HWC
            const int hw = L.fc_h * L.fc_w, C = L.fc_c;
            wperm.resize((size_t)L.filters * L.cin);
            for (int o = 0; o < L.filters; ++o)
                for (int ch = 0; ch < C; ++ch)
                    for (int q = 0; q < hw; ++q) wperm[(size_t)o * L.cin + (size_t)q * C + ch] = w[(size_t)o * L.cin + (size_t)ch * hw + q];
            w = wperm.data();
        } else if (L.s2d7) {
            // 7x7 / stride 2 / pad 3 over 3 channels == 4x4 / stride 1 / pad 2 over the 2x2 space-to-depth image (32 = 4 positions x 8 padded
            // channels): input row 2*oy + kh - 3 = 2*(oy + a - 2) + dy  <=>  kh = 2a + dy - 1 (taps outside 0..6 get zero weights)
            tmp = L; tmp.size = 4; tmp.cin = 32; PL = &tmp;
            wperm.assign((size_t)L.filters * 32 * 16, 0.f);
            for (int o = 0; o < L.filters; ++o)
                for (int ch = 0; ch < 3; ++ch)
                    for (int a4 = 0; a4 < 4; ++a4) for (int dy = 0; dy < 2; ++dy) { const int kh = 2 * a4 + dy - 1; if (kh < 0 || kh > 6) continue;
                        for (int b4 = 0; b4 < 4; ++b4) for (int dx = 0; dx < 2; ++dx) { const int kw = 2 * b4 + dx - 1; if (kw < 0 || kw > 6) continue;
                            wperm[(((size_t)o * 32 + (dy * 2 + dx) * 8 + ch) * 4 + a4) * 4 + b4] = w[(((size_t)o * 3 + ch) * 7 + kh) * 7 + kw]; } }
            w = wperm.data();
        }
        pack_conv(*PL, params, w, L.in_dt, in_sc.empty() ? nullptr : in_sc.data(), wbuf, bias, osc, c->semantics, !c->pair_of(L.in[0]) ? 0 : L.in[0] < 0 ? 1 : 2);      // (filter rows against a tensor stored as pairs: W_hi | W_hi | W_lo for the image's three blocks, W_hi 32 | W_lo 32 per group for an interleaved layer output)
        HIPCK(c, hipMemcpy(L.d_w, wbuf.data(), wbuf.size(), hipMemcpyHostToDevice));
        HIPCK(c, hipMemcpy(L.d_b, bias.data(), bias.size() * 4, hipMemcpyHostToDevice));
        if (L.d_sc) HIPCK(c, hipMemcpy(L.d_sc, osc.data(), osc.size() * 4, hipMemcpyHostToDevice));
    }
    if (int rc = tail_fragments(c)) return rc;
    c->weights_loaded = true;
    return YOLO_OK;
}

int yolo_load_darknet_weights(yolo_ctx *c, const char *path, int header_ints)
{
    if (!c) return YOLO_ERR_INVALID;
    FILE *f = path ? fopen(path, "rb") : nullptr;
    if (!f) return fail(c, YOLO_ERR_IO, "cannot open weights file '%s'", path ? path : "(null)");
    int32_t ver[3];
    if (fread(ver, 4, 3, f) != 3) { fclose(f); return fail(c, YOLO_ERR_IO, "truncated header in '%s'", path); }
    if (header_ints == 0) header_ints = (ver[0] * 10 + ver[1]) >= 2 ? 5 : 4;      // DN/parser.c:1259-1265
    if (header_ints != 4 && header_ints != 5) { fclose(f); return fail(c, YOLO_ERR_INVALID, "header_ints must be 0, 4 or 5"); }
    fseek(f, 0, SEEK_END); long end = ftell(f); fseek(f, header_ints * 4, SEEK_SET);
    size_t n = (size_t)(end - header_ints * 4) / 4;
    if (n != c->weights_count) { fclose(f); return fail(c, YOLO_ERR_IO, "'%s' holds %zu floats after a %d-int header, topology needs %zu", path, n, header_ints, c->weights_count); }
    std::vector<float> flat(n);
    size_t got = fread(flat.data(), 4, n, f); fclose(f);
    if (got != n) return fail(c, YOLO_ERR_IO, "short read on '%s'", path);
    return yolo_set_weights(c, flat.data(), n);
}

// ---- export artifact: one self-describing file = cfg text + run configuration + the folded, packed,
//      device-ready parameters of every conv (+ fp8 scales, + the tile plan).  Counterpart of the reference's frozen
//      `.pb` (D2T/YOLO_V3_convert_darkenet_to_Tensorflow.py:99-104, D2T/object_detect.py:64-99): `input` in,
//      boxes / scores / classes out, nothing else needed to run. ----
namespace {
struct ArtHeader { char magic[8]; uint32_t version, dtype, semantics, decode, n_layers, num_cfgs, cfg_len, reserved; };
const char kArtMagic[8] = {'Y', 'O', 'L', 'O', 'H', 'I', 'P', '1'};
const uint32_t kArtVersion = 3;          // 3: split-fp16 filters packed for the interleaved pair layout (round 6); 2: filters packed chunk-major (conv_kchunk); 1: tap-major
uint64_t fnv1a(uint64_t h, const void *p, size_t n) { const uint8_t *b = (const uint8_t *)p; for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; } return h; }
struct ArtWriter {
    FILE *f; uint64_t h = 1469598103934665603ull; bool ok = true;
    void put(const void *p, size_t n) { if (ok && n && fwrite(p, 1, n, f) != n) ok = false; h = fnv1a(h, p, n); }
};
struct ArtReader {
    FILE *f; uint64_t h = 1469598103934665603ull; bool ok = true;
    void get(void *p, size_t n) { if (ok && n && fread(p, 1, n, f) != n) ok = false; if (ok) h = fnv1a(h, p, n); }
};
}  // namespace

int yolo_export(yolo_ctx *c, const char *path)
{
    if (!c) return YOLO_ERR_INVALID;
    if (!c->weights_loaded) return fail(c, YOLO_ERR_STATE, "yolo_export before weights were loaded");
    HIPCK(c, hipSetDevice(c->device)); HIPCK(c, hipStreamSynchronize(c->stream));
    FILE *f = path ? fopen(path, "wb") : nullptr;
    if (!f) return fail(c, YOLO_ERR_IO, "cannot create '%s'", path ? path : "(null)");
    ArtWriter w{f};
    const uint32_t NL = (uint32_t)c->layers.size();
    ArtHeader hd; memset(&hd, 0, sizeof hd); memcpy(hd.magic, kArtMagic, 8);
    hd.version = kArtVersion; hd.dtype = c->dtype; hd.semantics = c->semantics; hd.decode = c->decode; hd.n_layers = NL;
    hd.num_cfgs = (uint32_t)conv_num_cfgs(); hd.cfg_len = (uint32_t)c->cfg_text.size();
    w.put(&hd, sizeof hd); w.put(c->cfg_text.data(), c->cfg_text.size());
    std::vector<float> sc(c->user_scale); sc.resize(NL, 1.f); w.put(sc.data(), NL * 4);
    std::vector<int32_t> plan(NL); yolo_get_tile_configs(c, plan.data()); w.put(plan.data(), NL * 4);
    std::vector<uint8_t> buf;
    for (auto &L : c->layers) {
        if (L.type != L_CONV && L.type != L_LOCAL) continue;
        uint64_t sz[3] = {(uint64_t)L.cout_pad * L.kpad * dt_size(L.in_dt), (uint64_t)L.cout_pad, L.d_sc ? (uint64_t)L.cout_pad : 0};
        if (L.type == L_LOCAL) { sz[0] = (uint64_t)L.H * L.W * L.filters * L.size * L.size * L.cin * dt_size(L.in_dt); sz[1] = (uint64_t)L.H * L.W * L.filters; sz[2] = 0; }
        w.put(sz, sizeof sz);
        const void *src[3] = {L.d_w, L.d_b, L.d_sc}; const size_t bytes[3] = {(size_t)sz[0], (size_t)sz[1] * 4, (size_t)sz[2] * 4};
        for (int k = 0; k < 3; ++k) {
            if (!bytes[k]) continue;
            buf.resize(bytes[k]);
            if (hipMemcpy(buf.data(), src[k], bytes[k], hipMemcpyDeviceToHost) != hipSuccess) { fclose(f); return fail(c, YOLO_ERR_HIP, "export: device read failed"); }
            w.put(buf.data(), bytes[k]);
        }
    }
    const uint64_t sum = w.h;
    if (w.ok && fwrite(&sum, 1, 8, f) != 8) w.ok = false;
    if (fclose(f) != 0) w.ok = false;
    return w.ok ? YOLO_OK : fail(c, YOLO_ERR_IO, "short write on '%s'", path);
}

yolo_ctx *yolo_create_from_file(const char *path, int max_batch, int device, void *stream, int keep_layers, char *err, size_t err_len)
{
    auto bail = [&](yolo_ctx *c, const std::string &m) -> yolo_ctx * { if (err && err_len) snprintf(err, err_len, "%s", m.c_str()); if (c) yolo_destroy(c); return nullptr; };
    FILE *f = path ? fopen(path, "rb") : nullptr;
    if (!f) return bail(nullptr, std::string("cannot open '") + (path ? path : "(null)") + "'");
    ArtReader r{f};
    ArtHeader hd; r.get(&hd, sizeof hd);
    if (!r.ok || memcmp(hd.magic, kArtMagic, 8) != 0 || hd.version != kArtVersion || hd.cfg_len > (1u << 24) || hd.n_layers > 4096) { fclose(f); return bail(nullptr, "not a YOLOHIP1 artifact (or an unsupported version)"); }
    std::string cfg_text(hd.cfg_len, '\0'); r.get(&cfg_text[0], hd.cfg_len);
    std::vector<float> sc(hd.n_layers); r.get(sc.data(), (size_t)hd.n_layers * 4);
    std::vector<int32_t> plan(hd.n_layers); r.get(plan.data(), (size_t)hd.n_layers * 4);
    if (!r.ok) { fclose(f); return bail(nullptr, "truncated artifact"); }
    yolo_config cfg; memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = sizeof cfg; cfg.cfg_text = cfg_text.c_str(); cfg.max_batch = max_batch; cfg.dtype = (int)hd.dtype; cfg.semantics = (int)hd.semantics;
    cfg.decode = (int)hd.decode; cfg.device = device; cfg.keep_layers = keep_layers; cfg.stream = stream;
    yolo_ctx *c = yolo_create(&cfg, err, err_len);
    if (!c) { fclose(f); return nullptr; }
    if (c->layers.size() != hd.n_layers) { fclose(f); return bail(c, "artifact layer count does not match its own cfg"); }
    if (c->dtype == YOLO_FP8 && yolo_set_act_scales(c, sc.data(), (int)hd.n_layers) != YOLO_OK) { fclose(f); return bail(c, c->err); }
    std::vector<uint8_t> buf;
    for (auto &L : c->layers) {
        if (L.type != L_CONV && L.type != L_LOCAL) continue;
        uint64_t sz[3]; r.get(sz, sizeof sz);
        uint64_t want[3] = {(uint64_t)L.cout_pad * L.kpad * dt_size(L.in_dt), (uint64_t)L.cout_pad, L.d_sc ? (uint64_t)L.cout_pad : 0};
        if (L.type == L_LOCAL) { want[0] = (uint64_t)L.H * L.W * L.filters * L.size * L.size * L.cin * dt_size(L.in_dt); want[1] = (uint64_t)L.H * L.W * L.filters; want[2] = 0; }
        if (!r.ok || sz[0] != want[0] || sz[1] != want[1] || sz[2] != want[2]) { fclose(f); return bail(c, "artifact parameters do not fit the topology (truncated file or different packing)"); }
        void *dst[3] = {L.d_w, L.d_b, L.d_sc}; const size_t bytes[3] = {(size_t)sz[0], (size_t)sz[1] * 4, (size_t)sz[2] * 4};
        for (int k = 0; k < 3; ++k) {
            if (!bytes[k]) continue;
            buf.resize(bytes[k]); r.get(buf.data(), bytes[k]);
            if (!r.ok) { fclose(f); return bail(c, "truncated artifact"); }
            if (hipMemcpy(dst[k], buf.data(), bytes[k], hipMemcpyHostToDevice) != hipSuccess) { fclose(f); return bail(c, "artifact upload failed"); }
        }
    }
    uint64_t sum = 0; const bool got = fread(&sum, 1, 8, f) == 8; fclose(f);
    if (!got || sum != r.h) return bail(c, "artifact checksum mismatch");
    if (tail_fragments(c) != YOLO_OK) return bail(c, c->err);
    c->weights_loaded = true;
    // the tile plan is only meaningful for the tile table it was tuned with and for a plan that fuses nothing it cannot
    if (hd.num_cfgs == (uint32_t)conv_num_cfgs() && !keep_layers) { if (yolo_set_tile_configs(c, plan.data()) != YOLO_OK) { std::vector<int32_t> none(hd.n_layers, -1); yolo_set_tile_configs(c, none.data()); } }
    return c;
}

}  // extern "C"
