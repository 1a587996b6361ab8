// Host side of libyolo_hip.so: darknet-cfg planner, BN-fold + filter packing, launch sequence, C ABI.
//
// What the reference does with a Python graph builder + tf.Session (V3/yolo_v3.py:195-267,
// D2T/YOLO_V3_convert_darkenet_to_Tensorflow.py:435-545) or with parse_network_cfg + forward_network
// (DN/parser.c:730-875, DN/network.c:188-211) is done here once at yolo_create():
//   * shapes are inferred, every layer gets a view into a small pool of HBM buffers (liveness-based
//     reuse so consecutive layers recycle the same few allocations and stay in L2 / Infinity Cache);
//   * route/concat is never executed: producers are planned to write straight into a channel window of
//     the concat buffer (DN/route_layer.c:74-89 and tf.concat V3/yolo_v3.py:247,259 become strides);
//   * `shortcut` after a conv is folded into that conv's epilogue (V3/yolo_v3.py:54-60);
//   * batch-norm is folded into the filters at weight-load time.
#include "../../include/yolo_hip.h"
#include "kernels.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace {

enum LType { L_CONV, L_SHORTCUT, L_ROUTE, L_UPSAMPLE, L_MAXPOOL, L_REORG, L_YOLO, L_REGION, L_DETECT, L_LOCAL };

struct Section { std::string type; std::map<std::string, std::string> kv; };

struct Layer {
    LType type;
    int H = 0, W = 0, C = 0;             // logical output geometry
    std::vector<int> in;                 // producer layer indices (-1 = network input)
    // conv
    int filters = 0, size = 0, stride = 1, pad = 0, bn = 0, act = ACT_LINEAR;
    int cin = 0, cin_pad = 0, kpad = 0, cout_pad = 0;
    void *d_w = nullptr; float *d_b = nullptr; float *d_sc = nullptr;   // filters, bias, fp8 per-channel dequant scale
    void *d_wf = nullptr;                   // bf16 1x1 conv that can ride in its producer's epilogue: its filters in MFMA-fragment order (tail_fragments)
    int in_dt = DT_BF16;                 // operand type of this conv's MFMA (filters are stored in it)
    int store_dt = DT_BF16;              // element type of this layer's output tensor (mixed plans: an fp8 network with bf16 islands, cfg key yolo_store)
    int tile_cfg = -1;
    int residual_from = -2;              // >= -1: fused shortcut source
    bool head = false;                   // conv feeding a yolo/region layer: fp32 output
    float *d_obj = nullptr;              // ... feeding a [yolo] layer (bf16 / fp8 networks): compact plane of its objectness logits [max_batch * H * W][anchors]
    bool stem_skip = false, stem = false;   // fused stem (conv_stem.hip): layer 0 is never materialised, layer 1 launches both
    bool blk_skip = false, blk = false;     // fused residual block (conv_block.hip): this 1x1 conv is computed inside the launch of the 3x3 conv that follows / this 3x3 conv launches both
    bool stem_tail = false;                 // ... and this 1x1 conv (layer 2) is computed by that launch too
    bool halo = false;                      // 3x3/s1, 32 -> 64 channels: halo-staged kernel instead of the tiled one
    // [connected] (YOLOv1's fully connected head, V1/YOLO_V1_Inference.py:196-206; DN/connected_layer.c:151): a 1x1 conv over the
    // producer's tensor flattened to one pixel per image; fc_h/w/c = the producer's geometry (darknet / the TF graph flatten CHW)
    bool fc = false; int fc_h = 0, fc_w = 0, fc_c = 0;
    // 7x7 / stride 2 / pad 3 first conv (YOLOv1): computed as a 4x4 / stride 1 conv over the 2x2 space-to-depth of the input
    bool s2d7 = false;
    int side = 0, sqr = 0;                  // [detection] head
    // fused 1x1 tail of the tiled conv kernel: `tail_layer` (on the producer) = index of the 1x1 conv that can be computed
    // in the producer's epilogue, `fused_into` (on that 1x1) = the producer; `tail_on` = the plan uses it
    int tail_layer = -1, fused_into = -1; bool tail_on = false;
    // shortcut/route bookkeeping
    bool noop = false;                   // output is an alias / was produced by someone else
    std::vector<int> copy_inputs;        // route inputs that must be copied (could not be placed)
    std::vector<int> copy_offsets;
    // pool / upsample / reorg
    int psize = 0, pstride = 0, ppad = 0;
    // head
    int na = 0, classes = 0, row_off = 0;
    std::vector<float> anchors;          // masked, in reference units
    // storage
    int storage = -1; int ch_off = 0;    // view = storage buffer + channel offset
    TView out;
};

struct Storage { int def = 1 << 30, last = -1; size_t bytes = 0; int phys = -1; int stride = 0; int dt = DT_BF16; bool persistent = false; };

}  // namespace

struct yolo_ctx {
    std::string err, cfg_text;
    int device = 0;
    hipStream_t stream = nullptr; bool own_stream = false;
    int max_batch = 1, dtype = YOLO_BF16, semantics = YOLO_SEM_TF, decode = YOLO_DECODE_RATIO, keep_layers = 0;
    int in_h = 0, in_w = 0, in_c = 0;
    std::vector<Layer> layers;
    std::vector<Storage> storages;
    std::vector<void *> phys; std::vector<size_t> phys_bytes;
    TView input;                          // [n, S, S, 8]
    void *d_zeros = nullptr;
    void *d_stage = nullptr; size_t stage_bytes = 0;     // host->device image staging
    TView s2d;                            // [n, S/2, S/2, 32]: space-to-depth of the input for a 7x7/2 first conv
    const uint8_t *stem_u8 = nullptr; float stem_scale = 1.f;      // uint8 image the fused stem reads itself (no conversion launch), or nullptr: c->input
    float in_mul = 1.f, in_add = 0.f;     // input normalisation after the /255: v * in_mul + in_add ([net] yolo_input_mul / yolo_input_add)
    float *d_det = nullptr; int rows = 0, attrs = 0;
    // lean detect path (yolo_detect*): the decode writes scores, labels and the four box numbers of every row, not the tensor
    bool lean_cnt_dirty = false;
    void *d_lean_list = nullptr; unsigned *d_lean_cnt = nullptr;      // lean decode: list of the boxes that pass the objectness pre-filter + its counters
    float *d_box4 = nullptr; bool lean = false, det_valid = false, lean_ok = false; float lean_thr = 0.f; int lean_heads = 0;      // lean_heads: [yolo] heads when all can share one decode launch, else 0
    // postprocess workspace
    float *d_scores = nullptr; int *d_labels = nullptr; int *d_cand = nullptr; unsigned long long *d_keys = nullptr;
    float4 *d_sbox = nullptr; int *d_slabel = nullptr; float *d_sscore = nullptr; int rows_pow2 = 0;
    void *d_boxes = nullptr; int *d_counts = nullptr; int boxes_cap = 0;
    int *d_srow = nullptr, *d_rows = nullptr;      // rows_out support: row of every sorted candidate [max_batch * rows], staging [boxes_cap]
    // darknet-flavoured outputs (yolo_darknet_boxes / yolo_last_layer_output): records, row list, count, last layer's planar output
    float *d_dn_rec = nullptr; int *d_dn_src = nullptr; int *d_dn_count = nullptr; float *d_dn_last = nullptr; size_t dn_last_cap = 0;
    // yolo_detect_graph state
    struct GKey { const void *img; int n, fmt; float scale, st, it; int mo, nm, sm; void *bo, *co; } gkey{};
    hipGraphExec_t gexec = nullptr; int gstate = 0;      // 0: next call eager, 1: next call captures, 2: replay, -1: capture unsupported
    bool weights_loaded = false;
    int scores_mode = -1;                 // what d_scores/d_labels hold: 0 max(obj*cls) from the decode, 1 objectness, -1 nothing
    size_t weights_count = 0;
    double conv_flops = 0;
    int last_n = 0;
    // fp8 scheme: stored value = e4m3(real / scale).  user_scale[i] is what yolo_set_act_scales gave for
    // layer i (1 by default); eff_scale[i] is the scale of the tensor layer i's view holds (inherited through
    // upsample / maxpool / reorg / single-input route; NaN for multi-input routes, which are per channel).
    std::vector<float> user_scale, eff_scale;
    int act_dt() const { return dtype == YOLO_FP32 ? DT_F32 : dtype == YOLO_FP8 ? DT_FP8 : dtype == YOLO_FP16 ? DT_F16 : DT_BF16; }
    bool half_like() const { return dtype == YOLO_BF16 || dtype == YOLO_FP16; }      // 16-bit storage: the same kernels, the same plan
    int gran() const { return dtype == YOLO_FP8 ? 16 : 8; }            // channel granule = one 16-B piece (8 for fp32 too)
    size_t esize() const { return dt_size(act_dt()); }
};

namespace {

int fail(yolo_ctx *c, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (c) c->err = buf;
    return code;
}
#define HIPCK(c, expr)                                                                       \
    do { hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) return fail(c, YOLO_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); } while (0)

inline int roundup(int x, int m) { return (x + m - 1) / m * m; }
inline int gran_of(int dt) { return dt == DT_FP8 ? 16 : 8; }      // channels per 16-byte piece (8 for fp32 tensors too)

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

// ---- cfg parsing (DN/parser.c:730-875 read_cfg + option_find_*) ------------------------------
bool parse_cfg(const char *text, std::vector<Section> &out, std::string &err)
{
    std::string s(text ? text : "");
    size_t pos = 0;
    while (pos <= s.size()) {
        size_t e = s.find('\n', pos); if (e == std::string::npos) e = s.size();
        std::string line = s.substr(pos, e - pos); pos = e + 1;
        size_t a = line.find_first_not_of(" \t\r"); if (a == std::string::npos) continue;
        size_t b = line.find_last_not_of(" \t\r"); line = line.substr(a, b - a + 1);
        if (line[0] == '#' || line[0] == ';') continue;
        if (line[0] == '[') {
            size_t r = line.find(']'); if (r == std::string::npos) { err = "cfg: unterminated section " + line; return false; }
            Section sec; sec.type = line.substr(1, r - 1); out.push_back(sec);
        } else {
            size_t eq = line.find('='); if (eq == std::string::npos || out.empty()) { err = "cfg: bad line '" + line + "'"; return false; }
            std::string k = line.substr(0, eq), v = line.substr(eq + 1);
            auto trim = [](std::string &t) { size_t x = t.find_first_not_of(" \t"); size_t y = t.find_last_not_of(" \t"); t = x == std::string::npos ? "" : t.substr(x, y - x + 1); };
            trim(k); trim(v); out.back().kv[k] = v;
        }
    }
    if (out.empty() || (out[0].type != "net" && out[0].type != "network")) { err = "cfg: first section must be [net]"; return false; }
    return true;
}
int opt_i(const Section &s, const char *k, int d) { auto it = s.kv.find(k); return it == s.kv.end() ? d : atoi(it->second.c_str()); }
std::string opt_s(const Section &s, const char *k, const char *d) { auto it = s.kv.find(k); return it == s.kv.end() ? std::string(d) : it->second; }
std::vector<float> opt_list(const Section &s, const char *k)
{
    std::vector<float> v; auto it = s.kv.find(k); if (it == s.kv.end()) return v;
    const char *p = it->second.c_str();
    while (*p) { char *e; double d = strtod(p, &e); if (e == p) break; v.push_back((float)d); p = e; while (*p == ',' || *p == ' ') ++p; }
    return v;
}

TView view_of(const yolo_ctx *c, int idx) { return idx < 0 ? c->input : c->layers[idx].out; }

int build_plan(yolo_ctx *c, const std::vector<Section> &secs)
{
    const Section &net = secs[0];
    c->in_h = opt_i(net, "height", 0); c->in_w = opt_i(net, "width", 0); c->in_c = opt_i(net, "channels", 3);
    if (c->in_h <= 0 || c->in_w != c->in_h || c->in_c != 3)
        return fail(c, YOLO_ERR_UNSUPPORTED, "cfg: need square input with 3 channels (got %dx%dx%d)", c->in_w, c->in_h, c->in_c);
    const int NL = (int)secs.size() - 1;
    c->layers.resize(NL);
    int H = c->in_h, W = c->in_w, C = c->in_c;
    auto dims = [&](int idx, int &h, int &w, int &ch) { if (idx < 0) { h = c->in_h; w = c->in_w; ch = c->in_c; } else { h = c->layers[idx].H; w = c->layers[idx].W; ch = c->layers[idx].C; } };
    c->rows = 0; c->attrs = 0; c->conv_flops = 0; c->weights_count = 0;
    for (int i = 0; i < NL; ++i) {
        const Section &s = secs[i + 1]; Layer &L = c->layers[i];
        L.in = {i - 1};
        if (s.type == "convolutional") {
            L.type = L_CONV; L.filters = opt_i(s, "filters", 1); L.size = opt_i(s, "size", 1); L.stride = opt_i(s, "stride", 1);
            L.pad = opt_i(s, "pad", 0) ? L.size / 2 : opt_i(s, "padding", 0);
            L.bn = opt_i(s, "batch_normalize", 0);
            std::string act = opt_s(s, "activation", "logistic");
            if (act == "leaky") L.act = ACT_LEAKY; else if (act == "linear") L.act = ACT_LINEAR;
            else return fail(c, YOLO_ERR_UNSUPPORTED, "layer %d: activation '%s' unsupported", i, act.c_str());
            if (L.size == 7 && L.stride == 2 && L.pad == 3 && i == 0 && C == 3 && H % 2 == 0 && W % 2 == 0 && c->dtype != YOLO_FP8) L.s2d7 = true;
            else if (L.size != 1 && L.size != 3) return fail(c, YOLO_ERR_UNSUPPORTED, "layer %d: conv size %d unsupported on the device path", i, L.size);
            // fp8 mode: the first conv still reads the bf16 image (3 real channels padded to 8) with bf16 filters; a conv reads its
            // producer's tensor in the type that tensor is stored in (cfg key `yolo_store=bf16` on a [convolutional] section of an fp8
            // network keeps that layer's output -- and what is derived from it without arithmetic -- in bf16: mixed-precision plans)
            L.in_dt = c->dtype == YOLO_FP8 ? (i == 0 ? DT_BF16 : c->layers[i - 1].store_dt) : c->act_dt();
            L.store_dt = c->act_dt();
            {
                const std::string st = opt_s(s, "yolo_store", "");
                if (!st.empty()) {
                    if (c->dtype != YOLO_FP8) return fail(c, YOLO_ERR_UNSUPPORTED, "layer %d: yolo_store is a key of fp8 networks (mixed e4m3 / bf16 plans)", i);
                    if (st == "bf16") L.store_dt = DT_BF16; else if (st == "fp8") L.store_dt = DT_FP8;
                    else return fail(c, YOLO_ERR_INVALID, "layer %d: yolo_store=%s (bf16 or fp8)", i, st.c_str());
                }
            }
            L.cin = C; L.cin_pad = roundup(C, L.in_dt == DT_FP8 ? 16 : 8);
            L.kpad = roundup(L.size * L.size * L.cin_pad, L.in_dt == DT_FP8 ? 128 : 64); L.cout_pad = roundup(L.filters, 256);
            if (L.s2d7) { L.cin_pad = 32; L.kpad = 16 * 32; }          // 4x4 taps x (2x2 positions x 8 padded channels)
            H = (H + 2 * L.pad - L.size) / L.stride + 1; W = (W + 2 * L.pad - L.size) / L.stride + 1; C = L.filters;
            c->conv_flops += 2.0 * L.size * L.size * L.cin * L.filters * (double)H * W;
            c->weights_count += (size_t)L.filters * (L.bn ? 4 : 1) + (size_t)L.filters * L.cin * L.size * L.size;
        } else if (s.type == "connected") {
            if (c->dtype == YOLO_FP8) return fail(c, YOLO_ERR_UNSUPPORTED, "layer %d: [connected] is not served in the fp8 configuration", i);
            if (opt_i(s, "batch_normalize", 0)) return fail(c, YOLO_ERR_UNSUPPORTED, "layer %d: batch-normalised [connected]", i);
            L.type = L_CONV; L.fc = true; L.fc_h = H; L.fc_w = W; L.fc_c = C;
            L.filters = opt_i(s, "output", 1); L.size = 1; L.stride = 1; L.pad = 0; L.bn = 0;
            std::string act = opt_s(s, "activation", "logistic");
            if (act == "leaky") L.act = ACT_LEAKY; else if (act == "linear") L.act = ACT_LINEAR;
            else return fail(c, YOLO_ERR_UNSUPPORTED, "layer %d: activation '%s' unsupported", i, act.c_str());
            L.in_dt = c->act_dt(); L.store_dt = c->act_dt();
            if ((long)H * W * C > (1L << 24)) return fail(c, YOLO_ERR_UNSUPPORTED, "layer %d: [connected] input too large", i);
            L.cin = H * W * C; L.cin_pad = roundup(L.cin, 8); L.kpad = roundup(L.cin_pad, 64); L.cout_pad = roundup(L.filters, 256);
            c->conv_flops += 2.0 * L.cin * L.filters;
            c->weights_count += (size_t)L.filters + (size_t)L.filters * L.cin;
            H = 1; W = 1; C = L.filters;
        } else if (s.type == "local") {
            // locally connected (DN/local_layer.c; darknet's own yolov1.cfg): `pad` is a flag AND the im2col pad amount (:10-24, :103)
            if (c->dtype == YOLO_FP8) return fail(c, YOLO_ERR_UNSUPPORTED, "layer %d: [local] is not served in the fp8 configuration", i);
            L.type = L_LOCAL; L.filters = opt_i(s, "filters", 1); L.size = opt_i(s, "size", 1); L.stride = opt_i(s, "stride", 1); L.pad = opt_i(s, "pad", 0);
            if (L.pad != 0 && L.pad != 1) return fail(c, YOLO_ERR_UNSUPPORTED, "layer %d: [local] pad must be 0 or 1", i);
            std::string act = opt_s(s, "activation", "logistic");
            if (act == "leaky") L.act = ACT_LEAKY; else if (act == "linear") L.act = ACT_LINEAR;
            else return fail(c, YOLO_ERR_UNSUPPORTED, "layer %d: activation '%s' unsupported", i, act.c_str());
            if (C % 8) return fail(c, YOLO_ERR_UNSUPPORTED, "layer %d: [local] needs a producer with a multiple of 8 channels", i);
            L.cin = C; L.cin_pad = C; L.in_dt = c->act_dt(); L.store_dt = c->act_dt();
            const int ho = ((L.pad ? H - 1 : H - L.size)) / L.stride + 1, wo = ((L.pad ? W - 1 : W - L.size)) / L.stride + 1;
            if (ho < 1 || wo < 1) return fail(c, YOLO_ERR_INVALID, "layer %d: [local] larger than its input", i);
            H = ho; W = wo; C = L.filters;
            c->conv_flops += 2.0 * L.size * L.size * L.cin * L.filters * (double)H * W;
            c->weights_count += (size_t)L.filters * H * W + (size_t)H * W * L.filters * L.cin * L.size * L.size;
        } else if (s.type == "dropout") {
            L.type = L_ROUTE;                       // inference: identity (DN/dropout_layer.c:38-40)
        } else if (s.type == "detection") {
            L.type = L_DETECT; L.classes = opt_i(s, "classes", 1); L.na = opt_i(s, "num", 1); L.side = opt_i(s, "side", 7); L.sqr = opt_i(s, "sqrt", 0);
            if (opt_i(s, "coords", 4) != 4 || opt_i(s, "softmax", 0)) return fail(c, YOLO_ERR_UNSUPPORTED, "layer %d: [detection] with coords != 4 or softmax", i);
            if (i == 0 || !c->layers[i - 1].fc) return fail(c, YOLO_ERR_UNSUPPORTED, "layer %d: [detection] must follow a [connected] layer", i);
            if (C != L.side * L.side * (L.classes + L.na * 5)) return fail(c, YOLO_ERR_INVALID, "layer %d: [detection] expects %d inputs, got %d", i, L.side * L.side * (L.classes + L.na * 5), C);
            if (c->attrs && c->attrs != 5 + L.classes) return fail(c, YOLO_ERR_UNSUPPORTED, "heads with different class counts");
            c->attrs = 5 + L.classes; L.row_off = c->rows; c->rows += L.side * L.side * L.na;
            c->layers[i - 1].head = true;
            H = L.side; W = L.side;
        } else if (s.type == "shortcut") {
            L.type = L_SHORTCUT; int f = opt_i(s, "from", -1); f = f < 0 ? i + f : f;
            if (f < 0 || f >= i) return fail(c, YOLO_ERR_INVALID, "layer %d: bad shortcut from", i);
            L.in = {i - 1, f};
            if (opt_s(s, "activation", "linear") != "linear") return fail(c, YOLO_ERR_UNSUPPORTED, "layer %d: shortcut activation", i);
            int h2, w2, c2; dims(f, h2, w2, c2);
            if (h2 != H || w2 != W || c2 != C) return fail(c, YOLO_ERR_UNSUPPORTED, "layer %d: shortcut shape mismatch", i);
        } else if (s.type == "route") {
            L.type = L_ROUTE; L.in.clear();
            std::vector<float> ls = opt_list(s, "layers");
            if (ls.empty()) return fail(c, YOLO_ERR_INVALID, "layer %d: route without layers", i);
            C = 0;
            for (float v : ls) {
                int l = (int)v; l = l < 0 ? i + l : l;
                if (l < 0 || l >= i) return fail(c, YOLO_ERR_INVALID, "layer %d: bad route index", i);
                L.in.push_back(l);
                int h2, w2, c2; dims(l, h2, w2, c2);
                if (L.in.size() == 1) { H = h2; W = w2; } else if (h2 != H || w2 != W) return fail(c, YOLO_ERR_INVALID, "layer %d: route spatial mismatch", i);
                C += c2;
            }
        } else if (s.type == "upsample") {
            L.type = L_UPSAMPLE; L.pstride = opt_i(s, "stride", 2);
            if (L.pstride != 2) return fail(c, YOLO_ERR_UNSUPPORTED, "layer %d: upsample stride %d", i, L.pstride);
            H *= 2; W *= 2;
        } else if (s.type == "maxpool") {
            L.type = L_MAXPOOL; L.pstride = opt_i(s, "stride", 1); L.psize = opt_i(s, "size", L.pstride);
            L.ppad = opt_i(s, "padding", (L.psize - 1) / 2);
            H = (H + 2 * L.ppad) / L.pstride; W = (W + 2 * L.ppad) / L.pstride;
        } else if (s.type == "reorg") {
            L.type = L_REORG; L.pstride = opt_i(s, "stride", 1);
            if (H % L.pstride || W % L.pstride) return fail(c, YOLO_ERR_INVALID, "layer %d: reorg stride", i);
            H /= L.pstride; W /= L.pstride; C *= L.pstride * L.pstride;
        } else if (s.type == "yolo" || s.type == "region") {
            L.type = s.type == "yolo" ? L_YOLO : L_REGION;
            L.classes = opt_i(s, "classes", 20);
            std::vector<float> an = opt_list(s, "anchors"), mask = opt_list(s, "mask");
            int total = opt_i(s, "num", 1);
            if ((int)an.size() < 2 * total) return fail(c, YOLO_ERR_INVALID, "layer %d: anchors/num mismatch", i);
            if (L.type == L_YOLO && !mask.empty()) { for (float m : mask) { int k = (int)m; if (k < 0 || k >= total) return fail(c, YOLO_ERR_INVALID, "layer %d: mask", i); L.anchors.push_back(an[2 * k]); L.anchors.push_back(an[2 * k + 1]); } }
            else L.anchors.assign(an.begin(), an.begin() + 2 * total);
            L.na = (int)L.anchors.size() / 2;
            if (L.na > 16) return fail(c, YOLO_ERR_UNSUPPORTED, "layer %d: more than 16 anchors", i);
            if (i == 0 || c->layers[i - 1].type != L_CONV) return fail(c, YOLO_ERR_UNSUPPORTED, "layer %d: head must follow a conv", i);
            if (C != L.na * (5 + L.classes)) return fail(c, YOLO_ERR_INVALID, "layer %d: head expects %d channels, conv gives %d", i, L.na * (5 + L.classes), C);
            if (H != W) return fail(c, YOLO_ERR_UNSUPPORTED, "layer %d: non-square grid", i);
            if (c->attrs && c->attrs != 5 + L.classes) return fail(c, YOLO_ERR_UNSUPPORTED, "heads with different class counts");
            c->attrs = 5 + L.classes; L.row_off = c->rows; c->rows += H * W * L.na;
            c->layers[i - 1].head = true;
        } else {
            return fail(c, YOLO_ERR_UNSUPPORTED, "layer %d: section [%s] is outside the inference hot path", i, s.type.c_str());
        }
        L.H = H; L.W = W; L.C = C;
        if (L.type != L_CONV && L.type != L_LOCAL) {
            // layers that move data keep the type of what they move; their operands must agree
            int dt = -1;
            for (int j : L.in) { const int dj = j < 0 ? c->act_dt() : c->layers[j].store_dt; if (dt >= 0 && dj != dt && (L.type == L_ROUTE || L.type == L_SHORTCUT)) return fail(c, YOLO_ERR_INVALID, "layer %d: operands stored in different types (yolo_store): a %s needs one type", i, L.type == L_ROUTE ? "route" : "shortcut"); if (dt < 0) dt = dj; }
            L.store_dt = dt >= 0 ? dt : c->act_dt();
        }
    }
    if (c->rows == 0) return fail(c, YOLO_ERR_INVALID, "cfg has no [yolo] / [region] / [detection] head");
    c->in_mul = (float)atof(opt_s(net, "yolo_input_mul", "1").c_str()); c->in_add = (float)atof(opt_s(net, "yolo_input_add", "0").c_str());
    if (c->rows > 32768) return fail(c, YOLO_ERR_UNSUPPORTED, "more than 32768 candidates per image");

    // ---- use counts, shortcut fusion, concat placement ----
    std::vector<int> uses(NL, 0);
    for (int i = 0; i < NL; ++i) for (int j : c->layers[i].in) if (j >= 0) uses[j]++;
    for (int i = 0; i < NL; ++i) {
        Layer &L = c->layers[i];
        if (L.type == L_SHORTCUT && !c->keep_layers) {
            Layer &P = c->layers[i - 1];
            if (P.type == L_CONV && uses[i - 1] == 1 && !P.head && L.in[1] != i - 1 && (L.in[1] < 0 || c->layers[L.in[1]].store_dt == P.store_dt)) { P.residual_from = L.in[1]; L.noop = true; }
        }
    }
    // fused stem: conv0 (3x3/s1, 3 -> 32) read only by conv1 (3x3/s2, 32 -> 64), bf16, nothing asking for layer 0's tensor
    // (the stem and halo kernels address their input with 32-bit buffer offsets: the whole-batch window must stay under 2 GiB)
    // (an e4m3 network whose first layers are stored in a 16-bit type -- a mixed plan, yolo_store=bf16 -- runs them through the same fused
    //  kernels: what counts is the type of the tensors a kernel touches, not the context's)
    const bool ctx16 = c->half_like() || c->dtype == YOLO_FP8;
    auto is16 = [](int dt) { return dt == DT_BF16 || dt == DT_F16; };
    if (ctx16 && !c->keep_layers && NL >= 2 && !getenv("YOLO_NO_STEM") && (double)c->max_batch * c->in_h * c->in_w * 8 * 2 < 2147483648.0) {
        const Layer &A = c->layers[0], &B = c->layers[1];
        if (A.type == L_CONV && B.type == L_CONV && uses[0] == 1 && B.in[0] == 0 && A.size == 3 && A.stride == 1 && A.pad == 1 && A.cin == 3 &&
            A.filters == 32 && B.size == 3 && B.stride == 2 && B.pad == 1 && B.filters == 64 && !A.head && !B.head && B.residual_from < -1 &&
            is16(A.in_dt) && A.store_dt == A.in_dt && B.in_dt == A.in_dt && B.store_dt == A.in_dt) {        // (layer 0 reads the staged image, which is kept in its operand type)
            c->layers[0].stem_skip = true; c->layers[1].stem = true;
            if (NL >= 3) {
                const Layer &T = c->layers[2];
                if (T.type == L_CONV && !T.fc && T.in[0] == 1 && T.size == 1 && T.stride == 1 && T.pad == 0 && T.filters == 32 && !T.head && T.residual_from < -1 &&
                    T.in_dt == A.in_dt && T.store_dt == A.in_dt)
                    c->layers[2].stem_tail = true;
            }
        }
    }
    if (ctx16 && !getenv("YOLO_NO_HALO"))
        for (int i = 1; i < NL; ++i) {
            Layer &L = c->layers[i];
            if (L.type == L_CONV && !L.head && !L.stem && !L.stem_skip && !L.stem_tail && L.size == 3 && L.stride == 1 && L.pad == 1 && L.cin == 32 && L.filters == 64 &&
                is16(L.in_dt) && L.store_dt == L.in_dt && (L.residual_from < 0 || c->layers[L.residual_from].store_dt == L.in_dt))
                L.halo = true;
        }
    // fused residual block (conv_block.hip): a 1x1 conv 128 -> 64 read only by the 3x3 conv 64 -> 128 that follows, whose folded shortcut
    // source is the 1x1's own input, on a grid that is whole 13 x 13 blocks (darknet-53's 104 x 104 stage at 416 x 416)
    if (ctx16 && !c->keep_layers && !getenv("YOLO_NO_RESBLOCK"))
        for (int i = 1; i + 1 < NL; ++i) {
            Layer &A = c->layers[i], &B = c->layers[i + 1];
            if (A.type == L_CONV && B.type == L_CONV && !A.fc && !B.fc && !A.head && !B.head && uses[i] == 1 && B.in[0] == i && A.in[0] >= 0 &&
                A.size == 1 && A.stride == 1 && A.pad == 0 && A.cin == 128 && A.filters == 64 && A.residual_from < -1 &&
                B.size == 3 && B.stride == 1 && B.pad == 1 && B.cin == 64 && B.filters == 128 && B.residual_from == A.in[0] &&
                B.H % 13 == 0 && B.W % 13 == 0 && A.in_dt == B.in_dt && (A.in_dt == DT_BF16 || A.in_dt == DT_F16) && A.store_dt == A.in_dt && B.store_dt == A.in_dt) { A.blk_skip = true; B.blk = true; }
        }
    // 1x1 convs that can ride in their producer's epilogue: conv i (bf16, 128 or 256 output channels, optionally with its
    // fused shortcut) read by a 1x1/s1 conv with half as many filters
    if ((c->half_like() || c->dtype == YOLO_FP8) && !c->keep_layers && !getenv("YOLO_NO_TAIL")) {
        for (int i = 0; i + 1 < NL; ++i) {
            Layer &P = c->layers[i];
            if (P.type != L_CONV || P.fc || P.head || P.stem || P.stem_skip || P.stem_tail || P.blk || (P.filters != 128 && P.filters != 256)) continue;
            int o = i;
            if (P.residual_from >= -1) o = i + 1;            // its shortcut was folded into it: consumers read layer i+1
            const int j = o + 1;
            if (j >= NL) continue;
            Layer &T = c->layers[j];
            if (T.type == L_CONV && !T.fc && T.in[0] == o && T.size == 1 && T.stride == 1 && T.pad == 0 && T.filters * 2 == P.filters && !T.head &&
                T.residual_from < -1 && !T.stem_tail && !T.blk_skip && T.in_dt == P.in_dt) { P.tail_layer = j; T.fused_into = i; }      // (same operand type: the tail runs on the producer's MFMA)
        }
    }
    // storage assignment: st_of[i] = storage holding layer i's output
    std::vector<int> place_route(NL, -1), place_off(NL, 0);
    for (int i = 0; i < NL; ++i) {
        Layer &L = c->layers[i];
        if (L.type != L_ROUTE || L.in.size() < 2) continue;
        int off = 0;
        for (int j : L.in) {
            int cj = j < 0 ? c->in_c : c->layers[j].C;
            bool ok = j >= 0 && place_route[j] < 0 && c->layers[j].type != L_ROUTE && !c->layers[j].head &&
                      c->layers[j].type != L_YOLO && c->layers[j].type != L_REGION && c->layers[j].type != L_DETECT && (cj % gran_of(L.store_dt) == 0) && (off % gran_of(L.store_dt) == 0);
            // a fused-away conv's real producer is the conv; the shortcut layer itself is what gets placed
            if (ok && c->layers[j].type == L_CONV && j + 1 < NL && c->layers[j + 1].noop && c->layers[j + 1].type == L_SHORTCUT) ok = false;
            if (ok) { place_route[j] = i; place_off[j] = off; }
            else { L.copy_inputs.push_back(j); L.copy_offsets.push_back(off); }
            off += cj;
        }
    }
    auto new_storage = [&](int stride, int dt, size_t pixels, bool persistent) {
        Storage s; s.stride = stride; s.dt = dt; s.bytes = pixels * (size_t)stride * dt_size(dt); s.persistent = persistent;
        c->storages.push_back(s); return (int)c->storages.size() - 1;
    };
    // routes first (so producers can point into them)
    for (int i = 0; i < NL; ++i) {
        Layer &L = c->layers[i];
        if (L.type == L_ROUTE && L.in.size() >= 2) {
            L.storage = new_storage(roundup(L.C, gran_of(L.store_dt)), L.store_dt, (size_t)c->max_batch * L.H * L.W, c->keep_layers); L.ch_off = 0;
        }
    }
    for (int i = 0; i < NL; ++i) {
        Layer &L = c->layers[i];
        if (L.type == L_YOLO || L.type == L_REGION || L.type == L_DETECT) { L.noop = true; L.storage = c->layers[i - 1].storage; L.ch_off = c->layers[i - 1].ch_off; continue; }
        if (L.type == L_ROUTE && L.in.size() == 1) { L.noop = true; int j = L.in[0]; if (j < 0) return fail(c, YOLO_ERR_UNSUPPORTED, "route to network input"); L.storage = c->layers[j].storage; L.ch_off = c->layers[j].ch_off; continue; }
        if (L.type == L_ROUTE) continue;
        if (L.stem_skip) { L.noop = true; continue; }               // lives in LDS only
        if (place_route[i] >= 0) { L.storage = c->layers[place_route[i]].storage; L.ch_off = place_off[i]; }
        else if (L.head) L.storage = new_storage(roundup(L.C, 4), DT_F32, (size_t)c->max_batch * L.H * L.W, true);
        else L.storage = new_storage(roundup(L.C, gran_of(L.store_dt)), L.store_dt, (size_t)c->max_batch * L.H * L.W, c->keep_layers);
    }
    // a conv whose shortcut was fused writes the shortcut layer's tensor
    for (int i = 0; i + 1 < NL; ++i) {
        Layer &L = c->layers[i];
        if (L.type == L_CONV && L.residual_from >= -1) {
            // its own storage slot is unused: redirect to the shortcut's
            Storage &mine = c->storages[L.storage];
            if (place_route[i] < 0) mine.bytes = 0;
            L.storage = c->layers[i + 1].storage; L.ch_off = c->layers[i + 1].ch_off;
        }
    }
    // liveness: def = first writer, last = last reader of any view
    for (int i = 0; i < NL; ++i) {
        Layer &L = c->layers[i];
        if (L.storage < 0) continue;
        if (!L.noop) { Storage &s = c->storages[L.storage]; s.def = std::min(s.def, L.fused_into >= 0 ? L.fused_into : i); s.last = std::max(s.last, i); }
        for (int j : L.in) if (j >= 0 && c->layers[j].storage >= 0) { Storage &s = c->storages[c->layers[j].storage]; s.last = std::max(s.last, i); }
        if (L.type == L_CONV && L.residual_from >= 0) { Storage &s = c->storages[c->layers[L.residual_from].storage]; s.last = std::max(s.last, i); }
    }
    // greedy pooled assignment
    std::vector<int> free_list;
    for (int i = 0; i < NL; ++i) {
        for (size_t k = 0; k < c->storages.size(); ++k) {
            Storage &s = c->storages[k];
            if (s.def != i || s.bytes == 0) continue;
            int pick = -1;
            if (!s.persistent) {
                for (size_t f = 0; f < free_list.size(); ++f)
                    if (pick < 0 || c->phys_bytes[free_list[f]] > c->phys_bytes[free_list[pick]]) pick = (int)f;
            }
            if (pick >= 0) { s.phys = free_list[pick]; free_list.erase(free_list.begin() + pick); c->phys_bytes[s.phys] = std::max(c->phys_bytes[s.phys], s.bytes); }
            else { s.phys = (int)c->phys_bytes.size(); c->phys_bytes.push_back(s.bytes); }
        }
        for (size_t k = 0; k < c->storages.size(); ++k) {
            Storage &s = c->storages[k];
            if (s.last == i && s.phys >= 0 && !s.persistent) free_list.push_back(s.phys);
        }
    }
    return YOLO_OK;
}

int allocate(yolo_ctx *c)
{
    c->phys.assign(c->phys_bytes.size(), nullptr);
    for (size_t i = 0; i < c->phys_bytes.size(); ++i) {
        HIPCK(c, hipMalloc(&c->phys[i], c->phys_bytes[i] + 256));
        HIPCK(c, hipMemsetAsync(c->phys[i], 0, c->phys_bytes[i] + 256, c->stream));
    }
    for (auto &L : c->layers) {
        L.out.n = c->max_batch; L.out.h = L.H; L.out.w = L.W; L.out.c = L.C;      // geometry even when nothing is stored
        if (L.storage < 0) continue;
        Storage &s = c->storages[L.storage];
        if (s.phys < 0) return fail(c, YOLO_ERR_STATE, "internal: storage without buffer");
        L.out.n = c->max_batch; L.out.h = L.H; L.out.w = L.W; L.out.c = L.C; L.out.stride = s.stride; L.out.dt = s.dt;
        L.out.ptr = (char *)c->phys[s.phys] + (size_t)L.ch_off * dt_size(s.dt);
    }
    // network input: 3 real channels padded to 8
    c->input.dt = c->dtype == YOLO_FP32 ? DT_F32 : c->dtype == YOLO_FP16 ? DT_F16 : DT_BF16;            // fp8 mode keeps the image in bf16
    size_t in_bytes = (size_t)c->max_batch * c->in_h * c->in_w * 8 * dt_size(c->input.dt);
    HIPCK(c, hipMalloc(&c->input.ptr, in_bytes));
    c->input.n = c->max_batch; c->input.h = c->in_h; c->input.w = c->in_w; c->input.c = 8; c->input.stride = 8;
    HIPCK(c, hipMalloc(&c->d_zeros, 4096)); HIPCK(c, hipMemsetAsync(c->d_zeros, 0, 4096, c->stream));
    for (size_t i = 0; i < c->layers.size(); ++i) {
        const Layer &L = c->layers[i];
        if (L.s2d7) {
            c->s2d = c->input; c->s2d.h = c->in_h / 2; c->s2d.w = c->in_w / 2; c->s2d.c = 32; c->s2d.stride = 32;
            HIPCK(c, hipMalloc(&c->s2d.ptr, (size_t)c->max_batch * c->s2d.h * c->s2d.w * 32 * dt_size(c->s2d.dt)));
        }
        if (L.fc) {             // the flattened producer must be dense: one pixel of fc_h * fc_w * fc_c contiguous elements per image
            const TView in = view_of(c, L.in[0]);
            if (in.stride != in.c || in.c != L.fc_c) return fail(c, YOLO_ERR_UNSUPPORTED, "layer %zu: [connected] needs a dense producer (channels a multiple of 8, not part of a concat)", i);
        }
    }
    c->stage_bytes = (size_t)c->max_batch * c->in_h * c->in_w * 3 * 4;
    HIPCK(c, hipMalloc(&c->d_stage, c->stage_bytes));
    size_t nr = (size_t)c->max_batch * c->rows;
    HIPCK(c, hipMalloc((void **)&c->d_det, nr * c->attrs * 4));
    HIPCK(c, hipMalloc((void **)&c->d_box4, nr * 16));
    HIPCK(c, hipMalloc(&c->d_lean_list, nr * 16)); HIPCK(c, hipMalloc((void **)&c->d_lean_cnt, 16)); HIPCK(c, hipMemsetAsync(c->d_lean_cnt, 0, 16, c->stream));
    c->lean_ok = true;                  // every head a [yolo] head the cell-per-wave decode serves
    for (auto &L : c->layers) {
        if (L.type == L_REGION || L.type == L_DETECT) c->lean_ok = false;
        if (L.type == L_YOLO && L.na * (5 + L.classes) > 256) c->lean_ok = false;
    }
    c->lean_heads = 0;
    if (c->lean_ok) {
        int classes = -1; bool same = true;
        for (auto &L : c->layers) if (L.type == L_YOLO) { ++c->lean_heads; same = same && (classes < 0 || classes == L.classes) && L.na <= 16 && 5 + L.classes <= 128; classes = L.classes; }
        if (!same || (size_t)c->max_batch * c->rows * 340 >= 0xffffffffull) c->lean_heads = 0;      // 32-bit element offsets in the kernel
    }
    c->rows_pow2 = 1; while (c->rows_pow2 < c->rows) c->rows_pow2 <<= 1;
    HIPCK(c, hipMalloc((void **)&c->d_scores, nr * 4)); HIPCK(c, hipMalloc((void **)&c->d_labels, nr * 4));
    HIPCK(c, hipMalloc((void **)&c->d_cand, nr * 4)); HIPCK(c, hipMalloc((void **)&c->d_keys, (size_t)c->max_batch * c->rows_pow2 * 8));
    HIPCK(c, hipMalloc((void **)&c->d_sbox, nr * 16)); HIPCK(c, hipMalloc((void **)&c->d_slabel, nr * 4)); HIPCK(c, hipMalloc((void **)&c->d_sscore, nr * 4));
    HIPCK(c, hipMalloc((void **)&c->d_counts, (size_t)c->max_batch * 4));
    // filters
    for (auto &L : c->layers) if (L.type == L_CONV) {
        size_t wb = (size_t)L.cout_pad * L.kpad * dt_size(L.in_dt);
        HIPCK(c, hipMalloc(&L.d_w, wb)); HIPCK(c, hipMemsetAsync(L.d_w, 0, wb, c->stream));
        HIPCK(c, hipMalloc((void **)&L.d_b, (size_t)L.cout_pad * 4)); HIPCK(c, hipMemsetAsync(L.d_b, 0, (size_t)L.cout_pad * 4, c->stream));
        if (L.in_dt == DT_FP8) { HIPCK(c, hipMalloc((void **)&L.d_sc, (size_t)L.cout_pad * 4)); HIPCK(c, hipMemsetAsync(L.d_sc, 0, (size_t)L.cout_pad * 4, c->stream)); }
    }
    for (auto &L : c->layers) if (L.type == L_LOCAL) {
        const size_t wn = (size_t)L.H * L.W * L.filters * L.size * L.size * L.cin;
        HIPCK(c, hipMalloc(&L.d_w, wn * dt_size(L.in_dt))); HIPCK(c, hipMalloc((void **)&L.d_b, (size_t)L.H * L.W * L.filters * 4));
    }
    if (c->dtype != YOLO_FP32)
        for (size_t i = 1; i < c->layers.size(); ++i) {
            const Layer &Y = c->layers[i]; Layer &P = c->layers[i - 1];
            if (Y.type == L_YOLO && P.type == L_CONV && P.head && !P.fc) HIPCK(c, hipMalloc((void **)&P.d_obj, (size_t)c->max_batch * P.H * P.W * Y.na * 4));
        }
    HIPCK(c, hipStreamSynchronize(c->stream));
    return YOLO_OK;
}

// layers whose kernel is fixed by a fusion (nothing for the tile tuner to choose)
static bool fixed_kernel(const Layer &L) { return L.stem || L.stem_skip || L.stem_tail || L.halo || L.blk || L.blk_skip; }

ConvArgs conv_args(const yolo_ctx *c, const Layer &L, int n)
{
    ConvArgs a; memset(&a, 0, sizeof a);
    TView in = view_of(c, L.in[0]);
    a.in = in.ptr; a.in_stride = in.stride; a.wt = L.d_w; a.bias = L.d_b;
    a.out = L.out.ptr; a.out_stride = L.out.stride; a.out_dt = L.out.dt; a.in_dt = L.in_dt; a.oscale = L.d_sc;
    a.out_inv_scale = 1.f; a.res_scale = 1.f; a.mid_scale = 1.f; a.mid_inv_scale = 1.f;
    const int li = (int)(&L - c->layers.data());
    if (L.residual_from >= -1) { TView r = view_of(c, L.residual_from); a.res = r.ptr; a.res_stride = r.stride; }
    // the tail runs on the producer's operand type: bf16 needs the fragment-order copy of the 1x1 filters (tail_fragments), e4m3 an
    // e4m3-packed 1x1 conv; anything else leaves w2 null and run_layer refuses the plan instead of launching with a null w2f
    if (L.tail_on && L.tail_layer >= 0 && c->layers[L.tail_layer].in_dt == L.in_dt && (L.in_dt == DT_FP8 || c->layers[L.tail_layer].d_wf)) {
        const Layer &T = c->layers[L.tail_layer];
        a.w2 = T.d_w; a.w2f = T.d_wf; a.b2 = T.d_b; a.out2 = T.out.ptr; a.out2_stride = T.out.stride; a.K2pad = T.kpad; a.act2 = T.act;
        a.oscale2 = T.d_sc; a.out2_inv_scale = T.out.dt == DT_FP8 ? 1.f / c->eff_scale[L.tail_layer] : 1.f;
    }
    if (L.out.dt == DT_FP8) {
        if (L.residual_from >= -1) {     // fused shortcut: this conv writes layer li+1's tensor
            a.mid_scale = c->user_scale[li]; a.mid_inv_scale = 1.f / c->user_scale[li];
            a.res_scale = c->eff_scale[L.residual_from]; a.out_inv_scale = 1.f / c->eff_scale[li + 1];
        } else a.out_inv_scale = 1.f / c->eff_scale[li];
    }
    a.N = n; a.H = in.h; a.W = in.w; a.Cin_pad = L.cin_pad; a.Ho = L.H; a.Wo = L.W; a.Cout = L.filters;
    a.ksize = L.size; a.stride = L.stride; a.pad = L.pad; a.Kpad = L.kpad; a.kchunk = conv_kchunk(L.cin_pad, L.in_dt); a.act = L.act; a.zeros = c->d_zeros;
    if (L.d_obj && li + 1 < (int)c->layers.size()) { const Layer &Y = c->layers[li + 1]; a.obj_out = L.d_obj; a.obj_attrs = 5 + Y.classes; a.obj_na = Y.na; conv_magic((uint32_t)a.obj_attrs, a.obj_mul, a.obj_shift); }
    if (L.fc) { a.H = a.W = 1; a.in_stride = L.fc_h * L.fc_w * in.stride; }         // one "pixel" per image: the flattened producer
    if (L.s2d7) { a.in = c->s2d.ptr; a.in_stride = 32; a.H = c->s2d.h; a.W = c->s2d.w; a.ksize = 4; a.stride = 1; a.pad = 2; }   // see pack_s2d7
    conv_finalize(a);
    return a;
}

int run_layer(yolo_ctx *c, int i, int n)
{
    Layer &L = c->layers[i];
    hipStream_t s = c->stream;
    auto nview = [&](TView v) { v.n = n; return v; };
    switch (L.type) {
    case L_CONV: {
        if (L.stem_skip || L.stem_tail) break;
        if (L.fused_into >= 0 && c->layers[L.fused_into].tail_on) break;        // computed in the producer's epilogue
        if (L.stem) {
            const Layer &A = c->layers[0];
            StemArgs t; memset(&t, 0, sizeof t);
            t.in = c->input.ptr; t.in_stride = c->input.stride;
            t.in_u8 = c->stem_u8; t.in_scale = c->stem_scale; t.in_mul = c->in_mul; t.in_add = c->in_add;
            t.w0 = A.d_w; t.b0 = A.d_b; t.Kpad0 = A.kpad; t.C0 = A.filters; t.act0 = A.act;
            t.w1 = L.d_w; t.b1 = L.d_b; t.Kpad1 = L.kpad; t.C1 = L.filters; t.act1 = L.act;
            if (i + 1 < (int)c->layers.size() && c->layers[i + 1].stem_tail) {
                const Layer &T = c->layers[i + 1];
                t.w2 = T.d_w; t.b2 = T.d_b; t.Kpad2 = T.kpad; t.C2 = T.filters; t.act2 = T.act; t.out2 = T.out.ptr; t.out2_stride = T.out.stride;
            }
            t.out = L.out.ptr; t.out_stride = L.out.stride; t.N = n; t.H = A.H; t.W = A.W; t.Ho = L.H; t.Wo = L.W; t.zeros = c->d_zeros; t.dt = L.in_dt;
            HIPCK(c, launch_conv_stem(t, s));
            break;
        }
        if (L.blk_skip || L.blk) {
            // fused residual block: the 1x1 (blk_skip) is computed inside the 3x3's launch; were the batch window ever beyond the
            // kernel's 32-bit offsets, both run as ordinary layers
            const Layer &A = L.blk ? c->layers[i - 1] : L, &B = L.blk ? L : c->layers[i + 1];
            const TView x = view_of(c, A.in[0]);
            BlockArgs b; memset(&b, 0, sizeof b);
            b.x = x.ptr; b.x_stride = x.stride; b.w1 = A.d_w; b.b1 = A.d_b; b.Kpad1 = A.kpad; b.act1 = A.act;
            b.w2 = B.d_w; b.b2 = B.d_b; b.Kpad2 = B.kpad; b.act2 = B.act; b.out = B.out.ptr; b.out_stride = B.out.stride;
            b.N = n; b.H = B.H; b.W = B.W; b.C = B.filters; b.Cmid = A.filters; b.dt = B.in_dt;
            if (conv_resblock_ok(b)) {
                if (L.blk) HIPCK(c, launch_conv_resblock(b, s));
                break;
            }
        }
        ConvArgs a = conv_args(c, L, n);
        if (L.tail_on && !a.w2) return fail(c, YOLO_ERR_STATE, "layer %d: the plan folds the 1x1 conv %d into this layer, but its filters are not available in the producer's operand type", i, L.tail_layer);
        if (L.s2d7) HIPCK(c, launch_reorg(nview(c->input), nview(c->s2d), 2, 0, s));      // tf.space_to_depth order: (dy, dx, channel)
        if (L.halo) {          // small-Cin 3x3: input tile staged once in LDS (conv_stem.hip)
            HaloArgs h; memset(&h, 0, sizeof h);
            h.in = a.in; h.in_stride = a.in_stride; h.w = a.wt; h.b = a.bias; h.Kpad = a.Kpad; h.Cin = L.cin; h.Cout = L.filters; h.act = L.act;
            h.res = a.res; h.res_stride = a.res_stride; h.out = a.out; h.out_stride = a.out_stride; h.N = n; h.H = L.H; h.W = L.W; h.dt = L.in_dt;
            if (conv_halo_ok(h)) { HIPCK(c, launch_conv_halo(h, s)); break; }
            // window over 2 GiB (very large batches): the tiled kernel below checks its own window
        }
        if (c->dtype == YOLO_FP32) { HIPCK(c, launch_conv_f32(a, s)); }
        else if (L.in_dt == DT_FP8) {
            int cfg = L.tile_cfg >= 0 && conv_cfg_fp8_ok(L.tile_cfg) ? L.tile_cfg : conv_pick_cfg(a);
            if (conv_cfg_is_halo(cfg) && !conv_halo13_ok(a)) cfg = conv_pick_cfg(a);      // e.g. a smaller batch window or another input size
            if (a.w2 && !conv_cfg_tail_ok(cfg, a.Cout, a.in_dt == DT_FP8)) return fail(c, YOLO_ERR_STATE, "layer %d: tile config %d cannot run the fused 1x1 tail", i, cfg);
            HIPCK(c, launch_conv_fp8(a, cfg, s));
        } else {
            int cfg = L.tile_cfg >= 0 ? L.tile_cfg : conv_pick_cfg(a);
            if (cfg == CONV_CFG_DIRECT && !conv_c8_direct_ok(a)) cfg = conv_pick_cfg(a);
            if (conv_cfg_is_halo(cfg) && !conv_halo13_ok(a)) cfg = conv_pick_cfg(a);
            if (a.w2 && !conv_cfg_tail_ok(cfg, a.Cout, a.in_dt == DT_FP8)) return fail(c, YOLO_ERR_STATE, "layer %d: tile config %d cannot run the fused 1x1 tail", i, cfg);
            HIPCK(c, launch_conv_bf16(a, cfg, s));
        }
        break; }
    case L_SHORTCUT:
        if (!L.noop) {
            float sa = 1.f, sb = 1.f, so = 1.f;
            if (c->dtype == YOLO_FP8) { sa = c->eff_scale[L.in[0]]; sb = c->eff_scale[L.in[1]]; so = 1.f / c->eff_scale[i]; }
            HIPCK(c, launch_add(nview(view_of(c, L.in[0])), nview(view_of(c, L.in[1])), nview(L.out), s, sa, sb, so));
        }
        break;
    case L_ROUTE:
        for (size_t k = 0; k < L.copy_inputs.size(); ++k) {
            TView src = nview(view_of(c, L.copy_inputs[k])); TView dst = nview(L.out);
            dst.ptr = (char *)dst.ptr + (size_t)L.copy_offsets[k] * dt_size(dst.dt); dst.c = src.c;
            if (src.c % 8) return fail(c, YOLO_ERR_UNSUPPORTED, "route copy of %d channels", src.c);
            HIPCK(c, launch_copy(src, dst, s));
        }
        break;
    case L_LOCAL: HIPCK(c, launch_local(nview(view_of(c, L.in[0])), nview(L.out), L.d_w, L.d_b, L.size, L.stride, L.pad, L.act, s)); break;
    case L_UPSAMPLE: HIPCK(c, launch_upsample2x(nview(view_of(c, L.in[0])), nview(L.out), c->semantics == YOLO_SEM_TF, s)); break;
    case L_MAXPOOL: HIPCK(c, launch_maxpool(nview(view_of(c, L.in[0])), nview(L.out), L.psize, L.pstride, L.ppad, s)); break;
    case L_REORG: HIPCK(c, launch_reorg(nview(view_of(c, L.in[0])), nview(L.out), L.pstride, c->semantics == YOLO_SEM_DARKNET, s)); break;
    case L_DETECT: {
        const Layer &P = c->layers[i - 1];
        HIPCK(c, launch_decode_v1((const float *)P.out.ptr, P.out.stride, n, L.side, L.na, L.classes, L.sqr, c->d_det, c->rows, L.row_off,
                                  c->d_scores, c->d_labels, s));
        break; }
    case L_YOLO: case L_REGION: {
        if (c->lean && c->lean_thr > 0.f && c->lean_heads >= 1 && c->lean_heads <= 4 && !getenv("YOLO_NO_LEAN_MULTI")) {
            // lean detect path: every [yolo] head is decoded by ONE launch, issued at the last head (the head tensors keep their own buffers)
            bool later_head = false;
            for (size_t k = i + 1; k < c->layers.size(); ++k) later_head |= c->layers[k].type == L_YOLO;
            if (later_head) break;
            LeanArgs la; memset(&la, 0, sizeof la);
            long begin = 0;
            for (size_t k = 0; k < c->layers.size(); ++k) {
                const Layer &Y = c->layers[k];
                if (Y.type != L_YOLO) continue;
                const Layer &P = c->layers[k - 1];
                LeanHead &h = la.h[la.nheads++];
                h.raw = (const float *)P.out.ptr; h.obj = P.d_obj; h.raw_stride = P.out.stride; h.g = Y.H; h.na = Y.na; h.row_off = Y.row_off; h.box_begin = begin;
                const int stride = c->in_h / Y.H;
                for (int q = 0; q < 2 * Y.na; ++q) h.anchors[q] = (float)(1.0 * (double)Y.anchors[q] / (double)stride);
                begin += (long)n * Y.H * Y.W * Y.na;
            }
            la.total = begin; la.n = n; la.classes = L.classes; la.img_size = c->in_h; la.mode = c->decode; la.rows_total = c->rows;
            la.box4 = c->d_box4; la.reject_below = c->lean_thr; la.list = (uint4 *)c->d_lean_list; la.list_count = c->d_lean_cnt; la.list_cap = (unsigned)((size_t)c->max_batch * c->rows);
            // the list counter must be zero: the NMS launch of the previous detect call resets it; if none ran since the last decode
            // (a failed call in between), a memset does
            if (c->lean_cnt_dirty) HIPCK(c, hipMemsetAsync(c->d_lean_cnt, 0, 16, s));
            c->lean_cnt_dirty = true;
            HIPCK(c, launch_decode_lean(la, c->d_scores, c->d_labels, s));
            break;
        }
        DecodeArgs d; memset(&d, 0, sizeof d);
        const Layer &P = c->layers[i - 1];
        d.raw = (const float *)P.out.ptr; d.raw_stride = P.out.stride; d.n = n; d.g = L.H; d.na = L.na; d.classes = L.classes;
        d.img_size = c->in_h; d.mode = c->decode; d.region = L.type == L_REGION;
        const int stride = c->in_h / L.H;
        for (int k = 0; k < 2 * L.na; ++k)
            d.anchors[k] = L.type == L_YOLO ? (float)(1.0 * (double)L.anchors[k] / (double)stride) : L.anchors[k];
        d.det = c->lean ? nullptr : c->d_det; d.box4 = c->lean ? c->d_box4 : nullptr; d.rows_total = c->rows; d.row_off = L.row_off;
        d.reject_below = c->lean ? c->lean_thr : -INFINITY;
        HIPCK(c, launch_decode(d, c->d_scores, c->d_labels, s));
        break; }
    }
    return YOLO_OK;
}

int stage_in(yolo_ctx *c, const void *images, int n, int fmt, int loc, float scale)
{
    if (n < 1 || n > c->max_batch) return fail(c, YOLO_ERR_INVALID, "batch %d outside 1..%d", n, c->max_batch);
    if (!images) return fail(c, YOLO_ERR_INVALID, "images == NULL");
    size_t npix = (size_t)n * c->in_h * c->in_w;
    const void *src = images;
    if (loc == YOLO_HOST) {
        HIPCK(c, hipMemcpyAsync(c->d_stage, images, npix * 3 * (fmt == YOLO_IMG_U8 ? 1 : 4), hipMemcpyHostToDevice, c->stream));
        src = c->d_stage;
    }
    // uint8 images of a network whose first layers run as the fused stem: the stem converts the pixels itself (conv_stem.hip, U8 form)
    c->stem_u8 = nullptr;
    if (fmt == YOLO_IMG_U8 && c->layers.size() > 1 && c->layers[1].stem && !getenv("YOLO_NO_STEM_U8") && (double)npix * 3 < 2147483648.0 && ((size_t)src & 3) == 0) {
        c->stem_u8 = (const uint8_t *)src; c->stem_scale = scale;
        return YOLO_OK;
    }
    HIPCK(c, launch_preprocess(src, fmt, n, c->in_h * c->in_w, scale, c->input.ptr, c->input.dt, 8, c->stream, c->in_mul, c->in_add));
    return YOLO_OK;
}

int run_network(yolo_ctx *c, int n, bool lean = false)
{
    c->lean = lean && c->lean_ok;
    for (int i = 0; i < (int)c->layers.size(); ++i) { int r = run_layer(c, i, n); if (r) { c->lean = false; return r; } }
    c->last_n = n; c->scores_mode = 0; c->det_valid = !c->lean;
    return YOLO_OK;
}

int copy_out(yolo_ctx *c, void *dst, const void *src, size_t bytes, int loc)
{
    if (loc == YOLO_HOST) { HIPCK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream)); HIPCK(c, hipStreamSynchronize(c->stream)); }
    else HIPCK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->stream));
    return YOLO_OK;
}

int post(yolo_ctx *c, const float *det, int n, int rows, int attrs, float score_thr, float iou_thr, int max_out,
         int nms_mode, int select_mode, int img_h, int img_w, int scores_ready, yolo_box *boxes_out, int32_t *counts_out, int out_loc, int32_t *rows_out = nullptr)
{
    if (max_out < 1) return fail(c, YOLO_ERR_INVALID, "max_out < 1");
    if (nms_mode < 0 || nms_mode > 4 || select_mode < 0 || select_mode > 1) return fail(c, YOLO_ERR_INVALID, "bad nms/select mode");
    size_t need = (size_t)n * max_out;
    if ((int)need > c->boxes_cap) {
        // a captured detect graph holds the old pointer (memset, NMS writes, D2D copy): it must not be replayed
        if (c->gexec) { hipGraphExecDestroy(c->gexec); c->gexec = nullptr; }
        if (c->gstate > 0) c->gstate = 0;
        HIPCK(c, hipStreamSynchronize(c->stream));
        if (c->d_boxes) HIPCK(c, hipFree(c->d_boxes));
        if (c->d_rows) HIPCK(c, hipFree(c->d_rows));
        c->d_boxes = nullptr; c->d_rows = nullptr; c->boxes_cap = 0;
        HIPCK(c, hipMalloc(&c->d_boxes, need * sizeof(yolo_box))); HIPCK(c, hipMalloc((void **)&c->d_rows, need * 4)); c->boxes_cap = (int)need;
    }
    if (rows_out && !c->d_srow) HIPCK(c, hipMalloc((void **)&c->d_srow, (size_t)c->max_batch * c->rows * 4));
    PostArgs p; memset(&p, 0, sizeof p);
    p.det = det; p.box4 = (det == c->d_det && !c->det_valid) ? c->d_box4 : nullptr; p.n = n; p.rows = rows; p.attrs = attrs; p.score_thr = score_thr; p.iou_thr = iou_thr; p.max_out = max_out;
    p.nms_mode = nms_mode; p.select_mode = select_mode; p.img_h = img_h; p.img_w = img_w; p.scores_ready = scores_ready;
    p.scores = c->d_scores; p.labels = c->d_labels; p.cand = c->d_cand; p.keys = c->d_keys; p.rows_pow2 = c->rows_pow2;
    p.sbox = c->d_sbox; p.slabel = c->d_slabel; p.sscore = c->d_sscore; p.boxes_out = c->d_boxes; p.counts_out = c->d_counts;
    // device-resident outputs are written by the NMS kernel itself (it also zeroes the unused slots): no memset, no copies
    const bool direct_b = boxes_out && out_loc != YOLO_HOST, direct_c = counts_out && out_loc != YOLO_HOST;
    if (direct_b) p.boxes_out = boxes_out;
    if (direct_c) p.counts_out = (int *)counts_out;
    if (rows_out) { p.srow = c->d_srow; p.rows_out = out_loc != YOLO_HOST ? (int *)rows_out : c->d_rows; }
    if (c->lean_cnt_dirty) p.zero_word = c->d_lean_cnt;
    HIPCK(c, launch_postprocess(p, c->stream));
    c->lean_cnt_dirty = false;
    if (rows_out && out_loc == YOLO_HOST) { int r = copy_out(c, rows_out, c->d_rows, need * 4, out_loc); if (r) return r; }
    if (boxes_out && !direct_b) { int r = copy_out(c, boxes_out, c->d_boxes, need * sizeof(yolo_box), out_loc); if (r) return r; }
    if (counts_out && !direct_c) { int r = copy_out(c, counts_out, c->d_counts, (size_t)n * 4, out_loc); if (r) return r; }
    return YOLO_OK;
}

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

// fold + pack one conv's parameters (host).  w_oihw: [cout][cin][k][k].  wdt: element type of the packed filters.
// fp8: `in_scale` (per input channel, or null = 1) is folded into the filters first, then every output channel c is
// scaled so that its largest |w| maps to 448: code = e4m3(w * in_scale / osc[c]), osc[c] = max|w * in_scale| / 448.
void pack_conv(const Layer &L, const float *bn_or_bias, const float *w_oihw, int wdt, const float *in_scale,
               std::vector<uint8_t> &wbuf, std::vector<float> &bias, std::vector<float> &osc, int semantics = YOLO_SEM_TF)
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
        for (int ci = 0; ci < cin; ++ci)
            for (int t = 0; t < k * k; ++t) {
                const float v = row[(size_t)ci * k * k + t];
                const int kc = conv_kchunk(L.cin_pad, wdt);                              // K order: see conv_igemm.hip `stage`
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

}  // namespace

// ================================================================================================
extern "C" {

yolo_ctx *yolo_create(const yolo_config *cfg, char *err, size_t err_len)
{
    auto bail = [&](yolo_ctx *c, const std::string &m) -> yolo_ctx * { if (err && err_len) snprintf(err, err_len, "%s", m.c_str()); if (c) yolo_destroy(c); return nullptr; };
    if (!cfg || cfg->struct_size != sizeof(yolo_config)) return bail(nullptr, "yolo_create: bad yolo_config (struct_size)");
    if (cfg->max_batch < 1) return bail(nullptr, "yolo_create: max_batch < 1");
    if (cfg->dtype != YOLO_BF16 && cfg->dtype != YOLO_FP32 && cfg->dtype != YOLO_FP8 && cfg->dtype != YOLO_FP16) return bail(nullptr, "yolo_create: dtype");
    yolo_ctx *c = new yolo_ctx();
    c->device = cfg->device; c->max_batch = cfg->max_batch; c->dtype = cfg->dtype; c->semantics = cfg->semantics;
    c->decode = cfg->decode; c->keep_layers = cfg->keep_layers;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return bail(c, "yolo_create: no HIP device (this library has no CPU fallback)");
    if (hipSetDevice(c->device) != hipSuccess) return bail(c, "yolo_create: hipSetDevice failed");
    if (cfg->stream) c->stream = (hipStream_t)cfg->stream;
    else { if (hipStreamCreate(&c->stream) != hipSuccess) return bail(c, "yolo_create: hipStreamCreate failed"); c->own_stream = true; }
    std::vector<Section> secs; std::string perr;
    c->cfg_text = cfg->cfg_text ? cfg->cfg_text : "";
    if (!parse_cfg(cfg->cfg_text, secs, perr)) return bail(c, perr);
    if (build_plan(c, secs) != YOLO_OK) return bail(c, c->err);
    if (allocate(c) != YOLO_OK) return bail(c, c->err);
    resolve_scales(c);
    return c;
}

void yolo_destroy(yolo_ctx *c)
{
    if (!c) return;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    for (void *p : c->phys) if (p) hipFree(p);
    for (auto &L : c->layers) { if (L.d_w) hipFree(L.d_w); if (L.d_b) hipFree(L.d_b); if (L.d_sc) hipFree(L.d_sc); if (L.d_wf) hipFree(L.d_wf); if (L.d_obj) hipFree(L.d_obj); }
    void *ptrs[] = {c->input.ptr, c->d_zeros, c->d_stage, c->d_det, c->d_scores, c->d_labels, c->d_cand, c->d_keys, c->d_sbox, c->d_slabel, c->d_sscore, c->d_boxes, c->d_counts,
                    c->d_dn_rec, c->d_dn_src, c->d_dn_count, c->d_dn_last, c->d_box4, c->s2d.ptr, c->d_srow, c->d_rows, c->d_lean_list, c->d_lean_cnt};
    for (void *p : ptrs) if (p) hipFree(p);
    if (c->gexec) hipGraphExecDestroy(c->gexec);
    if (c->own_stream && c->stream) hipStreamDestroy(c->stream);
    delete c;
}

const char *yolo_last_error(const yolo_ctx *c) { return c ? c->err.c_str() : g_op_err.c_str(); }

size_t yolo_weights_count(const yolo_ctx *c) { return c ? c->weights_count : 0; }

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

// The fused 1x1 tail (conv_igemm_kernel.h) reads its filters as MFMA A fragments straight from global memory: lane (l15, lq) of a wave
// takes the 16 bytes at k = (kk * 4 + lq) * 8 of row ct2 * 16 + l15.  From the [row][K] image one such wave-load touches sixteen
// 64-byte pieces 2 * K bytes apart, and every workgroup of the layer asks for the same 64 KB at the same moment; a copy in fragment
// order -- [channel tile][K step][lane][8 bf16] -- makes each wave-load one contiguous KiB.  Built from the packed filters already on
// the device, so both ways of loading parameters (weight stream, export artifact) share it.
static int tail_fragments(yolo_ctx *c)
{
    std::vector<uint16_t> src, dst;
    for (auto &T : c->layers) {
        if (T.type != L_CONV || T.fused_into < 0) continue;
        if (T.in_dt != DT_BF16 && T.in_dt != DT_F16) continue;          // (16-bit tails only; also the bf16 islands of a mixed e4m3 plan)
        const int C2 = T.filters, K = T.kpad;                        // K == the producer's channel count, a multiple of 32
        if (C2 % 16 || K % 32) continue;
        src.resize((size_t)T.cout_pad * K); dst.resize((size_t)C2 * K);
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
        pack_conv(*PL, params, w, L.in_dt, in_sc.empty() ? nullptr : in_sc.data(), wbuf, bias, osc, c->semantics);
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
const uint32_t kArtVersion = 2;          // 2: filters packed chunk-major (conv_kchunk); a version-1 file holds tap-major filters
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

int yolo_input_size(const yolo_ctx *c, int *h, int *w, int *ch) { if (!c) return YOLO_ERR_INVALID; if (h) *h = c->in_h; if (w) *w = c->in_w; if (ch) *ch = c->in_c; return YOLO_OK; }
int yolo_num_rows(const yolo_ctx *c) { return c ? c->rows : YOLO_ERR_INVALID; }
int yolo_num_attrs(const yolo_ctx *c) { return c ? c->attrs : YOLO_ERR_INVALID; }
int yolo_num_layers(const yolo_ctx *c) { return c ? (int)c->layers.size() : YOLO_ERR_INVALID; }
int yolo_head_geometry(const yolo_ctx *c, int head, int *kind, int *grid, int *anchors, int *row_offset)
{
    if (!c || head < 0) return YOLO_ERR_INVALID;
    int k = 0;
    for (auto &L : c->layers) {
        if (L.type != L_YOLO && L.type != L_REGION && L.type != L_DETECT) continue;
        if (k++ != head) continue;
        if (kind) *kind = L.type == L_REGION ? 1 : L.type == L_DETECT ? 2 : 0;
        if (grid) *grid = L.H;
        if (anchors) *anchors = L.na;
        if (row_offset) *row_offset = L.row_off;
        return YOLO_OK;
    }
    return YOLO_ERR_INVALID;             // no such head
}
double yolo_conv_flops(const yolo_ctx *c) { return c ? c->conv_flops : 0; }
double yolo_conv_bytes(const yolo_ctx *c, int n)
{
    if (!c) return 0;
    double b = 0;
    for (auto &L : c->layers) if (L.type == L_CONV) {
        TView in = view_of(c, L.in[0]);
        const double ie = (double)dt_size(L.in_dt), oe = (double)dt_size(L.out.dt);
        if (!L.stem && !L.stem_tail && !L.blk) b += (double)n * in.h * in.w * L.cin * ie;         // the fused stem / residual block keep these inputs in LDS
        if (!L.stem_skip && !L.blk_skip) b += (double)n * L.H * L.W * L.filters * oe;
        b += (double)L.filters * L.cin * L.size * L.size * ie;
    }
    return b;
}

// lean: the caller goes straight on to threshold + NMS (yolo_detect*): the decoded tensor is not written, see yolo_ctx::lean
static int forward_impl(yolo_ctx *c, const void *images, int n, int fmt, int loc, float scale, float *det_out, int out_loc, bool lean, float score_thr = 0.f)
{
    if (!c) return YOLO_ERR_INVALID;
    c->lean_thr = score_thr;
    if (!c->weights_loaded) return fail(c, YOLO_ERR_STATE, "yolo_forward before weights were loaded");
    if (fmt != YOLO_IMG_U8 && fmt != YOLO_IMG_F32 && fmt != YOLO_IMG_F32_CHW) return fail(c, YOLO_ERR_INVALID, "bad image format");
    HIPCK(c, hipSetDevice(c->device));
    int r = stage_in(c, images, n, fmt, loc, scale); if (r) return r;
    r = run_network(c, n, lean && !det_out); if (r) return r;
    if (det_out) return copy_out(c, det_out, c->d_det, (size_t)n * c->rows * c->attrs * 4, out_loc);
    return YOLO_OK;
}

int yolo_forward(yolo_ctx *c, const void *images, int n, int fmt, int loc, float scale, float *det_out, int out_loc)
{
    return forward_impl(c, images, n, fmt, loc, scale, det_out, out_loc, false);
}

int yolo_forward_image_u8(yolo_ctx *c, const uint8_t *image, int h, int w, int loc, float *det_out, int out_loc)
{
    if (!c) return YOLO_ERR_INVALID;
    if (!c->weights_loaded) return fail(c, YOLO_ERR_STATE, "yolo_forward_image_u8 before weights were loaded");
    if (!image || h < 1 || w < 1) return fail(c, YOLO_ERR_INVALID, "bad image");
    HIPCK(c, hipSetDevice(c->device));
    const uint8_t *src = image; void *tmp = nullptr;
    if (loc == YOLO_HOST) {
        HIPCK(c, hipMalloc(&tmp, (size_t)h * w * 3));
        HIPCK(c, hipMemcpyAsync(tmp, image, (size_t)h * w * 3, hipMemcpyHostToDevice, c->stream)); src = (const uint8_t *)tmp;
    }
    c->stem_u8 = nullptr;
    hipError_t e = launch_resize_u8(src, h, w, c->in_h, c->input.ptr, c->input.dt, 8, 8, c->stream, c->in_mul, c->in_add);
    int r = e == hipSuccess ? run_network(c, 1) : fail(c, YOLO_ERR_HIP, "resize: %s", hipGetErrorString(e));
    if (tmp) { hipStreamSynchronize(c->stream); hipFree(tmp); }
    if (r) return r;
    if (det_out) return copy_out(c, det_out, c->d_det, (size_t)c->rows * c->attrs * 4, out_loc);
    return YOLO_OK;
}

int yolo_forward_letterbox_chw(yolo_ctx *c, const float *image_chw, int w, int h, int loc, float *det_out, int out_loc)
{
    if (!c) return YOLO_ERR_INVALID;
    if (!c->weights_loaded) return fail(c, YOLO_ERR_STATE, "yolo_forward_letterbox_chw before weights were loaded");
    if (!image_chw || h < 1 || w < 1) return fail(c, YOLO_ERR_INVALID, "bad image");
    HIPCK(c, hipSetDevice(c->device));
    const float *src = image_chw; void *tmp = nullptr;
    if (loc == YOLO_HOST) {
        HIPCK(c, hipMalloc(&tmp, (size_t)h * w * 3 * 4));
        HIPCK(c, hipMemcpyAsync(tmp, image_chw, (size_t)h * w * 3 * 4, hipMemcpyHostToDevice, c->stream)); src = (const float *)tmp;
    }
    c->stem_u8 = nullptr;
    hipError_t e = launch_letterbox_chw(src, w, h, c->in_h, c->input.ptr, c->input.dt, 8, c->stream);
    int r = e == hipSuccess ? run_network(c, 1) : fail(c, YOLO_ERR_HIP, "letterbox: %s", hipGetErrorString(e));
    if (tmp) { hipStreamSynchronize(c->stream); hipFree(tmp); }
    if (r) return r;
    if (det_out) return copy_out(c, det_out, c->d_det, (size_t)c->rows * c->attrs * 4, out_loc);
    return YOLO_OK;
}

int yolo_postprocess_rows(yolo_ctx *c, int n, float score_thr, float iou_thr, int max_out, int nms_mode, int select_mode,
                          yolo_box *boxes_out, int32_t *counts_out, int32_t *rows_out, int out_loc)
{
    if (!c) return YOLO_ERR_INVALID;
    if (n < 1 || n > c->last_n) return fail(c, YOLO_ERR_STATE, "postprocess of %d images but the last forward ran %d", n, c->last_n);
    HIPCK(c, hipSetDevice(c->device));
    const int want = nms_mode == YOLO_NMS_NUMPY_V3 ? 1 : 0;
    if (!c->det_valid && score_thr < c->lean_thr)
        return fail(c, YOLO_ERR_STATE, "the last forward ran through yolo_detect* with score threshold %g and pruned the scores below it; a lower threshold needs yolo_forward", c->lean_thr);
    if (!c->det_valid && want != c->scores_mode)
        return fail(c, YOLO_ERR_STATE, "the last forward ran through yolo_detect* without materialising the decoded tensor; this NMS flavour needs it (call yolo_forward)");
    const int ready = c->scores_mode == want;
    c->scores_mode = want;
    return post(c, c->d_det, n, c->rows, c->attrs, score_thr, iou_thr, max_out, nms_mode, select_mode,
                nms_mode == YOLO_NMS_PER_CLASS ? c->in_h : 0, nms_mode == YOLO_NMS_PER_CLASS ? c->in_w : 0, ready, boxes_out, counts_out, out_loc, rows_out);
}

int yolo_postprocess(yolo_ctx *c, int n, float score_thr, float iou_thr, int max_out, int nms_mode, int select_mode,
                     yolo_box *boxes_out, int32_t *counts_out, int out_loc)
{
    return yolo_postprocess_rows(c, n, score_thr, iou_thr, max_out, nms_mode, select_mode, boxes_out, counts_out, nullptr, out_loc);
}

int yolo_detect(yolo_ctx *c, const void *images, int n, int fmt, int loc, float scale, float score_thr, float iou_thr,
                int max_out, int nms_mode, int select_mode, yolo_box *boxes_out, int32_t *counts_out, int out_loc)
{
    int r = forward_impl(c, images, n, fmt, loc, scale, nullptr, YOLO_DEVICE, nms_mode != YOLO_NMS_NUMPY_V3, score_thr); if (r) return r;
    return yolo_postprocess(c, n, score_thr, iou_thr, max_out, nms_mode, select_mode, boxes_out, counts_out, out_loc);
}

int yolo_detect_graph(yolo_ctx *c, const void *images, int n, int fmt, float scale, float score_thr, float iou_thr, int max_out,
                      int nms_mode, int select_mode, yolo_box *boxes_out, int32_t *counts_out)
{
    if (!c) return YOLO_ERR_INVALID;
    if (!images || !boxes_out || !counts_out) return fail(c, YOLO_ERR_INVALID, "yolo_detect_graph needs device pointers for images, boxes_out and counts_out");
    yolo_ctx::GKey k{images, n, fmt, scale, score_thr, iou_thr, max_out, nms_mode, select_mode, (void *)boxes_out, (void *)counts_out};
    if (memcmp(&k, &c->gkey, sizeof k) != 0) {
        if (c->gexec) { hipGraphExecDestroy(c->gexec); c->gexec = nullptr; }
        c->gkey = k; if (c->gstate >= 0) c->gstate = 0;
    }
    auto eager = [&]() -> int {
        int r = forward_impl(c, images, n, fmt, YOLO_DEVICE, scale, nullptr, YOLO_DEVICE, nms_mode != YOLO_NMS_NUMPY_V3, score_thr); if (r) return r;
        return yolo_postprocess(c, n, score_thr, iou_thr, max_out, nms_mode, select_mode, boxes_out, counts_out, YOLO_DEVICE);
    };
    if (c->gstate <= 0) { int r = eager(); if (r == YOLO_OK && c->gstate == 0) c->gstate = 1; return r; }
    HIPCK(c, hipSetDevice(c->device));
    if (c->gstate == 1) {
        hipGraph_t g = nullptr;
        if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); c->gstate = -1; return eager(); }
        int r = eager();
        hipError_t e = hipStreamEndCapture(c->stream, &g);
        if (r != YOLO_OK || e != hipSuccess || !g) { (void)hipGetLastError(); if (g) hipGraphDestroy(g); c->gstate = -1; return r ? r : eager(); }
        e = hipGraphInstantiate(&c->gexec, g, nullptr, nullptr, 0);
        hipGraphDestroy(g);
        if (e != hipSuccess) { (void)hipGetLastError(); c->gexec = nullptr; c->gstate = -1; return eager(); }
        c->gstate = 2;
    }
    HIPCK(c, hipGraphLaunch(c->gexec, c->stream));
    // the replay leaves the context exactly as the eager call it was captured from would (run_network / forward_impl): a later
    // yolo_postprocess / yolo_darknet_boxes must see that the decoded tensor was (not) written and which threshold pruned the scores
    c->lean = nms_mode != YOLO_NMS_NUMPY_V3 && c->lean_ok; c->det_valid = !c->lean; c->lean_thr = score_thr;
    c->last_n = n; c->scores_mode = nms_mode == YOLO_NMS_NUMPY_V3 ? 1 : 0;
    return YOLO_OK;
}

// ---- darknet-flavoured views of the last forward (image 0), used by the veneer libdarknet_hip.so ----
int yolo_darknet_boxes(yolo_ctx *c, int w, int h, float thresh, int relative, float *records, int cap, int *count)
{
    if (!c || !count) return YOLO_ERR_INVALID;
    if (c->last_n < 1 || !c->det_valid) return fail(c, YOLO_ERR_STATE, "yolo_darknet_boxes needs a yolo_forward* pass first");
    if (w < 1 || h < 1 || cap < 0 || (cap > 0 && !records)) return fail(c, YOLO_ERR_INVALID, "bad image size / capacity");
    HIPCK(c, hipSetDevice(c->device));
    if (!c->d_dn_rec) {
        HIPCK(c, hipMalloc((void **)&c->d_dn_rec, (size_t)c->rows * c->attrs * 4));
        HIPCK(c, hipMalloc((void **)&c->d_dn_src, (size_t)c->rows * 4));
        HIPCK(c, hipMalloc((void **)&c->d_dn_count, 4));
    }
    DnBoxesArgs a; memset(&a, 0, sizeof a);
    a.det = c->d_det; a.attrs = c->attrs;
    for (size_t li = 0; li < c->layers.size(); ++li) {
        const Layer &L = c->layers[li];
        if (L.type != L_YOLO && L.type != L_REGION && L.type != L_DETECT) continue;
        if (a.nheads == 8) return fail(c, YOLO_ERR_UNSUPPORTED, "more than 8 heads");
        if (L.type == L_DETECT) {
            if (a.nheads) return fail(c, YOLO_ERR_UNSUPPORTED, "a [detection] head next to other heads");
            a.raw = (const float *)c->layers[li - 1].out.ptr; a.side = L.side; a.classes = L.classes; a.sqr = L.sqr;
            a.kind[0] = 2; a.grid[0] = L.side; a.na[0] = L.na; a.off[0] = L.row_off; a.nheads = 1;
            continue;
        }
        if (a.raw) return fail(c, YOLO_ERR_UNSUPPORTED, "a [detection] head next to other heads");
        a.kind[a.nheads] = L.type == L_REGION; a.grid[a.nheads] = L.H; a.na[a.nheads] = L.na; a.off[a.nheads] = L.row_off; ++a.nheads;
    }
    a.thresh = thresh; a.w = w; a.h = h; a.netw = c->in_w; a.neth = c->in_h; a.relative = relative;
    a.cap = cap < c->rows ? cap : c->rows;
    a.rec = a.cap > 0 ? c->d_dn_rec : nullptr; a.src = c->d_dn_src; a.count = c->d_dn_count;
    HIPCK(c, launch_darknet_boxes(a, c->stream));
    int n = 0;
    HIPCK(c, hipMemcpyAsync(&n, c->d_dn_count, 4, hipMemcpyDeviceToHost, c->stream)); HIPCK(c, hipStreamSynchronize(c->stream));
    *count = n;
    const int got = n < a.cap ? n : a.cap;
    if (got > 0) { HIPCK(c, hipMemcpyAsync(records, c->d_dn_rec, (size_t)got * c->attrs * 4, hipMemcpyDeviceToHost, c->stream)); HIPCK(c, hipStreamSynchronize(c->stream)); }
    return YOLO_OK;
}

size_t yolo_last_layer_size(const yolo_ctx *c)
{
    if (!c) return 0;
    for (int i = (int)c->layers.size() - 1; i >= 0; --i) {
        const Layer &L = c->layers[i];
        if (L.type == L_DETECT) return (size_t)L.side * L.side * (L.classes + 5 * L.na);      // the layer copies its input (DN/detection_layer.c:50-57)
        if (L.type == L_YOLO || L.type == L_REGION) return (size_t)L.H * L.W * L.na * (5 + L.classes);
    }
    return 0;
}

int yolo_last_layer_output_batch(yolo_ctx *c, int n, float *out, size_t out_floats)
{
    if (!c || !out) return YOLO_ERR_INVALID;
    if (c->last_n < 1) return fail(c, YOLO_ERR_STATE, "yolo_last_layer_output before a forward pass");
    if (n < 1 || n > c->last_n) return fail(c, YOLO_ERR_INVALID, "yolo_last_layer_output of %d images but the last forward ran %d", n, c->last_n);
    const int li = (int)c->layers.size() - 1;
    if (li < 1 || (c->layers[li].type != L_YOLO && c->layers[li].type != L_REGION && c->layers[li].type != L_DETECT))
        return fail(c, YOLO_ERR_UNSUPPORTED, "the last layer is not a detection head");
    const Layer &L = c->layers[li]; const Layer &P = c->layers[li - 1];
    const size_t per = yolo_last_layer_size(c), need = per * (size_t)n;
    if (out_floats < need) return fail(c, YOLO_ERR_INVALID, "output buffer too small (%zu < %zu floats)", out_floats, need);
    HIPCK(c, hipSetDevice(c->device));
    if (L.type == L_DETECT) {             // the prediction vectors as the fully connected layer left them (fp32, one "pixel" of P.out.stride floats per image)
        HIPCK(c, hipMemcpy2DAsync(out, per * 4, P.out.ptr, (size_t)P.out.stride * 4, per * 4, (size_t)n, hipMemcpyDeviceToHost, c->stream)); HIPCK(c, hipStreamSynchronize(c->stream));
        return YOLO_OK;
    }
    if (!c->d_dn_last || c->dn_last_cap < need) {
        if (c->d_dn_last) { HIPCK(c, hipStreamSynchronize(c->stream)); HIPCK(c, hipFree(c->d_dn_last)); c->d_dn_last = nullptr; c->dn_last_cap = 0; }
        HIPCK(c, hipMalloc((void **)&c->d_dn_last, need * 4)); c->dn_last_cap = need;
    }
    const size_t cells = (size_t)L.H * L.W;
    for (int b = 0; b < n; ++b)
        HIPCK(c, launch_head_darknet_layout((const float *)P.out.ptr + (size_t)b * cells * P.out.stride, P.out.stride, (int)cells, L.na, L.classes, L.type == L_REGION, c->d_dn_last + (size_t)b * per, c->stream));
    HIPCK(c, hipMemcpyAsync(out, c->d_dn_last, need * 4, hipMemcpyDeviceToHost, c->stream)); HIPCK(c, hipStreamSynchronize(c->stream));
    return YOLO_OK;
}

int yolo_last_layer_output(yolo_ctx *c, float *out, size_t out_floats) { return yolo_last_layer_output_batch(c, 1, out, out_floats); }

// The raw tensor a detection head decodes: the head conv's fp32 output [n, grid, grid, anchors * (5 + classes)] (what the reference's
// graph builders return before any decode: V2/model_darknet19_slim.py:198-200, V3/yolo_v3.py:239-263 `predictions`).
int yolo_head_raw(yolo_ctx *c, int head, int n, float *out, size_t out_floats)
{
    if (!c || !out || head < 0) return YOLO_ERR_INVALID;
    if (c->last_n < 1 || n < 1 || n > c->last_n) return fail(c, YOLO_ERR_STATE, "yolo_head_raw of %d images but the last forward ran %d", n, c->last_n);
    int k = 0;
    for (size_t li = 1; li < c->layers.size(); ++li) {
        const Layer &L = c->layers[li];
        if (L.type != L_YOLO && L.type != L_REGION && L.type != L_DETECT) continue;
        if (k++ != head) continue;
        const Layer &P = c->layers[li - 1];
        const size_t px = (size_t)n * P.H * P.W, need = px * P.C;
        if (out_floats < need) return fail(c, YOLO_ERR_INVALID, "output buffer too small (%zu < %zu floats)", out_floats, need);
        if (P.out.dt != DT_F32 || !P.out.ptr) return fail(c, YOLO_ERR_STATE, "internal: head conv output is not fp32");
        HIPCK(c, hipSetDevice(c->device));
        HIPCK(c, hipMemcpy2DAsync(out, (size_t)P.C * 4, P.out.ptr, (size_t)P.out.stride * 4, (size_t)P.C * 4, px, hipMemcpyDeviceToHost, c->stream));
        HIPCK(c, hipStreamSynchronize(c->stream));
        return YOLO_OK;
    }
    return fail(c, YOLO_ERR_INVALID, "no detection head %d", head);
}

int yolo_synchronize(yolo_ctx *c) { if (!c) return YOLO_ERR_INVALID; HIPCK(c, hipStreamSynchronize(c->stream)); return YOLO_OK; }

int yolo_layer_output(yolo_ctx *c, int index, int n, float *out, size_t out_floats, int *dims_out)
{
    if (!c) return YOLO_ERR_INVALID;
    if (!c->keep_layers) return fail(c, YOLO_ERR_STATE, "yolo_layer_output needs keep_layers=1");
    if (index < 0 || index >= (int)c->layers.size() || n < 1 || n > c->last_n) return fail(c, YOLO_ERR_INVALID, "bad layer index / n");
    const Layer &L = c->layers[index];
    if (dims_out) { dims_out[0] = L.H; dims_out[1] = L.W; dims_out[2] = L.C; }
    size_t need = (size_t)n * L.H * L.W * L.C;
    if (!out) return YOLO_OK;
    if (out_floats < need) return fail(c, YOLO_ERR_INVALID, "output buffer too small");
    HIPCK(c, hipSetDevice(c->device));
    float *tmp = nullptr; HIPCK(c, hipMalloc((void **)&tmp, need * 4));
    TView v = L.out; v.n = n;
    float vs = 1.f;
    if (v.dt == DT_FP8) {
        vs = c->eff_scale[index];
        if (vs != vs) { hipFree(tmp); return fail(c, YOLO_ERR_UNSUPPORTED, "layer %d concatenates tensors with different fp8 scales; read its sources", index); }
    }
    hipError_t e = launch_to_f32(v, tmp, c->stream, vs);
    if (e == hipSuccess) e = hipMemcpyAsync(out, tmp, need * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(tmp);
    if (e != hipSuccess) return fail(c, YOLO_ERR_HIP, "layer_output: %s", hipGetErrorString(e));
    return YOLO_OK;
}

int yolo_time_forward(yolo_ctx *c, int n, int iters, float *total_ms, float *conv_ms)
{
    if (!c) return YOLO_ERR_INVALID;
    if (!c->weights_loaded) return fail(c, YOLO_ERR_STATE, "weights not loaded");
    if (n < 1 || n > c->max_batch || iters < 1) return fail(c, YOLO_ERR_INVALID, "bad n/iters");
    HIPCK(c, hipSetDevice(c->device));
    hipEvent_t e0, e1; HIPCK(c, hipEventCreate(&e0)); HIPCK(c, hipEventCreate(&e1));
    c->lean = false;
    if (total_ms) {
        HIPCK(c, hipEventRecord(e0, c->stream));
        for (int it = 0; it < iters; ++it) { int r = run_network(c, n); if (r) return r; }
        HIPCK(c, hipEventRecord(e1, c->stream)); HIPCK(c, hipEventSynchronize(e1));
        float ms = 0; HIPCK(c, hipEventElapsedTime(&ms, e0, e1)); *total_ms = ms / iters;
    }
    if (conv_ms) {
        // conv time = (all layers) - (all layers except the convs), each timed as ONE event pair around `iters` passes:
        // events around every conv would add a record-to-record gap per launch (+5 % here), and calibrating that gap away
        // over-corrects; the difference of two bulk timings agrees with rocprofv3's kernel trace to ~1 %
        float all_ms = 0, rest_ms = 0;
        for (int pass = 0; pass < 2; ++pass) {
            HIPCK(c, hipEventRecord(e0, c->stream));
            for (int it = 0; it < iters; ++it)
                for (int i = 0; i < (int)c->layers.size(); ++i) {
                    if (pass == 1 && c->layers[i].type == L_CONV) continue;
                    int r = run_layer(c, i, n); if (r) return r;
                }
            HIPCK(c, hipEventRecord(e1, c->stream)); HIPCK(c, hipEventSynchronize(e1));
            float ms = 0; HIPCK(c, hipEventElapsedTime(&ms, e0, e1)); (pass == 0 ? all_ms : rest_ms) = ms / iters;
        }
        *conv_ms = all_ms - rest_ms;
        c->last_n = n; c->scores_mode = 0; c->det_valid = true;
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    return YOLO_OK;
}

int yolo_time_layers(yolo_ctx *c, int n, int iters, float *ms_out)
{
    if (!c || !ms_out) return YOLO_ERR_INVALID;
    if (!c->weights_loaded) return fail(c, YOLO_ERR_STATE, "weights not loaded");
    if (n < 1 || n > c->max_batch || iters < 1) return fail(c, YOLO_ERR_INVALID, "bad n/iters");
    HIPCK(c, hipSetDevice(c->device));
    const int NL = (int)c->layers.size();
    c->lean = false; c->det_valid = true;
    std::vector<hipEvent_t> ev(NL + 1);
    for (auto &e : ev) HIPCK(c, hipEventCreate(&e));
    std::vector<double> acc(NL, 0.0);
    for (int it = 0; it < iters; ++it) {
        HIPCK(c, hipEventRecord(ev[0], c->stream));
        for (int i = 0; i < NL; ++i) { int r = run_layer(c, i, n); if (r) return r; HIPCK(c, hipEventRecord(ev[i + 1], c->stream)); }
        HIPCK(c, hipStreamSynchronize(c->stream));
        for (int i = 0; i < NL; ++i) { float ms = 0; HIPCK(c, hipEventElapsedTime(&ms, ev[i], ev[i + 1])); acc[i] += ms; }
    }
    for (int i = 0; i < NL; ++i) ms_out[i] = (float)(acc[i] / iters);
    for (auto &e : ev) hipEventDestroy(e);
    c->last_n = n;
    return YOLO_OK;
}

// Tile selection is measured IN SITU: every candidate configuration is timed inside the real layer sequence (per-layer
// events around a full forward), not as the same kernel launched back to back.  Back-to-back timing flatters
// configurations that live off a warm L2: in the real sequence each layer's filters come cold from HBM (124 MB of
// filters and up to 350 MB of activations pass through the 32 MB of L2 / 256 MB of Infinity Cache between two uses), and
// the deep, filter-heavy layers ran 0.069 ms in the network against 0.050 ms in isolation.
int yolo_autotune(yolo_ctx *c, int n, int iters)
{
    if (!c) return YOLO_ERR_INVALID;
    if (!c->weights_loaded) return fail(c, YOLO_ERR_STATE, "weights not loaded");
    if (c->dtype == YOLO_FP32) return YOLO_OK;
    if (n < 1 || n > c->max_batch || iters < 1) return fail(c, YOLO_ERR_INVALID, "bad n/iters");
    HIPCK(c, hipSetDevice(c->device));
    const int NL = (int)c->layers.size();
    auto shape_key = [&](const Layer &L) {
        ConvArgs a = conv_args(c, L, n);
        char key[128]; snprintf(key, sizeof key, "%d_%d_%d_%d_%d_%d_%d%d_%d", a.H, a.W, a.Cin_pad, a.Cout, a.ksize, a.stride, a.in_dt, a.out_dt, a.res != nullptr);
        return std::string(key);
    };
    auto valid = [&](const Layer &L, int cfg) {
        if (fixed_kernel(L)) return false;                     // fused stem: nothing to choose
        ConvArgs a = conv_args(c, L, n);
        if (cfg == CONV_CFG_DIRECT) return conv_c8_direct_ok(a);
        if (a.in_dt == DT_FP8) return conv_cfg_fp8_ok(cfg);
        return true;
    };
    for (auto &L : c->layers) L.tail_on = false;
    std::vector<int> fallback(NL, -1);
    for (int i = 0; i < NL; ++i) if (c->layers[i].type == L_CONV && !fixed_kernel(c->layers[i])) { ConvArgs a = conv_args(c, c->layers[i], n); fallback[i] = conv_pick_cfg(a); }
    std::map<std::string, std::map<int, double>> score;          // shape -> cfg -> summed ms over the layers of that shape
    std::vector<float> ms(NL);
    for (int ci = 0; ci <= conv_num_cfgs(); ++ci) {
        const int cfg = ci == conv_num_cfgs() ? CONV_CFG_DIRECT : ci;
        bool any = false;
        for (int i = 0; i < NL; ++i) {
            Layer &L = c->layers[i];
            if (L.type != L_CONV) continue;
            const bool ok = valid(L, cfg);
            L.tile_cfg = ok ? cfg : fallback[i]; any |= ok;
        }
        if (!any) continue;
        // a configuration a layer cannot launch (LDS / 2 GiB window) must not abort the pass: probe once
        for (int i = 0; i < NL; ++i) {
            Layer &L = c->layers[i];
            if (L.type != L_CONV || L.tile_cfg != cfg || fixed_kernel(L)) continue;
            ConvArgs a = conv_args(c, L, n);
            hipError_t e = a.in_dt == DT_FP8 ? launch_conv_fp8(a, cfg, c->stream) : launch_conv_bf16(a, cfg, c->stream);
            if (e != hipSuccess) { (void)hipGetLastError(); L.tile_cfg = fallback[i]; }
        }
        int r = yolo_time_layers(c, n, iters, ms.data()); if (r) return r;
        for (int i = 0; i < NL; ++i) {
            const Layer &L = c->layers[i];
            if (L.type == L_CONV && L.tile_cfg == cfg && !fixed_kernel(L)) score[shape_key(L)][cfg] += ms[i];
        }
        if (getenv("YOLO_TUNE_VERBOSE")) {
            std::map<std::string, double> seen;
            for (int i = 0; i < NL; ++i) if (c->layers[i].type == L_CONV && c->layers[i].tile_cfg == cfg && !fixed_kernel(c->layers[i])) seen[shape_key(c->layers[i])] = score[shape_key(c->layers[i])][cfg];
            for (auto &kv : seen) fprintf(stderr, "tune %s cfg %d %-16s %.4f ms (sum over the layers of this shape, in situ)\n", kv.first.c_str(), cfg, conv_cfg_name(cfg), kv.second);
        }
    }
    for (int i = 0; i < NL; ++i) {
        Layer &L = c->layers[i];
        if (L.type != L_CONV || fixed_kernel(L)) continue;
        auto it = score.find(shape_key(L));
        int best = fallback[i]; double bt = 1e30;
        if (it != score.end()) for (auto &kv : it->second) if (kv.second < bt) { bt = kv.second; best = kv.first; }
        L.tile_cfg = best;
    }
    // second pass: fold 1x1 convs into their producers where that beats the best unfused pair.  Base = the plan just
    // chosen; candidate = every tail-capable tile shape on all producers at once; decided per producer shape.
    {
        int r = yolo_time_layers(c, n, iters, ms.data()); if (r) return r;
        std::vector<float> base(ms);
        std::vector<int> base_cfg(NL, -1);
        for (int i = 0; i < NL; ++i) base_cfg[i] = c->layers[i].tile_cfg;
        std::map<std::string, std::pair<double, int>> best;         // producer shape -> (pair time, cfg), cfg -1 = unfused
        for (int i = 0; i < NL; ++i) {
            const Layer &L = c->layers[i];
            if (L.type != L_CONV || L.tail_layer < 0) continue;
            auto &b = best[shape_key(L)];
            if (b.second == 0 && b.first == 0) b = {0.0, -1};
            b.first += base[i] + base[L.tail_layer];
        }
        for (int cfg = 0; cfg < conv_num_cfgs(); ++cfg) {
            bool any = false;
            for (int i = 0; i < NL; ++i) {
                Layer &L = c->layers[i];
                if (L.type != L_CONV || L.tail_layer < 0) continue;
                bool ok = conv_cfg_tail_ok(cfg, L.filters, L.in_dt == DT_FP8) && c->layers[L.tail_layer].in_dt == L.in_dt;
                if (ok && conv_cfg_is_halo(cfg)) { ConvArgs a = conv_args(c, L, n); ok = conv_halo13_ok(a); }
                L.tile_cfg = ok ? cfg : base_cfg[i]; L.tail_on = ok; any |= ok;
            }
            if (!any) continue;
            r = yolo_time_layers(c, n, iters, ms.data()); if (r) return r;
            std::map<std::string, double> t;
            for (int i = 0; i < NL; ++i) {
                const Layer &L = c->layers[i];
                if (L.type == L_CONV && L.tail_layer >= 0 && L.tail_on) t[shape_key(L)] += ms[i] + ms[L.tail_layer];
            }
            for (auto &kv : t) {
                auto &b = best[kv.first];
                if (getenv("YOLO_TUNE_VERBOSE")) fprintf(stderr, "tune-tail %s cfg %d %-16s fused pair %.4f ms (unfused best so far %.4f)\n", kv.first.c_str(), cfg, conv_cfg_name(cfg), kv.second, b.first);
                if (kv.second < b.first) b = {kv.second, cfg};
            }
        }
        for (int i = 0; i < NL; ++i) {
            Layer &L = c->layers[i];
            if (L.type != L_CONV || L.tail_layer < 0) continue;
            const auto &b = best[shape_key(L)];
            L.tail_on = b.second >= 0; L.tile_cfg = b.second >= 0 ? b.second : base_cfg[i];
        }
    }
    if (c->gexec) { hipGraphExecDestroy(c->gexec); c->gexec = nullptr; } if (c->gstate > 0) c->gstate = 0;
    return YOLO_OK;
}

int yolo_get_tile_configs(const yolo_ctx *c, int32_t *cfgs)
{
    if (!c || !cfgs) return YOLO_ERR_INVALID;
    // a conv whose plan folds the following 1x1 conv into its epilogue is reported as cfg + 10000
    for (size_t i = 0; i < c->layers.size(); ++i) {
        const Layer &L = c->layers[i];
        cfgs[i] = L.type == L_CONV ? (L.tail_on && L.tile_cfg >= 0 ? L.tile_cfg + 10000 : L.tile_cfg) : -1;
    }
    return YOLO_OK;
}

int yolo_set_tile_configs(yolo_ctx *c, const int32_t *cfgs)
{
    if (!c || !cfgs) return YOLO_ERR_INVALID;
    for (size_t i = 0; i < c->layers.size(); ++i) {
        if (c->layers[i].type != L_CONV) continue;
        int v = cfgs[i]; bool tail = false;
        if (v >= 10000) { v -= 10000; tail = true; }
        if (v != -1 && v != CONV_CFG_DIRECT && (v < 0 || v >= conv_num_cfgs())) return fail(c, YOLO_ERR_INVALID, "layer %zu: tile config %d out of range", i, v);
        if (tail && (c->layers[i].tail_layer < 0 || !conv_cfg_tail_ok(v, c->layers[i].filters, c->layers[i].in_dt == DT_FP8) || c->layers[c->layers[i].tail_layer].in_dt != c->layers[i].in_dt))
            return fail(c, YOLO_ERR_INVALID, "layer %zu: plan asks for a fused 1x1 tail this layer / tile config cannot run", i);
        if (tail && conv_cfg_is_halo(v)) {
            ConvArgs a = conv_args(c, c->layers[i], c->max_batch);
            if (!conv_halo13_ok(a)) return fail(c, YOLO_ERR_INVALID, "layer %zu: the halo-staged tile config %d does not apply to this layer, so it cannot carry the fused 1x1 tail", i, v);
        }
        c->layers[i].tile_cfg = v; c->layers[i].tail_on = tail;
    }
    if (c->gexec) { hipGraphExecDestroy(c->gexec); c->gexec = nullptr; } if (c->gstate > 0) c->gstate = 0;
    return YOLO_OK;
}

// ---- single operators -----------------------------------------------------------------------
int yolo_op_conv_num_cfgs(void) { return conv_num_cfgs(); }

int yolo_op_conv2d(const float *x, int n, int h, int w, int cin, const float *w_hwio, const float *bias, int k, int stride,
                   int cout, int act, const float *residual, float *out, int dtype, int tile_cfg, int device)
{
    if (!x || !w_hwio || !out || n < 1 || (k != 1 && k != 3) || stride < 1) { g_op_err = "conv2d: bad arguments"; return YOLO_ERR_INVALID; }
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
    if (conv_cfg_is_halo(tile_cfg) && (f32 || !conv_halo13_ok(a))) { g_op_err = "conv2d: tile config not applicable to this shape (halo-staged form: 3x3, stride 1, size a multiple of 13, whole channel chunks)"; return YOLO_ERR_UNSUPPORTED; }
    hipError_t e;
    if (dt != DT_F32 && dt != DT_F16 && getenv("YOLO_CONV_DIAG") && a.Cin_pad % (dt == DT_FP8 ? 128 : 64) == 0) {
        // developer diagnostic: phase cycle sums of the stamped p176c128_s2 build (YOLO_CONV_DIAG=free: of the free-running halo form
        // f176c256), printed to stderr
        const bool dwide = !strcmp(getenv("YOLO_CONV_DIAG"), "free4");        // four waves of 176 x 64
        const bool dfree = (dwide || !strcmp(getenv("YOLO_CONV_DIAG"), "free")) && conv_halo13_ok(a) && dt == DT_BF16;
        const int wv = dfree && !dwide ? 8 : 4;
        const long tiles = dfree ? (long)n * (h / 13) * (w / 13) * ((cout + 255) / 256) : (((long)n * ho * wo + 175) / 176) * ((cout + 127) / 128);
        a.dbg = (unsigned long long *)S.alloc((size_t)tiles * wv * 16 * 8);
        if (dfree && !dwide && getenv("YOLO_CONV_DIAG_TAIL") && cout == 256) {
            // time the fused 1x1 tail too: any 128 x 256 filter block will do (the main filters' first rows), output to scratch
            a.w2 = a.wt; a.w2f = a.wt; a.K2pad = a.Kpad; a.b2 = a.bias; a.act2 = ACT_LEAKY; a.out2_stride = 128;
            a.out2 = S.alloc((size_t)n * ho * wo * 128 * 2);
        }
        for (int rep = 0; rep < 200; ++rep) e = dfree ? launch_conv_halo13_diag(a, S.s, dwide ? 1 : 0) : launch_conv_diag(a, S.s);     // long enough for the clock to settle under load
        std::vector<unsigned long long> hd((size_t)tiles * wv * 16);
        S.download(hd.data(), a.dbg, hd.size() * 8);
        double sum[16] = {0}; size_t cnt = hd.size() / 16;
        const unsigned long long kt = hd[5] >> 40;
        for (size_t i = 0; i < cnt; ++i)
            for (int q = 0; q < 16; ++q) sum[q] += q == 5 ? (double)(hd[i * 16 + 5] & 0xffffffffffull) : (double)hd[i * 16 + q];
        for (double &v : sum) v /= cnt;
        fprintf(stderr, "diag: waves %zu KT %llu | per K-step cycles: wait+barrier %.0f  issue %.0f  ds_read+mfma %.0f  (loop total/KT %.0f) | epilogue %.0f cycles | shader clock %.0f MHz\n",
                cnt, kt, sum[0] / kt, sum[1] / kt, sum[2] / kt, sum[3] / kt, sum[4], sum[5]);
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
