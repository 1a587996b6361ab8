// Planner of libyolo_hip.so: darknet cfg text -> layers, fusions, buffer pool (see yolo_ctx.h for the map of the host side).
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
#include "yolo_ctx.h"

namespace yolo_impl {

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
    // split-fp16 networks: per-tensor storage form (mixed plans): pairs unless a [convolutional] section says yolo_pair=0
    // ([net] yolo_pair_input for the image); layers that move data inherit, both operands of a shortcut / all inputs of a route must agree
    c->in_pair = c->split() && opt_i(net, "yolo_pair_input", 1) != 0;
    for (int i = 0; i < NL; ++i) {
        const Section &s = secs[i + 1]; Layer &L = c->layers[i];
        L.in = {i - 1};
        if (!c->split() && (s.kv.count("yolo_pair") || net.kv.count("yolo_pair_input"))) return fail(c, YOLO_ERR_UNSUPPORTED, "layer %d: yolo_pair is a key of split-fp16 networks (YOLO_FP16X2)", i);
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
            // the conv kernel's view of a pair tensor: a layer output is interleaved per 32-channel group, 2 * roundup(C, 32) elements per
            // pixel and tap (pair K loop); the network input keeps the three blocks hi | lo | hi of its 8 padded channels (plain K loop)
            if (c->pair_of(i - 1)) L.cin_pad = i - 1 < 0 ? 3 * roundup(C, 8) : pair_width(C);
            L.pair = c->split() && opt_i(s, "yolo_pair", 1) != 0;
            L.kpad = roundup(L.size * L.size * L.cin_pad, L.in_dt == DT_FP8 ? 128 : 64); L.cout_pad = roundup(L.filters, 256);
            if (L.s2d7) { L.cin_pad = 32; L.kpad = 16 * 32; }          // 4x4 taps x (2x2 positions x 8 padded channels)
            H = (H + 2 * L.pad - L.size) / L.stride + 1; W = (W + 2 * L.pad - L.size) / L.stride + 1; C = L.filters;
            c->conv_flops += 2.0 * L.size * L.size * L.cin * L.filters * (double)H * W;
            c->weights_count += (size_t)L.filters * (L.bn ? 4 : 1) + (size_t)L.filters * L.cin * L.size * L.size;
        } else if (s.type == "connected") {
            if (c->dtype == YOLO_FP8 || c->split()) return fail(c, YOLO_ERR_UNSUPPORTED, "layer %d: [connected] is not served in the fp8 / split-fp16 configurations", i);
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
            if (c->dtype == YOLO_FP8 || c->split()) return fail(c, YOLO_ERR_UNSUPPORTED, "layer %d: [local] is not served in the fp8 / split-fp16 configurations", i);
            L.type = L_LOCAL; L.filters = opt_i(s, "filters", 1); L.size = opt_i(s, "size", 1); L.stride = opt_i(s, "stride", 1); L.pad = opt_i(s, "pad", 0);
            if (L.pad != 0 && L.pad != 1) return fail(c, YOLO_ERR_UNSUPPORTED, "layer %d: [local] pad must be 0 or 1", i);
            // pad=1 pads by ONE pixel whatever the size (DN/local_layer.c:103 im2col) while the output size assumes size / 2 (:10-24): they only agree for 3x3
            if (L.pad == 1 && L.size != 3) return fail(c, YOLO_ERR_UNSUPPORTED, "layer %d: [local] with pad=1 needs size=3 (darknet's own output size and im2col disagree otherwise)", i);
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
            if (c->split() && !L.in.empty() && L.type != L_YOLO && L.type != L_REGION && L.type != L_DETECT) {
                L.pair = c->pair_of(L.in[0]);
                for (int j : L.in) if (c->pair_of(j) != L.pair) return fail(c, YOLO_ERR_INVALID, "layer %d: operands stored in different forms (yolo_pair): a %s needs pairs or plain fp16 throughout", i, L.type == L_ROUTE ? "route" : L.type == L_SHORTCUT ? "shortcut" : "layer");
            }
        }
    }
    for (int i = 0; i < NL; ++i) if (c->layers[i].head) c->layers[i].pair = false;        // heads are fp32
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
            if (P.type == L_CONV && uses[i - 1] == 1 && !P.head && L.in[1] != i - 1 && (L.in[1] < 0 || c->layers[L.in[1]].store_dt == P.store_dt)) {
                if (c->pair_of(L.in[1]) != P.pair) return fail(c, YOLO_ERR_INVALID, "layer %d: the conv in front of this shortcut and its other operand are stored in different forms (yolo_pair)", i);
                P.residual_from = L.in[1]; L.noop = true;
            }
        }
    }
    // fused stem: conv0 (3x3/s1, 3 -> 32) read only by conv1 (3x3/s2, 32 -> 64), bf16, nothing asking for layer 0's tensor
    // (the stem and halo kernels address their input with 32-bit buffer offsets: the whole-batch window must stay under 2 GiB)
    // (an e4m3 network whose first layers are stored in a 16-bit type -- a mixed plan, yolo_store=bf16 -- runs them through the same fused
    //  kernels: what counts is the type of the tensors a kernel touches, not the context's)
    const bool ctx16 = c->half_like() || c->dtype == YOLO_FP8 || c->split();        // (a split-fp16 network: where the tensors a fused kernel touches are PLAIN fp16 -- mixed plans)
    auto is16 = [](int dt) { return dt == DT_BF16 || dt == DT_F16; };
    // fused residual block (conv_block.hip, conv_block64.hip): a 1x1 conv C -> C/2 read only by the 3x3 conv C/2 -> C that follows, whose folded
    // shortcut source is the 1x1's own input, on a grid that is (nearly) whole 13 x 13 blocks: C = 128, darknet-53's 104 x 104 stage at
    // 416 x 416, and (round 5) C = 64, its first residual block at 208 x 208 -- found BEFORE the stem's 1x1 tail and the halo-staged 32 -> 64
    // form, which would otherwise take those two layers
    if (ctx16 && !c->keep_layers && !getenv("YOLO_NO_RESBLOCK"))
        for (int i = 1; i + 1 < NL; ++i) {
            Layer &A = c->layers[i], &B = c->layers[i + 1];
            const bool c128 = A.cin == 128 && A.filters == 64 && B.cin == 64 && B.filters == 128;
            // (C = 64: built and bit-identical, but SLOWER than what it replaces -- 130 us against 107 for the halo-staged 32 -> 64 conv, the stem no
            //  faster without its 1x1 tail: the block is instruction-issue-bound, ~800 instructions per wave and block for 62 MFMAs, docs/NOTEBOOK.md
            //  round 5 -- so it is opt-in: YOLO_RESBLOCK64=1)
            const bool c64 = A.cin == 64 && A.filters == 32 && B.cin == 32 && B.filters == 64 && getenv("YOLO_RESBLOCK64") && !getenv("YOLO_NO_RESBLOCK64");
            if (A.type == L_CONV && B.type == L_CONV && !A.fc && !B.fc && !A.head && !B.head && uses[i] == 1 && B.in[0] == i && A.in[0] >= 0 &&
                A.size == 1 && A.stride == 1 && A.pad == 0 && (c128 || c64) && A.residual_from < -1 &&
                B.size == 3 && B.stride == 1 && B.pad == 1 && B.residual_from == A.in[0] &&
                ((long)((B.H + 12) / 13) * ((B.W + 12) / 13) * 169 * 100 <= (long)B.H * B.W * 115) && A.in_dt == B.in_dt && (A.in_dt == DT_BF16 || A.in_dt == DT_F16) && A.store_dt == A.in_dt && B.store_dt == A.in_dt &&
                c->layers[A.in[0]].store_dt == A.in_dt && !A.pair && !B.pair && !c->pair_of(A.in[0])) { A.blk_skip = true; B.blk = true; }
        }
    if (ctx16 && !c->keep_layers && NL >= 2 && !getenv("YOLO_NO_STEM") && (double)c->max_batch * c->in_h * c->in_w * 8 * 2 < 2147483648.0) {
        const Layer &A = c->layers[0], &B = c->layers[1];
        if (A.type == L_CONV && B.type == L_CONV && uses[0] == 1 && B.in[0] == 0 && A.size == 3 && A.stride == 1 && A.pad == 1 && A.cin == 3 &&
            A.filters == 32 && B.size == 3 && B.stride == 2 && B.pad == 1 && B.filters == 64 && !A.head && !B.head && B.residual_from < -1 &&
            is16(A.in_dt) && A.store_dt == A.in_dt && B.in_dt == A.in_dt && B.store_dt == A.in_dt && !c->in_pair && !A.pair && !B.pair) {        // (layer 0 reads the staged image, which is kept in its operand type)
            c->layers[0].stem_skip = true; c->layers[1].stem = true;
            if (NL >= 3) {
                const Layer &T = c->layers[2];
                if (T.type == L_CONV && !T.fc && T.in[0] == 1 && T.size == 1 && T.stride == 1 && T.pad == 0 && T.filters == 32 && !T.head && T.residual_from < -1 &&
                    T.in_dt == A.in_dt && T.store_dt == A.in_dt && !T.blk_skip && !T.pair)        // (blk_skip: the fused first residual block computes it)
                    c->layers[2].stem_tail = true;
            }
        }
    }
    // split-fp16: conv0 + conv1 on pairs in one launch (conv_stem_pair.hip) -- both tensors pairs, the image in its three blocks
    if (c->split() && !c->keep_layers && NL >= 2 && c->in_pair && !getenv("YOLO_NO_PAIR_STEM")) {
        const Layer &A = c->layers[0], &B = c->layers[1];
        if (A.type == L_CONV && B.type == L_CONV && uses[0] == 1 && B.in[0] == 0 && A.size == 3 && A.stride == 1 && A.pad == 1 && A.cin == 3 && A.bn == B.bn &&
            (A.filters == 32) && B.size == 3 && B.stride == 2 && B.pad == 1 && B.filters == 64 && !A.head && !B.head && A.residual_from < -1 && B.residual_from < -1 &&
            A.pair && B.pair && c->in_h % 2 == 0 && c->in_w % 2 == 0 && A.kpad >= 216 && B.kpad == 576) {
            c->layers[0].pstem_skip = true; c->layers[1].pstem = true;
        }
    }
    if (ctx16 && !getenv("YOLO_NO_HALO"))
        for (int i = 1; i < NL; ++i) {
            Layer &L = c->layers[i];
            if (L.type == L_CONV && !L.head && !L.stem && !L.stem_skip && !L.stem_tail && !L.blk && L.size == 3 && L.stride == 1 && L.pad == 1 && L.cin == 32 && L.filters == 64 &&
                is16(L.in_dt) && L.store_dt == L.in_dt && (L.residual_from < 0 || c->layers[L.residual_from].store_dt == L.in_dt) && !L.pair && !c->pair_of(L.in[0]))
                L.halo = true;
            // darknet-53's 64 -> 128 downsampling conv: window-staged, filters in registers (conv_s2.hip)
            if (L.type == L_CONV && !L.head && !L.stem && !L.stem_skip && !L.stem_tail && !L.blk && !L.blk_skip && L.size == 3 && L.stride == 2 && L.pad == 1 && L.cin == 64 &&
                L.filters == 128 && L.residual_from < -1 && is16(L.in_dt) && L.store_dt == L.in_dt && !L.pair && !c->pair_of(L.in[0]) && !getenv("YOLO_NO_S2"))
                L.s2 = true;
        }
    // 1x1 convs that can ride in their producer's epilogue: conv i (bf16, 128 or 256 output channels, optionally with its
    // fused shortcut) read by a 1x1/s1 conv with half as many filters
    if (ctx16 && !c->keep_layers && !getenv("YOLO_NO_TAIL")) {
        for (int i = 0; i + 1 < NL; ++i) {
            Layer &P = c->layers[i];
            if (P.type != L_CONV || P.fc || P.head || P.stem || P.stem_skip || P.stem_tail || P.blk || P.s2 || P.halo || (P.filters != 128 && P.filters != 256)) continue;      // (fixed kernels host no tail: run_layer would skip the 1x1)
            if (c->split() && (P.pair || c->pair_of(P.in[0]))) continue;       // (split-fp16 networks: the tail rides on plain fp16 layers only)
            int o = i;
            if (P.residual_from >= -1) o = i + 1;            // its shortcut was folded into it: consumers read layer i+1
            const int j = o + 1;
            if (j >= NL) continue;
            Layer &T = c->layers[j];
            if (T.type == L_CONV && !T.fc && T.in[0] == o && T.size == 1 && T.stride == 1 && T.pad == 0 && T.filters * 2 == P.filters && !T.head &&
                T.residual_from < -1 && !T.stem_tail && !T.blk_skip && T.in_dt == P.in_dt && !T.pair) { P.tail_layer = j; T.fused_into = i; }      // (same operand type: the tail runs on the producer's MFMA)
            // round 5: a detection head (1x1, <= 256 filters, fp32 out, linear) as the tail of the 256-channel 3x3 in front of it when nobody else
            // reads that conv (darknet-53's 52 x 52 head): the head tensor is formed from the tile in LDS, bit-identical to the stand-alone launch
            else if (T.type == L_CONV && !T.fc && T.head && T.in[0] == o && uses[o] == 1 && T.size == 1 && T.stride == 1 && T.pad == 0 && P.filters == 256 && T.filters <= 256 &&
                     P.size == 3 && P.residual_from < -1 && T.act == ACT_LINEAR && T.in_dt == P.in_dt && (T.in_dt == DT_BF16 || T.in_dt == DT_F16) && !getenv("YOLO_NO_HEAD_TAIL") && j + 1 < NL && c->layers[j + 1].type == L_YOLO) { P.tail_layer = j; T.fused_into = i; }
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
            // (pair tensors, round 6: interleaved per 32-channel group, so a source of whole groups at a whole-group offset is a window of the
            //  concatenation like any other tensor; element offset 2 x the channel offset)
            const int gran = L.pair ? 32 : gran_of(L.store_dt);
            bool ok = j >= 0 && place_route[j] < 0 && c->layers[j].type != L_ROUTE && !c->layers[j].head &&
                      c->layers[j].type != L_YOLO && c->layers[j].type != L_REGION && c->layers[j].type != L_DETECT && (cj % gran == 0) && (off % gran == 0);
            // a fused-away conv's real producer is the conv; the shortcut layer itself is what gets placed
            if (ok && c->layers[j].type == L_CONV && j + 1 < NL && c->layers[j + 1].noop && c->layers[j + 1].type == L_SHORTCUT) ok = false;
            if (ok) { place_route[j] = i; place_off[j] = L.pair ? 2 * off : off; }
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
            L.storage = new_storage(L.pair ? pair_width(L.C) : roundup(L.C, gran_of(L.store_dt)), L.store_dt, (size_t)c->max_batch * L.H * L.W, c->keep_layers); L.ch_off = 0;
        }
    }
    for (int i = 0; i < NL; ++i) {
        Layer &L = c->layers[i];
        if (L.type == L_YOLO || L.type == L_REGION || L.type == L_DETECT) { L.noop = true; L.storage = c->layers[i - 1].storage; L.ch_off = c->layers[i - 1].ch_off; continue; }
        if (L.type == L_ROUTE && L.in.size() == 1) { L.noop = true; int j = L.in[0]; if (j < 0) return fail(c, YOLO_ERR_UNSUPPORTED, "route to network input"); L.storage = c->layers[j].storage; L.ch_off = c->layers[j].ch_off; continue; }
        if (L.type == L_ROUTE) continue;
        if (L.stem_skip || L.pstem_skip) { L.noop = true; continue; }               // lives in LDS only
        if (place_route[i] >= 0) { L.storage = c->layers[place_route[i]].storage; L.ch_off = place_off[i]; }
        else if (L.head) L.storage = new_storage(roundup(L.C, 4), DT_F32, (size_t)c->max_batch * L.H * L.W, true);
        else L.storage = new_storage(L.pair ? pair_width(L.C) : roundup(L.C, gran_of(L.store_dt)), L.store_dt, (size_t)c->max_batch * L.H * L.W, c->keep_layers);
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
    c->input.dt = c->dtype == YOLO_FP32 ? DT_F32 : (c->dtype == YOLO_FP16 || c->split()) ? DT_F16 : DT_BF16;            // fp8 mode keeps the image in bf16
    const int in_stride = c->in_pair ? 24 : 8;                 // split fp16: hi | lo | hi blocks of the 8 padded channels
    size_t in_bytes = (size_t)c->max_batch * c->in_h * c->in_w * in_stride * dt_size(c->input.dt);
    HIPCK(c, hipMalloc(&c->input.ptr, in_bytes)); HIPCK(c, hipMemsetAsync(c->input.ptr, 0, in_bytes, c->stream));      // defined even if a timing pass runs before any image was staged
    c->input.n = c->max_batch; c->input.h = c->in_h; c->input.w = c->in_w; c->input.c = 8; c->input.stride = in_stride;
    if (c->split()) {
        // fp32 staging for the layers that run in fp32 between a join and a split (input conversion, upsample, pooling, reorg)
        size_t cap = (size_t)c->max_batch * c->in_h * c->in_w * 8;
        for (auto &L : c->layers) if (L.type == L_UPSAMPLE || L.type == L_MAXPOOL || L.type == L_REORG) {
            const TView in = view_of(c, L.in[0]);
            cap = std::max(cap, (size_t)c->max_batch * in.h * in.w * roundup(in.c, 32));
            cap = std::max(cap, (size_t)c->max_batch * L.H * L.W * roundup(L.C, 32));
        }
        c->f32_cap = cap;
        HIPCK(c, hipMalloc((void **)&c->d_f32a, cap * 4)); HIPCK(c, hipMalloc((void **)&c->d_f32b, cap * 4));
    }
    HIPCK(c, hipMalloc(&c->d_zeros, 4096)); HIPCK(c, hipMemsetAsync(c->d_zeros, 0, 4096, c->stream));
    for (size_t i = 0; i < c->layers.size(); ++i) {
        const Layer &L = c->layers[i];
        if (L.s2d7 && c->split()) return fail(c, YOLO_ERR_UNSUPPORTED, "layer %zu: a 7x7 / stride 2 first conv is not served in the split-fp16 configuration", i);
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
    // The 16-bit / e4m3 conv kernels address their input with 32-bit buffer offsets: the whole-batch activation window of every conv must
    // stay under 2 GiB (launch_conv_bf16's own check).  Refuse a max_batch that cannot run HERE, with the number that can, instead of a bare
    // 'invalid value' from the first forward (a split-fp16 tensor is 2 x as wide, so 416 x 416 stops above batch 96).
    if (c->dtype != YOLO_FP32)
        for (size_t i = 0; i < c->layers.size(); ++i) {
            const Layer &L = c->layers[i];
            if (L.type != L_CONV || L.fc || L.s2d7 || L.stem_skip || L.stem || L.stem_tail || L.blk_skip || L.pstem_skip || L.pstem) continue;     // (fused layers: their launch checks its own windows)
            const TView in = view_of(c, L.in[0]);
            const double per_image = (double)in.h * in.w * in.stride * dt_size(L.in_dt), slack = 2.0 * (in.w + 1) * in.stride * dt_size(L.in_dt);
            if (per_image * c->max_batch + slack >= 2147483648.0)
                return fail(c, YOLO_ERR_UNSUPPORTED, "max_batch %d: layer %zu reads %.0f bytes per image%s, and a conv's whole-batch input must stay below 2 GiB (32-bit buffer offsets): at most %d images per context",
                            c->max_batch, i, per_image, c->split() ? " (split fp16 pairs: 2 x the channels)" : "", (int)((2147483648.0 - slack - 1) / per_image));
        }
    return YOLO_OK;
}

bool fixed_kernel(const Layer &L) { return L.stem || L.stem_skip || L.stem_tail || L.halo || L.s2 || L.blk || L.blk_skip || L.pstem || L.pstem_skip; }

}  // namespace yolo_impl
