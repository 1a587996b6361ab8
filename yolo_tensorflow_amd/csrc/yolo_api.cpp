// C ABI of libyolo_hip.so (include/yolo_hip.h): context life cycle, introspection, and the darknet-flavoured views of the last forward
// that the veneer libdarknet_hip.so is built on.  The rest of the exports live next to what they drive: yolo_pack.cpp (weights,
// artifact), yolo_run.cpp (forward / detect / timing / tile plans), yolo_ops.cpp (single operators).
#include "yolo_ctx.h"

extern "C" {

yolo_ctx *yolo_create(const yolo_config *cfg, char *err, size_t err_len)
{
    auto bail = [&](yolo_ctx *c, const std::string &m) -> yolo_ctx * { if (err && err_len) snprintf(err, err_len, "%s", m.c_str()); if (c) yolo_destroy(c); return nullptr; };
    if (!cfg || cfg->struct_size != sizeof(yolo_config)) return bail(nullptr, "yolo_create: bad yolo_config (struct_size)");
    if (cfg->max_batch < 1) return bail(nullptr, "yolo_create: max_batch < 1");
    if (cfg->dtype != YOLO_BF16 && cfg->dtype != YOLO_FP32 && cfg->dtype != YOLO_FP8 && cfg->dtype != YOLO_FP16 && cfg->dtype != YOLO_FP16X2) return bail(nullptr, "yolo_create: dtype");
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
                    c->d_dn_rec, c->d_dn_src, c->d_dn_count, c->d_dn_last, c->d_box4, c->s2d.ptr, c->d_srow, c->d_rows, c->d_lean_list, c->d_lean_cnt, c->d_f32a, c->d_f32b};
    for (void *p : ptrs) if (p) hipFree(p);
    if (c->gexec) hipGraphExecDestroy(c->gexec);
    if (c->own_stream && c->stream) hipStreamDestroy(c->stream);
    delete c;
}

const char *yolo_last_error(const yolo_ctx *c) { return c ? c->err.c_str() : g_op_err.c_str(); }

size_t yolo_weights_count(const yolo_ctx *c) { return c ? c->weights_count : 0; }

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
        if (!L.stem_skip && !L.blk_skip && !L.pstem_skip) b += (double)n * L.H * L.W * L.filters * oe;
        b += (double)L.filters * L.cin * L.size * L.size * ie;
    }
    return b;
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
    hipError_t e = hipSuccess;
    float *wide = nullptr;
    if (L.pair && v.dt == DT_F16) {        // split pairs: join into a Cp-strided fp32 image first, then gather the C logical channels
        const int cp = roundup(L.C, 32);
        e = hipMalloc((void **)&wide, (size_t)n * L.H * L.W * cp * 4);
        if (e == hipSuccess) e = launch_split_to_f32(v.ptr, v.stride, cp, wide, cp, (size_t)n * L.H * L.W, c->stream);
        v.ptr = wide; v.stride = cp; v.dt = DT_F32;
    }
    if (e == hipSuccess) e = launch_to_f32(v, tmp, c->stream, vs);
    if (e == hipSuccess) e = hipMemcpyAsync(out, tmp, need * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(tmp); if (wide) hipFree(wide);
    if (e != hipSuccess) return fail(c, YOLO_ERR_HIP, "layer_output: %s", hipGetErrorString(e));
    return YOLO_OK;
}

}  // extern "C"
