// Launch sequence of libyolo_hip.so: layer -> kernel launch, input staging, threshold + NMS, the captured detect step, layer timing and
// the in-situ tile autotuner.
#include "yolo_ctx.h"

namespace yolo_impl {

int fail(yolo_ctx *c, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (c) c->err = buf;
    return code;
}

ConvArgs conv_args(const yolo_ctx *c, const Layer &L, int n)
{
    ConvArgs a; memset(&a, 0, sizeof a);
    TView in = view_of(c, L.in[0]);
    a.in = in.ptr; a.in_stride = in.stride; a.wt = L.d_w; a.bias = L.d_b;
    a.out = L.out.ptr; a.out_stride = L.out.stride; a.out_dt = L.out.dt; a.in_dt = L.in_dt; a.oscale = L.d_sc;
    // split fp16: a conv that writes PAIRS leaves through the SPLIT epilogue (interleaved hi | lo groups); the fp32 heads of such a network keep
    // running on the split instantiations too (their tile table); a conv that writes plain fp16 (mixed plans) has the ordinary fp16 epilogue
    if (L.pair || (c->split() && L.out.dt == DT_F32)) a.split = 1;
    a.pairk = L.in[0] >= 0 && c->pair_of(L.in[0]) ? 1 : 0;          // the input is an interleaved pair tensor: the pair K loop (the network input's three blocks: the plain one)
    a.out_inv_scale = 1.f; a.res_scale = 1.f; a.mid_scale = 1.f; a.mid_inv_scale = 1.f;
    const int li = (int)(&L - c->layers.data());
    if (L.residual_from >= -1) { TView r = view_of(c, L.residual_from); a.res = r.ptr; a.res_stride = r.stride; }
    // the tail runs on the producer's operand type: bf16 needs the fragment-order copy of the 1x1 filters (tail_fragments), e4m3 an
    // e4m3-packed 1x1 conv; anything else leaves w2 null and run_layer refuses the plan instead of launching with a null w2f
    if (L.tail_on && L.tail_layer >= 0 && c->layers[L.tail_layer].in_dt == L.in_dt && (L.in_dt == DT_FP8 || c->layers[L.tail_layer].d_wf)) {
        const Layer &T = c->layers[L.tail_layer];
        a.w2 = T.d_w; a.w2f = T.d_wf; a.b2 = T.d_b; a.out2 = T.out.ptr; a.out2_stride = T.out.stride; a.K2pad = T.kpad; a.act2 = T.act;
        a.oscale2 = T.d_sc; a.out2_inv_scale = T.out.dt == DT_FP8 ? 1.f / c->eff_scale[L.tail_layer] : 1.f;
        if (T.head) {            // the tail is a detection head: fp32 rows + the objectness plane of the [yolo] layer behind it
            a.tail_f32 = 1; a.C2out = T.filters;
            if (T.d_obj && L.tail_layer + 1 < (int)c->layers.size()) { const Layer &Y = c->layers[L.tail_layer + 1]; a.obj_out = T.d_obj; a.obj_attrs = 5 + Y.classes; a.obj_na = Y.na; conv_magic((uint32_t)a.obj_attrs, a.obj_mul, a.obj_shift); }
        }
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

// split fp16: the untuned choice among the instantiated shapes (conv_cfg_split_ok)
static int split_default_cfg(const ConvArgs &a)
{
    // the image layer: the direct pair kernel wherever it applies -- and ONLY it (its K grouping differs from the tiled kernel's: were both
    // selectable, a tuned plan would no longer be the built-in plan bit for bit)
    if (conv_c8_direct_pair_ok(a)) return CONV_CFG_DIRECT;
    const long M = (long)a.N * a.Ho * a.Wo;
    if (a.Cout <= 32) return 4;
    if (a.Cout <= 64) return M >= 65536 ? 8 : 6;
    const long tiles128 = ((M + 127) / 128) * ((a.Cout + 127) / 128);
    if (tiles128 < 512) return M < 8192 && tiles128 < 128 ? 14 : 2;
    return 0;
}

// split fp16: a layer that moves or interpolates values runs in fp32 between a join (hi + lo) and a split
static int via_f32(yolo_ctx *c, const Layer &L, int n, int kind)
{
    hipStream_t s = c->stream;
    const TView in = view_of(c, L.in[0]);
    const int cpi = roundup(in.c, 32), cpo = roundup(L.C, 32);
    const size_t pin = (size_t)n * in.h * in.w, pout = (size_t)n * L.H * L.W;
    if (pin * cpi > c->f32_cap || pout * cpo > c->f32_cap) return fail(c, YOLO_ERR_STATE, "internal: fp32 staging too small");
    HIPCK(c, launch_split_to_f32(in.ptr, in.stride, cpi, c->d_f32a, cpi, pin, s));
    TView a; a.ptr = c->d_f32a; a.n = n; a.h = in.h; a.w = in.w; a.c = cpi; a.stride = cpi; a.dt = DT_F32;
    TView b; b.ptr = c->d_f32b; b.n = n; b.h = L.H; b.w = L.W; b.c = cpo; b.stride = cpo; b.dt = DT_F32;
    if (kind == 0) HIPCK(c, launch_upsample2x(a, b, c->semantics == YOLO_SEM_TF, s));
    else if (kind == 1) HIPCK(c, launch_maxpool(a, b, L.psize, L.pstride, L.ppad, s));
    else {          // reorg scrambles channels: it runs on the LOGICAL channel counts (the padding of a pair tensor is not part of it)
        a.c = in.c; b.c = L.C;
        if (in.c % 32 || L.C % 32) return fail(c, YOLO_ERR_UNSUPPORTED, "reorg of a pair tensor of %d channels (whole 32-channel groups only)", in.c);
        HIPCK(c, launch_reorg(a, b, L.pstride, c->semantics == YOLO_SEM_DARKNET, s));
    }
    HIPCK(c, launch_split_from_f32(c->d_f32b, cpo, L.out.ptr, L.out.stride, cpo, pout, s));
    return YOLO_OK;
}

int run_layer(yolo_ctx *c, int i, int n)
{
    Layer &L = c->layers[i];
    hipStream_t s = c->stream;
    auto nview = [&](TView v) { v.n = n; return v; };
    switch (L.type) {
    case L_CONV: {
        if (L.stem_skip || L.stem_tail || L.pstem_skip) break;
        if (L.pstem) {           // split-fp16: conv0 + conv1 in one launch (conv_stem_pair.hip)
            const Layer &A = c->layers[0];
            StemPairArgs t; memset(&t, 0, sizeof t);
            t.in = c->input.ptr; t.w0 = A.d_w; t.b0 = A.d_b; t.Kpad0 = A.kpad; t.C0 = A.filters; t.act0 = A.act;
            t.in_u8 = c->stem_u8; t.in_scale = c->stem_scale; t.in_mul = c->in_mul; t.in_add = c->in_add;
            t.w1 = L.d_w; t.b1 = L.d_b; t.Kpad1 = L.kpad; t.act1 = L.act;
            t.out = L.out.ptr; t.out_stride = L.out.stride; t.N = n; t.H = A.H; t.W = A.W; t.Ho = L.H; t.Wo = L.W;
            if (c->input.stride != 24 || !conv_stem_pair_ok(t)) return fail(c, YOLO_ERR_STATE, "layer %d: the fused split-fp16 stem does not apply to this plan", i);
            HIPCK(c, launch_conv_stem_pair(t, s));
            break;
        }
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
            if (b.C == 64 ? conv_resblock64_ok(b) : conv_resblock_ok(b)) {
                if (L.blk) HIPCK(c, b.C == 64 ? launch_conv_resblock64(b, s) : launch_conv_resblock(b, s));
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
        if (L.s2) {            // 3x3/s2 64 -> 128: window staged once in LDS, filters in registers (conv_s2.hip)
            HaloArgs h; memset(&h, 0, sizeof h);
            const TView in = view_of(c, L.in[0]);
            h.in = a.in; h.in_stride = a.in_stride; h.w = a.wt; h.b = a.bias; h.Kpad = a.Kpad; h.Cin = L.cin; h.Cout = L.filters; h.act = L.act;
            h.res = a.res; h.out = a.out; h.out_stride = a.out_stride; h.N = n; h.H = in.h; h.W = in.w; h.dt = L.in_dt;
            if (conv_s2_ok(h)) { HIPCK(c, launch_conv_s2(h, s)); break; }
        }
        if (a.split || a.pairk) {
            auto inst = [&](int cfg) { return a.pairk ? conv_cfg_pairk_ok(cfg, a.split != 0) : conv_cfg_split_ok(cfg); };
            int cfg = L.tile_cfg >= 0 && inst(L.tile_cfg) ? L.tile_cfg : split_default_cfg(a);
            if (conv_c8_direct_pair_ok(a)) cfg = CONV_CFG_DIRECT;          // the image layer: the direct pair kernel, whatever the plan says
            if (conv_cfg_is_halo(cfg) && (!conv_halo_cfg_ok(a, cfg) || a.out_dt == DT_F32)) cfg = split_default_cfg(a);
            HIPCK(c, launch_conv_bf16(a, cfg, s));
        }
        else if (c->dtype == YOLO_FP32) { HIPCK(c, launch_conv_f32(a, s)); }
        else if (L.in_dt == DT_FP8) {
            int cfg = L.tile_cfg >= 0 && conv_cfg_fp8_ok(L.tile_cfg) ? L.tile_cfg : conv_pick_cfg(a);
            if (conv_cfg_is_halo(cfg) && !conv_halo_cfg_ok(a, cfg)) cfg = conv_pick_cfg(a);      // e.g. a smaller batch window or another input size
            if (a.w2 && !conv_cfg_tail_ok(cfg, a.Cout, a.in_dt == DT_FP8, a.tail_f32 != 0)) return fail(c, YOLO_ERR_STATE, "layer %d: tile config %d cannot run the fused 1x1 tail", i, cfg);
            HIPCK(c, launch_conv_fp8(a, cfg, s));
        } else {
            int cfg = L.tile_cfg >= 0 ? L.tile_cfg : conv_pick_cfg(a);
            if (cfg == CONV_CFG_DIRECT && !conv_c8_direct_ok(a)) cfg = conv_pick_cfg(a);
            if (conv_cfg_is_halo(cfg) && !conv_halo_cfg_ok(a, cfg)) cfg = conv_pick_cfg(a);
            if (a.w2 && !conv_cfg_tail_ok(cfg, a.Cout, a.in_dt == DT_FP8, a.tail_f32 != 0)) return fail(c, YOLO_ERR_STATE, "layer %d: tile config %d cannot run the fused 1x1 tail", i, cfg);
            HIPCK(c, launch_conv_bf16(a, cfg, s));
        }
        break; }
    case L_SHORTCUT:
        if (!L.noop && L.pair) {
            HIPCK(c, launch_add_split(view_of(c, L.in[0]).ptr, view_of(c, L.in[0]).stride, view_of(c, L.in[1]).ptr, view_of(c, L.in[1]).stride, L.out.ptr, L.out.stride, roundup(L.C, 32), (size_t)n * L.H * L.W, s));
        } else if (!L.noop) {
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
            if (L.pair) {        // interleaved pairs: the source's 2 * Cp elements at twice the channel offset (whole 32-channel groups)
                if (src.c % 32 || L.copy_offsets[k] % 32) return fail(c, YOLO_ERR_UNSUPPORTED, "route copy of a pair tensor: %d channels at offset %d (whole 32-channel groups only)", src.c, L.copy_offsets[k]);
                TView dv = nview(L.out);
                dv.ptr = (char *)L.out.ptr + (size_t)2 * L.copy_offsets[k] * 2; dv.c = 2 * src.c; src.c = 2 * src.c;
                HIPCK(c, launch_copy(src, dv, s));
                continue;
            }
            HIPCK(c, launch_copy(src, dst, s));
        }
        break;
    case L_LOCAL: HIPCK(c, launch_local(nview(view_of(c, L.in[0])), nview(L.out), L.d_w, L.d_b, L.size, L.stride, L.pad, L.act, s)); break;
    case L_UPSAMPLE: if (L.pair) { if (getenv("YOLO_PAIR_UPSAMPLE_VIA_F32")) { if (int r = via_f32(c, L, n, 0)) return r; } else HIPCK(c, launch_upsample2x_pair(nview(view_of(c, L.in[0])), nview(L.out), c->semantics == YOLO_SEM_TF, s)); break; } HIPCK(c, launch_upsample2x(nview(view_of(c, L.in[0])), nview(L.out), c->semantics == YOLO_SEM_TF, s)); break;
    case L_MAXPOOL: if (L.pair) { if (int r = via_f32(c, L, n, 1)) return r; break; } HIPCK(c, launch_maxpool(nview(view_of(c, L.in[0])), nview(L.out), L.psize, L.pstride, L.ppad, s)); break;
    case L_REORG: if (L.pair) { if (int r = via_f32(c, L, n, 2)) return r; break; } HIPCK(c, launch_reorg(nview(view_of(c, L.in[0])), nview(L.out), L.pstride, c->semantics == YOLO_SEM_DARKNET, s)); break;
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
    if (fmt == YOLO_IMG_U8 && c->layers.size() > 1 && (c->layers[1].stem || c->layers[1].pstem) && !getenv("YOLO_NO_STEM_U8") && (double)npix * 3 < 2147483648.0 && ((size_t)src & 3) == 0) {
        c->stem_u8 = (const uint8_t *)src; c->stem_scale = scale; c->stem_u8_n = n;
        return YOLO_OK;
    }
    if (c->in_pair) {        // the image in fp32 (exact for uint8 pixels x scale up to fp32 rounding), then split pairs
        HIPCK(c, launch_preprocess(src, fmt, n, c->in_h * c->in_w, scale, c->d_f32a, DT_F32, 8, c->stream, c->in_mul, c->in_add));
        HIPCK(c, launch_split_from_f32(c->d_f32a, 8, c->input.ptr, 24, 8, npix, c->stream, PAIR_B3));
        return YOLO_OK;
    }
    HIPCK(c, launch_preprocess(src, fmt, n, c->in_h * c->in_w, scale, c->input.ptr, c->input.dt, 8, c->stream, c->in_mul, c->in_add));
    return YOLO_OK;
}

// ---- image windows over the first layers (round 6) -------------------------------------------------------------------------------
// The first layers of a darknet move tensors far larger than the 256 MiB Infinity Cache at batch 32 (YOLOv3-416: 709 MB out of the
// first conv as split-fp16 pairs, 177-354 MB per tensor down to the 104 x 104 stage; half of that in bf16) and run at their HBM floors:
// a tensor is written to memory by one launch and read back from memory by the next.  Images are independent, so the same launches can
// walk the batch in WINDOWS of `b` images -- layers 0..E on images [0, b), then on [b, 2b), ... -- with b chosen so that what a window's
// layers touch fits the cache: a tensor is then still resident when its consumer reads it (8.6 TB/s against ~4-6 from HBM,
// MI355X_MICROARCH.md 'Infinity Cache') and the pooled buffers are overwritten by the next window while resident.  Same kernels, same
// per-image arithmetic (every tile walks K in the same order whatever the batch: tests/test_gpu_network.py batch independence), E * n / b
// launches instead of E.  The window is applied by shifting the views the launch code reads (layer outputs, the network input, the
// staged uint8 batch) and running the layers with n = b; nothing below run_layer knows about it.
struct WindowPlan { int last = -1, b = 0; };            // layers 0..last run in windows of b images (last < 0: no windows)
static WindowPlan window_plan(const yolo_ctx *c, int n)
{
    WindowPlan w;
    // MEASURED (round 6, same-box A/B, tools/probe/ab_windows*.sh): it LOSES -- bf16 416 b32 13.55 -> 12.99 k img/s with windows of 8 images
    // over the fused first layers, split-fp16 5.10 -> 4.90 k with windows of 4 over layers 0-11, and every (last, b) variant tried sits
    // between: each extra launch costs its ~4-8 us and the consumer does not read faster.  OFF unless YOLO_WINDOWS asks for it ("auto", or
    // "last,b" for a probe); the schedule stays as a knob for other topologies / batch sizes.
    const char *e = getenv("YOLO_WINDOWS");
    if (!e || c->keep_layers || c->dtype == YOLO_FP32 || n < 4) return w;
    if (strcmp(e, "auto") != 0) {         // "last,b": probes
        int l = -1, b = 0;
        if (sscanf(e, "%d,%d", &l, &b) == 2 && l >= 0 && l < (int)c->layers.size() && b >= 1 && b < n) { w.last = l; w.b = b; }
        return w;
    }
    const double cache = 256.0 * 1048576.0;
    const int NL = (int)c->layers.size();
    auto vbytes = [&](const TView &v) { return (double)v.h * v.w * v.stride * dt_size(v.dt); };
    // per-image bytes each launch of the plain prefix (convs and the shortcuts folded into them) reads and writes in MATERIALISED tensors
    int prefix = -1;
    std::vector<double> need(NL, 0.0);
    for (int i = 0; i < NL; ++i) {
        const Layer &L = c->layers[i];
        if (L.type == L_SHORTCUT && L.noop) { prefix = i; continue; }
        if (L.type != L_CONV || L.fc || L.s2d7 || L.head || L.tail_on || L.fused_into >= 0) break;
        double bytes = 0;
        if (L.stem_skip || L.stem_tail || L.blk_skip || L.pstem_skip) bytes = 0;                                   // computed inside a neighbour's launch
        else if (L.stem) bytes = (double)c->in_h * c->in_w * 3 + vbytes(L.out) + (i + 1 < NL && c->layers[i + 1].stem_tail ? vbytes(c->layers[i + 1].out) : 0.0);
        else if (L.blk) bytes = 2.0 * vbytes(view_of(c, c->layers[i - 1].in[0])) + vbytes(L.out) - vbytes(view_of(c, c->layers[i - 1].in[0]));      // x in (shortcut from L2), y out
        else {
            bytes = vbytes(view_of(c, L.in[0])) + vbytes(L.out);
            if (L.residual_from >= -1) bytes += vbytes(view_of(c, L.residual_from));
        }
        need[i] = bytes; prefix = i;
    }
    // ... as far as a launch's whole-batch traffic exceeds most of the cache (beyond that point tensors already live in it)
    int last = -1;
    for (int i = 0; i <= prefix; ++i) if (need[i] * n > 0.7 * cache) last = i;
    if (last < 0) return w;
    double biggest = 0;
    for (int i = 0; i <= last; ++i) biggest = std::max(biggest, need[i]);
    int b = n;
    while (b > 1 && biggest * b > 0.55 * cache) b = (b + 1) / 2;
    if (b >= n || b < 2) return w;
    // a window must still fill the chip: the tiled layers need about one tile per CU (the fused first-layer kernels are persistent)
    for (int i = 0; i <= last; ++i) {
        const Layer &L = c->layers[i];
        if (L.type != L_CONV || fixed_kernel(L) || need[i] == 0) continue;
        const double tiles = std::ceil((double)b * L.H * L.W / 176.0) * std::ceil(L.filters / 128.0);
        if (tiles < 224) { last = i - 1; break; }
    }
    while (last >= 0 && (c->layers[last].blk_skip || c->layers[last].stem_skip)) --last;          // never split a fused group
    while (last + 1 <= prefix && ((c->layers[last + 1].type == L_SHORTCUT && c->layers[last + 1].noop) || c->layers[last + 1].stem_tail)) ++last;
    if (last < 0) return w;
    w.last = last; w.b = b;
    return w;
}
// shift every view layers 0..last (and the input) by `img` images (sign: +1 apply, -1 undo)
static void window_shift(yolo_ctx *c, int last, long img)
{
    auto adv = [&](TView &v) { if (v.ptr) v.ptr = (char *)v.ptr + img * (long)v.h * v.w * v.stride * (long)dt_size(v.dt); };
    adv(c->input);
    for (int i = 0; i <= last; ++i) adv(c->layers[i].out);
    if (c->stem_u8) c->stem_u8 += img * (long)c->in_h * c->in_w * 3;
}
// the layer sequence of one forward: windows over the first layers, then the rest on the whole batch.  `skip_conv`: the timing pass that
// runs everything but the convs; `ev` / `acc_ms`: per-layer events (yolo_time_layers)
static int run_layers(yolo_ctx *c, int n, bool skip_conv = false, std::vector<hipEvent_t> *ev = nullptr)
{
    const int NL = (int)c->layers.size();
    const WindowPlan w = window_plan(c, n);
    int first_whole = 0;
    if (w.last >= 0 && !ev) {
        for (int img = 0; img < n; img += w.b) {
            const int nb = std::min(w.b, n - img);
            window_shift(c, w.last, img);
            int r = YOLO_OK;
            for (int i = 0; i <= w.last && r == YOLO_OK; ++i) {
                if (skip_conv && c->layers[i].type == L_CONV) continue;
                r = run_layer(c, i, nb);
            }
            window_shift(c, w.last, -img);
            if (r) return r;
        }
        first_whole = w.last + 1;
    }
    for (int i = first_whole; i < NL; ++i) {
        if (!(skip_conv && c->layers[i].type == L_CONV)) { int r = run_layer(c, i, n); if (r) return r; }
        if (ev) HIPCK(c, hipEventRecord((*ev)[i + 1], c->stream));
    }
    return YOLO_OK;
}

int run_network(yolo_ctx *c, int n, bool lean)
{
    c->lean = lean && c->lean_ok;
    { int r = run_layers(c, n); if (r) { c->lean = false; return r; } }
    c->last_n = n; c->scores_mode = 0; c->det_valid = !c->lean;
    return YOLO_OK;
}

int post_args_ok(yolo_ctx *c, int max_out, int nms_mode, int select_mode)
{
    if (max_out < 1) return fail(c, YOLO_ERR_INVALID, "max_out < 1");
    if (nms_mode < 0 || nms_mode > 4 || select_mode < 0 || select_mode > 1) return fail(c, YOLO_ERR_INVALID, "bad nms/select mode");
    return YOLO_OK;
}

int copy_out(yolo_ctx *c, void *dst, const void *src, size_t bytes, int loc)
{
    if (loc == YOLO_HOST) { HIPCK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream)); HIPCK(c, hipStreamSynchronize(c->stream)); }
    else HIPCK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->stream));
    return YOLO_OK;
}

int post(yolo_ctx *c, const float *det, int n, int rows, int attrs, float score_thr, float iou_thr, int max_out,
         int nms_mode, int select_mode, int img_h, int img_w, int scores_ready, yolo_box *boxes_out, int32_t *counts_out, int out_loc, int32_t *rows_out)
{
    if (int r = post_args_ok(c, max_out, nms_mode, select_mode)) return r;
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

}  // namespace yolo_impl

extern "C" {

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
    hipError_t e = c->in_pair ? launch_resize_u8(src, h, w, c->in_h, c->d_f32a, DT_F32, 8, 8, c->stream, c->in_mul, c->in_add)
                              : launch_resize_u8(src, h, w, c->in_h, c->input.ptr, c->input.dt, 8, 8, c->stream, c->in_mul, c->in_add);
    if (e == hipSuccess && c->in_pair) e = launch_split_from_f32(c->d_f32a, 8, c->input.ptr, 24, 8, (size_t)c->in_h * c->in_w, c->stream, PAIR_B3);
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
    hipError_t e = c->in_pair ? launch_letterbox_chw(src, w, h, c->in_h, c->d_f32a, DT_F32, 8, c->stream) : launch_letterbox_chw(src, w, h, c->in_h, c->input.ptr, c->input.dt, 8, c->stream);
    if (e == hipSuccess && c->in_pair) e = launch_split_from_f32(c->d_f32a, 8, c->input.ptr, 24, 8, (size_t)c->in_h * c->in_w, c->stream, PAIR_B3);
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
    if (!c) return YOLO_ERR_INVALID;
    if (int r = post_args_ok(c, max_out, nms_mode, select_mode)) return r;      // before the forward: a decode whose NMS never runs leaves the lean list counter set
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
    if (int r = post_args_ok(c, max_out, nms_mode, select_mode)) return r;
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
    // the captured step holds no reset of the lean decode's list counter (its NMS node leaves it at zero): a step that failed between
    // its decode and its NMS since then left it set
    if (c->lean_cnt_dirty) { HIPCK(c, hipMemsetAsync(c->d_lean_cnt, 0, 16, c->stream)); c->lean_cnt_dirty = false; }
    HIPCK(c, hipGraphLaunch(c->gexec, c->stream));
    // the replay leaves the context exactly as the eager call it was captured from would (run_network / forward_impl): a later
    // yolo_postprocess / yolo_darknet_boxes must see that the decoded tensor was (not) written and which threshold pruned the scores
    c->lean = nms_mode != YOLO_NMS_NUMPY_V3 && c->lean_ok; c->det_valid = !c->lean; c->lean_thr = score_thr;
    c->last_n = n; c->scores_mode = nms_mode == YOLO_NMS_NUMPY_V3 ? 1 : 0;
    return YOLO_OK;
}

// the timing entry points' events: destroyed on every return path
namespace { struct EventSet {
    std::vector<hipEvent_t> e; bool ok = true;
    explicit EventSet(int n) { e.reserve(n); for (int i = 0; i < n; ++i) { hipEvent_t x; if (hipEventCreate(&x) != hipSuccess) { ok = false; break; } e.push_back(x); } }
    ~EventSet() { for (auto x : e) hipEventDestroy(x); }
    EventSet(const EventSet &) = delete; EventSet &operator=(const EventSet &) = delete;
}; }

int yolo_time_forward(yolo_ctx *c, int n, int iters, float *total_ms, float *conv_ms)
{
    if (!c) return YOLO_ERR_INVALID;
    if (!c->weights_loaded) return fail(c, YOLO_ERR_STATE, "weights not loaded");
    if (n < 1 || n > c->max_batch || iters < 1) return fail(c, YOLO_ERR_INVALID, "bad n/iters");
    HIPCK(c, hipSetDevice(c->device));
    EventSet evs(2); if (!evs.ok) return fail(c, YOLO_ERR_HIP, "hipEventCreate failed");
    hipEvent_t e0 = evs.e[0], e1 = evs.e[1];
    c->lean = false;
    if (c->stem_u8 && n > c->stem_u8_n) c->stem_u8 = nullptr;      // the staged uint8 buffer holds fewer images: the stem reads c->input instead
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
            for (int it = 0; it < iters; ++it) { int r = run_layers(c, n, pass == 1); if (r) return r; }        // (the forward's own schedule, image windows included)
            HIPCK(c, hipEventRecord(e1, c->stream)); HIPCK(c, hipEventSynchronize(e1));
            float ms = 0; HIPCK(c, hipEventElapsedTime(&ms, e0, e1)); (pass == 0 ? all_ms : rest_ms) = ms / iters;
        }
        *conv_ms = all_ms - rest_ms;
        c->last_n = n; c->scores_mode = 0; c->det_valid = true;
    }
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
    if (c->stem_u8 && n > c->stem_u8_n) c->stem_u8 = nullptr;      // see yolo_time_forward
    EventSet evs(NL + 1); if (!evs.ok) return fail(c, YOLO_ERR_HIP, "hipEventCreate failed");
    std::vector<hipEvent_t> &ev = evs.e;
    std::vector<double> acc(NL, 0.0);
    for (int it = 0; it < iters; ++it) {
        HIPCK(c, hipEventRecord(ev[0], c->stream));
        for (int i = 0; i < NL; ++i) { int r = run_layer(c, i, n); if (r) return r; HIPCK(c, hipEventRecord(ev[i + 1], c->stream)); }
        HIPCK(c, hipStreamSynchronize(c->stream));
        for (int i = 0; i < NL; ++i) { float ms = 0; HIPCK(c, hipEventElapsedTime(&ms, ev[i], ev[i + 1])); acc[i] += ms; }
    }
    for (int i = 0; i < NL; ++i) ms_out[i] = (float)(acc[i] / iters);
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
        char key[128]; snprintf(key, sizeof key, "%d_%d_%d_%d_%d_%d_%d%d_%d%d%d", a.H, a.W, a.Cin_pad, a.Cout, a.ksize, a.stride, a.in_dt, a.out_dt, a.res != nullptr, a.split, a.pairk);
        return std::string(key);
    };
    auto valid = [&](const Layer &L, int cfg) {
        if (fixed_kernel(L)) return false;                     // fused stem: nothing to choose
        ConvArgs a = conv_args(c, L, n);
        if (a.split && conv_c8_direct_pair_ok(a)) return cfg == CONV_CFG_DIRECT;
        if (cfg == CONV_CFG_DIRECT) return !a.split && conv_c8_direct_ok(a);
        if (a.pairk) return conv_cfg_pairk_ok(cfg, a.split != 0) && (!conv_cfg_is_halo(cfg) || (conv_halo_cfg_ok(a, cfg) && a.out_dt != DT_F32));
        if (a.split) return conv_cfg_split_ok(cfg) && (!conv_cfg_is_halo(cfg) || (conv_halo_cfg_ok(a, cfg) && a.out_dt != DT_F32));
        if (a.in_dt == DT_FP8) return conv_cfg_fp8_ok(cfg);
        return true;
    };
    for (auto &L : c->layers) L.tail_on = false;
    std::vector<int> fallback(NL, -1);
    for (int i = 0; i < NL; ++i) if (c->layers[i].type == L_CONV && !fixed_kernel(c->layers[i])) { ConvArgs a = conv_args(c, c->layers[i], n); fallback[i] = (a.split || a.pairk) ? split_default_cfg(a) : conv_pick_cfg(a); }
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
        std::map<std::string, std::pair<double, int>> best;         // producer shape (+ kind of tail) -> (pair time, cfg), cfg -1 = unfused
        auto tail_key = [&](const Layer &L) { return shape_key(L) + (c->layers[L.tail_layer].head ? "_head" : ""); };      // (a head as the tail: another set of configurations can host it)
        for (int i = 0; i < NL; ++i) {
            const Layer &L = c->layers[i];
            if (L.type != L_CONV || L.tail_layer < 0) continue;
            auto &b = best[tail_key(L)];
            if (b.second == 0 && b.first == 0) b = {0.0, -1};
            b.first += base[i] + base[L.tail_layer];
        }
        for (int cfg = 0; cfg < conv_num_cfgs(); ++cfg) {
            bool any = false;
            for (int i = 0; i < NL; ++i) {
                Layer &L = c->layers[i];
                if (L.type != L_CONV || L.tail_layer < 0) continue;
                bool ok = conv_cfg_tail_ok(cfg, L.filters, L.in_dt == DT_FP8, c->layers[L.tail_layer].head) && c->layers[L.tail_layer].in_dt == L.in_dt && valid(L, cfg);      // (valid: e.g. a shape the e4m3 table does not instantiate)
                if (ok && conv_cfg_is_halo(cfg)) { ConvArgs a = conv_args(c, L, n); ok = conv_halo_cfg_ok(a, cfg); }
                L.tile_cfg = ok ? cfg : base_cfg[i]; L.tail_on = ok; any |= ok;
            }
            if (!any) continue;
            r = yolo_time_layers(c, n, iters, ms.data()); if (r) return r;
            std::map<std::string, double> t;
            for (int i = 0; i < NL; ++i) {
                const Layer &L = c->layers[i];
                if (L.type == L_CONV && L.tail_layer >= 0 && L.tail_on) t[tail_key(L)] += ms[i] + ms[L.tail_layer];
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
            const auto &b = best[tail_key(L)];
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
        if (tail && (c->layers[i].tail_layer < 0 || !conv_cfg_tail_ok(v, c->layers[i].filters, c->layers[i].in_dt == DT_FP8, c->layers[c->layers[i].tail_layer].head) || c->layers[c->layers[i].tail_layer].in_dt != c->layers[i].in_dt))
            return fail(c, YOLO_ERR_INVALID, "layer %zu: plan asks for a fused 1x1 tail this layer / tile config cannot run", i);
        if (tail && fixed_kernel(c->layers[i])) return fail(c, YOLO_ERR_INVALID, "layer %zu runs a fixed kernel (stem / conv3 / block / stride-2): it hosts no fused 1x1 tail", i);
        if (tail && conv_cfg_is_halo(v)) {
            ConvArgs a = conv_args(c, c->layers[i], c->max_batch);
            if (!conv_halo_cfg_ok(a, v)) return fail(c, YOLO_ERR_INVALID, "layer %zu: the halo-staged tile config %d does not apply to this layer, so it cannot carry the fused 1x1 tail", i, v);
        }
        c->layers[i].tile_cfg = v; c->layers[i].tail_on = tail;
    }
    if (c->gexec) { hipGraphExecDestroy(c->gexec); c->gexec = nullptr; } if (c->gstate > 0) c->gstate = 0;
    return YOLO_OK;
}

}  // extern "C"
