// Private to libyolo_hip.so's host side: the planned network (layers, storage pool, context) and the functions its translation
// units share.  yolo_plan.cpp: darknet-cfg parser, planner, buffer pool.  yolo_pack.cpp: BN fold, filter packing, fp8 scales, weight
// stream / export artifact.  yolo_run.cpp: launch sequence, detect graph, timing, tile autotuner.  yolo_api.cpp: create / destroy,
// introspection, darknet-flavoured views.  yolo_ops.cpp: single-operator entry points.
#pragma once
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

namespace yolo_impl {

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
    bool pair = false;                   // YOLO_FP16X2 networks: this layer's output tensor is split-fp16 PAIRS (interleaved per 32-channel group, 2 x the channels); false there:
                                         // plain fp16 (mixed plans, cfg key yolo_pair=0 on a [convolutional] section; layers that move data inherit their operands' form)
    int tile_cfg = -1;
    int residual_from = -2;              // >= -1: fused shortcut source
    bool head = false;                   // conv feeding a yolo/region layer: fp32 output
    float *d_obj = nullptr;              // ... feeding a [yolo] layer (bf16 / fp8 networks): compact plane of its objectness logits [max_batch * H * W][anchors]
    bool stem_skip = false, stem = false;   // fused stem (conv_stem.hip): layer 0 is never materialised, layer 1 launches both
    bool pstem_skip = false, pstem = false; // ... the same fusion in a split-fp16 network (conv_stem_pair.hip): image pairs -> conv1's pairs in one launch
    bool blk_skip = false, blk = false;     // fused residual block (conv_block.hip): this 1x1 conv is computed inside the launch of the 3x3 conv that follows / this 3x3 conv launches both
    bool stem_tail = false;                 // ... and this 1x1 conv (layer 2) is computed by that launch too
    bool halo = false;                      // 3x3/s1, 32 -> 64 channels: halo-staged kernel instead of the tiled one
    bool s2 = false;                        // 3x3/s2, 64 -> 128 channels: window-staged kernel with register-resident filters (conv_s2.hip)
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


}  // namespace yolo_impl
using namespace yolo_impl;

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
    const uint8_t *stem_u8 = nullptr; float stem_scale = 1.f; int stem_u8_n = 0;      // uint8 image (of stem_u8_n images) the fused stem reads itself (no conversion launch), or nullptr: c->input.  May be the CALLER's buffer: only valid for a pass over <= stem_u8_n images while the caller keeps it (yolo_time_*)
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
    // split fp16 storage (YOLO_FP16X2): a layer output of C channels is [pixel][2 * Cp] f16, Cp = roundup(C, 32), interleaved per 32-channel group (32 hi | 32 lo);
    // the network input [pixel][3 * 8]: hi | lo | hi blocks of its 8 padded channels
    bool split() const { return dtype == YOLO_FP16X2; }
    bool in_pair = false;                 // ... the network input is stored as pairs ([net] yolo_pair_input, default 1 in a YOLO_FP16X2 network)
    bool pair_of(int idx) const { return idx < 0 ? in_pair : layers[idx].pair; }      // is tensor `idx` (-1: the input) stored as pairs?
    float *d_f32a = nullptr, *d_f32b = nullptr; size_t f32_cap = 0;      // split mode: fp32 staging of one tensor (input conversion, upsample / pool / reorg run in fp32 between a join and a split)
    int act_dt() const { return dtype == YOLO_FP32 ? DT_F32 : dtype == YOLO_FP8 ? DT_FP8 : (dtype == YOLO_FP16 || dtype == YOLO_FP16X2) ? DT_F16 : DT_BF16; }
    bool half_like() const { return dtype == YOLO_BF16 || dtype == YOLO_FP16; }      // 16-bit storage: the same kernels, the same plan
    int gran() const { return dtype == YOLO_FP8 ? 16 : 8; }            // channel granule = one 16-B piece (8 for fp32 too)
    size_t esize() const { return dt_size(act_dt()); }
};

namespace yolo_impl {

int fail(yolo_ctx *c, int code, const char *fmt, ...);
#define HIPCK(c, expr)                                                                       \
    do { hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) return fail(c, YOLO_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); } while (0)

inline int roundup(int x, int m) { return (x + m - 1) / m * m; }
inline int pair_width(int C) { return 2 * roundup(C, 32); }      // elements per pixel of an interleaved split-fp16 pair tensor of C channels (32 hi | 32 lo per group)
inline int gran_of(int dt) { return dt == DT_FP8 ? 16 : 8; }      // channels per 16-byte piece (8 for fp32 tensors too)
inline void drop_graph(yolo_ctx *c) { if (c->gexec) { hipGraphExecDestroy(c->gexec); c->gexec = nullptr; } if (c->gstate > 0) c->gstate = 0; }      // a plan / parameter change: the captured detect step must not be replayed

// yolo_pack.cpp
uint16_t f2bf(float f);
uint16_t f2h(float f);
uint8_t f2e4m3(float f);
void pack_conv(const Layer &L, const float *bn_or_bias, const float *w_oihw, int wdt, const float *in_scale,
               std::vector<uint8_t> &wbuf, std::vector<float> &bias, std::vector<float> &osc, int semantics = YOLO_SEM_TF, int split = 0);      // split: 0 plain, 1 pair input in three blocks (the image), 2 interleaved pair input
float h2f(uint16_t h);
void resolve_scales(yolo_ctx *c);
void channel_scales(const yolo_ctx *c, int idx, std::vector<float> &out);
int tail_fragments(yolo_ctx *c);
// yolo_plan.cpp
bool parse_cfg(const char *text, std::vector<Section> &out, std::string &err);
int build_plan(yolo_ctx *c, const std::vector<Section> &secs);
int allocate(yolo_ctx *c);
TView view_of(const yolo_ctx *c, int idx);
bool fixed_kernel(const Layer &L);      // layers whose kernel is fixed by a fusion (nothing for the tile tuner to choose)
// yolo_run.cpp
ConvArgs conv_args(const yolo_ctx *c, const Layer &L, int n);
int run_layer(yolo_ctx *c, int i, int n);
int stage_in(yolo_ctx *c, const void *images, int n, int fmt, int loc, float scale);
int run_network(yolo_ctx *c, int n, bool lean = false);
int copy_out(yolo_ctx *c, void *dst, const void *src, size_t bytes, int loc);
int post_args_ok(yolo_ctx *c, int max_out, int nms_mode, int select_mode);
int post(yolo_ctx *c, const float *det, int n, int rows, int attrs, float score_thr, float iou_thr, int max_out,
         int nms_mode, int select_mode, int img_h, int img_w, int scores_ready, yolo_box *boxes_out, int32_t *counts_out, int out_loc, int32_t *rows_out = nullptr);
// yolo_ops.cpp
extern thread_local std::string g_op_err;
TView make_view(void *p, int n, int h, int w, int c, int stride, int dt);

}  // namespace yolo_impl
