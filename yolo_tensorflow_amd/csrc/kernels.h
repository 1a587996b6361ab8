// Internal launcher interface between the host planner (yolo_api.cpp) and the gfx950 kernels.
#pragma once
// Cache policy of the conv kernels' OUTPUT stores (aux operand of buffer_store on gfx950: 0 plain write-back, 2 nt, 16 sc1, 17 sc0 sc1).
// Round 5: sc1 -- write-through to memory as the epilogue runs, instead of leaving up to 32 MB of dirty lines for the end-of-kernel L2
// write-back, which is serial with the next launch.  Same-box A/B, YOLOv3-416 batch 32 bf16 (tools/probe/ab/ab_multi.sh, four interleaved
// rounds): conv stack 2.495 -> 2.367 ms per forward, 12.44 -> 13.08 k img/s; sc0 sc1 the same, sc1 nt worse, nt alone no change.  Only for
// COALESCED stores (whole 128-byte lines per lane group): the fp32 head path, 16 bytes per lane into 64 different rows, stays plain.
#ifndef OUT_STORE_AUX
#define OUT_STORE_AUX 16
#endif
// probe knobs (tools/probe/ab), 0 = default policy in the shipped library: cache policy of the halo-staged form's activation tile loads
// (each byte is read once per workgroup) and of the epilogue's shortcut loads (read once)
#ifndef HALO_LOAD_AUX
#define HALO_LOAD_AUX 0
#endif
#ifndef RES_LOAD_AUX
#define RES_LOAD_AUX 0
#endif
#include <hip/hip_runtime.h>
#include <stdint.h>
// the same policy for a 16-byte store of a kernel that holds no descriptor for its output: a buffer store based at the tensor (`base`
// wave-uniform, byte offset below 2 GiB).  (NOT inline asm: a `global_store ... sc1` written as asm is a vector-memory operation the
// compiler's vmcnt bookkeeping does not see -- its counted waits for the loads around it then wait for one operation too few.  Round 5: the
// fp32 head path written that way returned garbage at batch 32 and passed every small test.)
typedef unsigned out_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void out_store16_at(const void *base, unsigned byte_off, unsigned x, unsigned y, unsigned z, unsigned w)
{
#if defined(__HIP_DEVICE_COMPILE__)
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, 0x80000000u, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(out_u32x4{x, y, z, w}, r, byte_off, 0, OUT_STORE_AUX);
#endif
}

typedef uint16_t bf16_t;   // raw bfloat16 bits in HBM
struct fp8_t { uint8_t b; };   // raw OCP e4m3 (e4m3fn) bits in HBM
struct f16_t { uint16_t b; };  // raw IEEE binary16 bits in HBM
enum { DT_BF16 = 0, DT_F32 = 1, DT_FP8 = 2, DT_F16 = 3 };
inline size_t dt_size(int dt) { return dt == DT_F32 ? 4 : dt == DT_FP8 ? 1 : 2; }
#define FP8_MAX 448.0f

// A view of an NHWC activation tensor living inside a (possibly wider) buffer: pixel p, channel c is
// at ptr[p * stride + c].  Concats are never materialised: producers write into a channel window of
// the wider buffer (DESIGN.md "route/concat").
struct TView {
    void *ptr = nullptr;
    int n = 0, h = 0, w = 0, c = 0;
    int stride = 0;          // elements per pixel of the underlying buffer
    int dt = DT_BF16;        // element type (DT_*)
};

enum { ACT_LINEAR = 0, ACT_LEAKY = 1 };

struct ConvArgs {
    const void *in; int in_stride;       // elements per input pixel; Cin_pad channels are readable
    const void *wt;                      // packed filters [Cout_pad][Kpad], K index = (kh*k+kw)*Cin_pad + c
    const float *bias;                   // [Cout_pad] fp32 (BN folded)
    void *out; int out_stride; int out_dt;
    const void *res; int res_stride;     // residual (same type as out) or nullptr
    // fp8 (e4m3) operands, DESIGN.md "fp8 scheme": value = code * scale.  in_dt selects the MFMA; `oscale` is the
    // per-output-channel dequantisation factor of the accumulator (filter scale; input scales are folded into the
    // filters), `out_inv_scale` = 1 / scale of the output tensor, `res_scale` = scale of the residual tensor.
    int in_dt;
    int split;                           // 1: split fp16 OUTPUT (YOLO_FP16X2): every 16-bit output value is stored as the pair hi | lo, interleaved per 32-channel group
                                         //    (64 bytes of hi, then 64 bytes of lo); out_stride >= 2 * roundup(Cout, 32) elements; the shortcut source (res) has the same form
    int pairk;                           // 1: the INPUT is such an interleaved pair tensor (Cin_pad = 2 * roundup(Cin, 32) elements per tap) and the filters are packed
                                         //    W_hi 32 | W_lo 32 per group: the conv runs the pair K loop (three MFMA products per K-step row pair, conv_igemm_kernel.h PAIRK)
    const float *oscale;                 // [Cout_pad] or nullptr (== 1)
    float out_inv_scale, res_scale;
    float mid_scale, mid_inv_scale;      // fused shortcut: this conv's own output scale (quantised before the add)
    // fused 1x1 tail (bf16, tile configurations with conv_cfg_tail_ok): out2[pixel][C2] = act2(W2 . out[pixel][:] + b2),
    // computed from the finished output tile while it is still in LDS (the 1x1 conv that follows a 3x3 in every darknet
    // residual block).  w2 == nullptr: none.  C2 = Cout / 2, W2 packed [C2 pad][K2pad], k = channel of `out`.
    const void *w2; const float *b2; void *out2; int out2_stride, K2pad, act2;
    const void *w2f;                     // bf16 tail: W2 in MFMA-fragment order [C2 / 16][K2 / 32][64 lanes][8] (yolo_api.cpp tail_fragments)
    const float *oscale2; float out2_inv_scale;      // fp8 tail: per-channel dequantisation scale of W2, 1 / scale of out2
    int N, H, W, Cin_pad;
    int Ho, Wo, Cout;
    int ksize, stride, pad;
    int Kpad;                            // multiple of 64 (bf16) / 128 (fp8) elements
    int kchunk;                          // channels per K-order chunk: k = (chunk * k*k + tap) * kchunk + c (conv_kchunk())
    int act;
    const void *zeros;                   // >= 64 B of zeros in device memory (padding source)
    // n / d for n < 2^31 as mulhi(n, mul) >> shift (shift == 255: d == 1); filled by conv_finalize()
    uint32_t howo_mul, howo_shift, wo_mul, wo_shift;
    // fp32 head convs feeding a [yolo] layer: objectness logits (channel an * obj_attrs + 4 of every pixel) are ALSO written to
    // obj_out[pixel * obj_na + an]; nullptr: off.  obj_mul / obj_shift: division by obj_attrs (conv_magic)
    float *obj_out; int obj_attrs, obj_na; uint32_t obj_mul, obj_shift;
    unsigned long long *dbg;             // diagnostic builds only: per-wave phase cycle sums
    int dbg_light;                       // diagnostic builds only: 1 = stamp once around the K loop and nothing inside it (the in-kernel clock measurement)
    int C2out;                                // ... its real filter count (255)
    int tail_f32;                             // the fused tail is a detection HEAD: C2 = up to 256 filters, fp32 output [pixel][out2_stride] + bias, linear, objectness plane (obj_*)
    // division constants of the tile decode, filled by the launcher for its tile shape (conv_tile_magic): channel tiles per pixel
    // tile; halo form: 13x13 blocks per image and per block row
    uint32_t tc_mul, tc_shift, bpi_mul, bpi_shift, bpr_mul, bpr_shift;
};
// K-order chunk of a conv whose filters are stored with `wdt` elements: one 128-byte LDS row of channels when the padded
// channel count is a multiple of that, else all channels (i.e. plain tap-major order); fp32 filters keep tap-major order
inline int conv_kchunk(int cin_pad, int wdt) { const int row = wdt == DT_FP8 ? 128 : 64; return (wdt != DT_F32 && cin_pad % row == 0) ? row : cin_pad; }
// host helper: derives the division constants from Ho, Wo (call after filling the geometry)
inline void conv_magic(uint32_t d, uint32_t &mul, uint32_t &shift)
{
    if (d <= 1) { mul = 0; shift = 255; return; }
    uint32_t l = 0; while ((1u << l) < d) ++l;          // ceil(log2 d)
    const unsigned k = 31 + l;
    mul = (uint32_t)(((unsigned long long)1 << k) / d + 1);
    shift = k - 32;
}
inline void conv_finalize(ConvArgs &a)
{
    conv_magic((uint32_t)(a.Ho * a.Wo), a.howo_mul, a.howo_shift);
    conv_magic((uint32_t)a.Wo, a.wo_mul, a.wo_shift);
}
// launcher side of the tile decode: BC = output channels per workgroup, bh x bw = block (rows x columns) of the halo form (0: tiled form)
inline ConvArgs conv_tile_magic(const ConvArgs &a0, int BC, int bh, int bw = 0)
{
    ConvArgs a = a0;
    if (bw == 0) bw = bh;
    conv_magic((uint32_t)((a.Cout + BC - 1) / BC), a.tc_mul, a.tc_shift);
    if (bh > 0) { const int br = (a.H + bh - 1) / bh, bc = (a.W + bw - 1) / bw; conv_magic((uint32_t)(br * bc), a.bpi_mul, a.bpi_shift); conv_magic((uint32_t)bc, a.bpr_mul, a.bpr_shift); }      // (ragged edge blocks included)
    return a;
}

// opt a kernel in to more than 64 KiB of dynamic LDS, once per (device, kernel) -- the attribute is per device
hipError_t conv_opt_in_lds(const void *kernel, size_t lds_bytes);
// bf16 MFMA implicit-GEMM conv.  cfg in [0, conv_num_cfgs()); returns hipError.
#define CONV_CFG_DIRECT 1000          // first-layer direct kernel (Cin padded 3 -> 8), outside the tile table
bool conv_c8_direct_ok(const ConvArgs &a);
int conv_num_cfgs();
const char *conv_cfg_name(int cfg);
// rough preference used when no autotune ran
int conv_pick_cfg(const ConvArgs &a);
hipError_t launch_conv_bf16(const ConvArgs &a, int cfg, hipStream_t s);
// halo-staged 3x3 / stride 1 form (conv_halo13.hip): bf16 or fp8 operands, spatial size a multiple of 13, whole 128-byte channel chunks
bool conv_halo13_ok(const ConvArgs &a);                  // the 13 x 13-block halo forms (tile configurations 36..43)
bool conv_halo_cfg_ok(const ConvArgs &a, int cfg);       // ... and the block shape of halo configuration `cfg` (round 5: 10 x 19 and 5 x 19 blocks, 54..56)
bool conv_cfg_is_halo(int cfg);
hipError_t launch_conv_halo13(const ConvArgs &a, int cfg, hipStream_t s);
hipError_t launch_conv_halo13_diag(const ConvArgs &a, hipStream_t s, int variant = 0);   // 0: eight waves of 176 x 32 (the shipped shape), 1: four waves of 176 x 64      // stamped free-running 176x256 build (tools only)
bool conv_cfg_tail_ok(int cfg, int cout, bool fp8, bool head = false);      // can tile configuration `cfg` run the fused 1x1 tail for a conv with `cout` channels
// fp8 (e4m3 x e4m3 -> fp32, v_mfma_f32_16x16x128_f8f6f4) variant of the same kernel; only the 128-B-row tile configs
bool conv_cfg_fp8_ok(int cfg);
bool conv_cfg_split_ok(int cfg);      // tile configurations instantiated for split fp16 storage (YOLO_FP16X2)
bool conv_cfg_pairk_ok(int cfg, bool split_out);      // ... and for a conv that READS interleaved pairs (pair K loop), writing pairs / plain fp16 or an fp32 head
hipError_t launch_conv_pair(const ConvArgs &a, int cfg, hipStream_t s);
// first layer of a split-fp16 network (image in three blocks hi | lo | hi -> interleaved pairs): the direct kernel, no LDS (conv_pair.hip)
// fused conv0 + conv1 of a split-fp16 network (conv_stem_pair.hip): image in three blocks -> conv1's interleaved pairs, conv0 never materialised
struct StemPairArgs {
    const void *in;                 // [N, H, W, 24] f16: hi | lo | hi blocks of the 8 padded channels
    const uint8_t *in_u8; float in_scale, in_mul, in_add;      // or (in_u8 != nullptr) the uint8 [N,H,W,3] image itself, converted in the kernel as k_preprocess + k_split_from_f32 would: x * in_scale [* in_mul + in_add], split
    const void *w0; const float *b0; int Kpad0, C0, act0;      // conv0: rows [tap][hi 8 | hi 8 | lo 8] (k = tap * 24 + ...), C0 = 16 or 32 filters
    const void *w1; const float *b1; int Kpad1, act1;          // conv1: rows [tap][W_hi 32 | W_lo 32] (k = tap * 64 + ...), 64 filters
    void *out; int out_stride;      // [N, Ho, Wo, >= 128] f16 interleaved pairs (two 32-channel groups)
    int N, H, W, Ho, Wo;
};
bool conv_stem_pair_ok(const StemPairArgs &a);
hipError_t launch_conv_stem_pair(const StemPairArgs &a, hipStream_t s);
bool conv_c8_direct_pair_ok(const ConvArgs &a);
hipError_t launch_conv_c8_direct_pair(const ConvArgs &a, hipStream_t s);      // the tiled pair-K-loop instantiations (conv_pair.hip); the halo ones: launch_conv_halo13
hipError_t launch_conv_fp8(const ConvArgs &a, int cfg, hipStream_t s);
hipError_t launch_conv_diag(const ConvArgs &a, hipStream_t s);   // stamped diagnostic build of p176c128_s2 (tools only)
// fused stem: conv 3x3/s1 (3 -> 32) + conv 3x3/s2 (32 -> 64), bf16 (conv_stem.hip)
struct StemArgs {
    const void *in; int in_stride;            // [N,H,W,8] bf16 image (3 real channels)
    const uint8_t *in_u8; float in_scale, in_mul, in_add;      // or (in_u8 != nullptr) the uint8 [N,H,W,3] image itself, converted in the kernel as k_preprocess would: x * in_scale [* in_mul + in_add]
    const void *w0; const float *b0; int Kpad0, C0, act0;    // layer 0: [C0 pad][Kpad0], k = tap*8 + ci
    const void *w1; const float *b1; int Kpad1, C1, act1;    // layer 1: [C1 pad][Kpad1], k = tap*C0 + ci
    void *out; int out_stride;                // [N,Ho,Wo,C1] bf16
    int dt;                                   // DT_BF16 or DT_F16: the 16-bit storage type of every tensor and filter of the launch
    // optional tail: a 1x1/s1 conv C1 -> C2 = 32 on the freshly produced layer-1 tile (darknet-53 layer 2); w2 == nullptr: none
    const void *w2; const float *b2; int Kpad2, C2, act2;    // [C2 pad][Kpad2], k = ci
    void *out2; int out2_stride;              // [N,Ho,Wo,C2] bf16
    int N, H, W, Ho, Wo;
    const void *zeros;
};
bool conv_stem_ok(const StemArgs &a);
hipError_t launch_conv_stem(const StemArgs &a, hipStream_t s);
// halo-staged 3x3/s1 conv, Cin = 32 -> Cout = 64, bf16, optional shortcut (conv_stem.hip)
struct HaloArgs {
    const void *in; int in_stride;            // [N,H,W,>=32] bf16
    const void *w; const float *b; int Kpad, Cin, Cout, act;   // [Cout pad][Kpad], k = tap*32 + ci
    const void *res; int res_stride;          // shortcut source [N,H,W,>=64] bf16 or nullptr
    void *out; int out_stride;
    int N, H, W;
    int dt;                                   // DT_BF16 or DT_F16
};
bool conv_halo_ok(const HaloArgs &a);
hipError_t launch_conv_halo(const HaloArgs &a, hipStream_t s);
// 3x3 / stride 2 / pad 1, 64 -> 128 channels, 16-bit storage, no shortcut (conv_s2.hip): same arguments (H, W: the INPUT size; k = tap*64 + ci)
bool conv_s2_ok(const HaloArgs &a);
hipError_t launch_conv_s2(const HaloArgs &a, hipStream_t s);
// fused residual block x + act2(conv3x3(act1(conv1x1(x)))), 128 -> 64 -> 128 channels, 16-bit storage (conv_block.hip)
struct BlockArgs {
    const void *x; int x_stride;              // [N,H,W,>=128]: input of the 1x1 and source of the shortcut
    const void *w1; const float *b1; int Kpad1, act1;         // 1x1: [64 pad][Kpad1], k = ci
    const void *w2; const float *b2; int Kpad2, act2;         // 3x3: [128 pad][Kpad2], k = tap*64 + ci
    void *out; int out_stride;                // [N,H,W,>=128]
    int N, H, W, C, Cmid;
    int dt;                                   // DT_BF16 or DT_F16
};
bool conv_resblock_ok(const BlockArgs &a);
hipError_t launch_conv_resblock(const BlockArgs &a, hipStream_t s);
// the same block at C = 64, Cmid = 32 (darknet-53's first residual block, 208 x 208 at 416 x 416): conv_block64.hip, two workgroups per CU,
// shortcut from the x tile in LDS
bool conv_resblock64_ok(const BlockArgs &a);
hipError_t launch_conv_resblock64(const BlockArgs &a, hipStream_t s);
// exact-fp32 MFMA conv (config 2); same argument meaning, in/wt/res are float
hipError_t launch_conv_f32(const ConvArgs &a, hipStream_t s);

// ---- memory-bound operators (ew_ops.hip) ------------------------------------------------------
hipError_t launch_preprocess(const void *img, int fmt /*0 u8, 1 f32*/, int n, int hw, float scale,
                             void *out, int out_dt, int out_stride, hipStream_t s, float post_mul = 1.0f, float post_add = 0.0f);
hipError_t launch_resize_u8(const uint8_t *img, int h, int w, int s_out, void *out, int out_dt,
                            int out_stride, int out_c, hipStream_t s, float post_scale = 1.0f, float post_add = 0.0f);
// cv2.resize (INTER_LINEAR, float32) of a uint8 [h,w,3] image to fp32 [oh,ow,3], optional BGR -> RGB, then / divisor (V2/utils.py:13-27)
hipError_t launch_resize_cv2_u8(const uint8_t *img, int h, int w, int oh, int ow, int swap_rb, float divisor, float *out, hipStream_t s);
hipError_t launch_upsample2x(const TView &in, const TView &out, int bilinear, hipStream_t s);
hipError_t launch_maxpool(const TView &in, const TView &out, int size, int stride, int pad, hipStream_t s);
hipError_t launch_reorg(const TView &in, const TView &out, int stride, int darknet, hipStream_t s);
// out = (a * sa + b * sb) * so   (the scales are the fp8 tensor scales; 1 for bf16 / fp32)
hipError_t launch_add(const TView &a, const TView &b, const TView &out, hipStream_t s, float sa = 1.f, float sb = 1.f, float so = 1.f);
hipError_t launch_copy(const TView &in, const TView &out, hipStream_t s);
// [local] (locally connected, DN/local_layer.c): w [locations][filters][k][k][C] in the tensors' type, bias [locations][filters] fp32
hipError_t launch_local(const TView &in, const TView &out, const void *w, const float *bias, int k, int stride, int pad, int act, hipStream_t s);
hipError_t launch_to_f32(const TView &in, float *out, hipStream_t s, float scale = 1.f);   // dense NHWC fp32 copy (* scale)
hipError_t launch_from_f32(const float *in, const TView &out, hipStream_t s, float scale = 1.f);   // (in * scale) -> view
// split fp16 storage (YOLO_FP16X2; ew_ops.hip).  Two layouts of a pair tensor of Cp padded channels:
//   PAIR_ILV  [pixel][stride >= 2 * Cp], Cp a multiple of 32: per 32-channel group 32 hi then 32 lo (every layer output; round 6)
//   PAIR_B3   [pixel][3 * Cp], Cp a multiple of 8: blocks hi | lo | hi (the network INPUT only: 8 padded channels, rounds 4-5's form)
enum { PAIR_ILV = 0, PAIR_B3 = 1 };
hipError_t launch_split_from_f32(const float *in, int in_stride, void *out, int out_stride, int Cp, size_t npix, hipStream_t s, int layout = PAIR_ILV);   // fp32 [pixel][in_stride >= Cp] -> pairs
hipError_t launch_split_to_f32(const void *in, int in_stride, int Cp, float *out, int out_stride, size_t npix, hipStream_t s, int layout = PAIR_ILV);      // pairs -> fp32 (hi + lo)
hipError_t launch_upsample2x_pair(const TView &in, const TView &out, int bilinear, hipStream_t s);      // 2x upsample of an interleaved pair tensor (join, fp32 lerp, split) in one launch
hipError_t launch_add_split(const void *a, int a_stride, const void *b, int b_stride, void *out, int out_stride, int Cp, size_t npix, hipStream_t s);      // shortcut on interleaved pair tensors

// ---- head decode + postprocess (post_ops.hip) ---------------------------------------------------
struct DecodeArgs {
    const float *raw; int raw_stride;   // [n, g*g, raw_stride] fp32 head conv output
    int n, g, na, classes;
    float anchors[2 * 16];              // pixels (yolo) or grid units (region), masked order
    int img_size;
    int mode;                           // yolo_decode
    int region;                         // 1: softmax/region head
    float *det; int rows_total; int row_off;   // det [n, rows_total, 5+classes]; nullptr: the decoded tensor is not materialised ...
    float *box4;                        // ... only (cx, cy, w, h) of every row, [n, rows_total, 4] (yolo heads, with scores/labels)
    float reject_below;                 // lean form: a box whose objectness is below this cannot reach the caller's score threshold
                                        // (score = objectness * class probability <= objectness): its class work is skipped and its
                                        // score is reported as the objectness itself.  -inf: every box is scored
};
// Lean decode of up to four [yolo] heads in ONE launch (yolo_detect*: the decodes of a three-scale network are three short,
// latency-bound launches otherwise; the head tensors keep their own buffers, so the early heads can wait for the last one)
struct LeanHead { const float *raw; const float *obj; int raw_stride, g, na, row_off; long box_begin; float anchors[2 * 16]; };      // obj: compact objectness-logit plane [n * g * g][na] written by the head conv, or nullptr (read from raw)
struct LeanArgs {
    int nheads; LeanHead h[4];
    long total;                          // boxes of all heads: n * sum(g * g * na)
    int n, classes, img_size, mode, rows_total;
    float *box4; float reject_below;
    uint4 *list; unsigned *list_count; unsigned list_cap;      // boxes that pass the objectness pre-filter (descriptor each); list_count[0] = entries, [1] = phase-2 workgroups done (both zero between launches)
};
hipError_t launch_decode_lean(const LeanArgs &a, float *scores, int *labels, hipStream_t s);
// scores/labels (nullable): per-row max_k(obj*cls_k) and its first argmax, written alongside the decode
hipError_t launch_decode(const DecodeArgs &a, float *scores, int *labels, hipStream_t s);
// YOLOv1 [detection] head (row D1): raw [n][raw_stride] fp32 = [cls S*S*C | conf S*S*B | box S*S*B*4] -> det rows (cx, cy, w, h, conf, cls...)
hipError_t launch_decode_v1(const float *raw, int raw_stride, int n, int side, int num, int classes, int sqr, float *det, int rows_total,
                            int row_off, float *scores, int *labels, hipStream_t s);

struct PostArgs {
    const float *det; int n, rows, attrs;
    float score_thr, iou_thr; int max_out, nms_mode, select_mode;
    const float *box4;                  // non-null: [n, rows, 4] (cx,cy,w,h) rows written by the lean decode; `det` is not read
    int scores_ready;                   // 1: scores/labels were produced by the decode kernel already
    int corners_in;                     // 1: det rows already hold (x0,y0,x1,y1) instead of (cx,cy,w,h)
    int img_h, img_w;                   // V2 numpy flavour only: pixel box scaling (V2/utils.py:32-43); 0 = off
    // workspace (device), sized for n images: scores/labels/cand/slabel/sscore [n*rows], sbox float4 [n*rows],
    // keys u64 [n*rows_pow2]
    float *scores; int *labels; int *cand; unsigned long long *keys; int rows_pow2;
    float4 *sbox; int *slabel; float *sscore;
    void *boxes_out; int *counts_out;   // yolo_box [n*max_out], int [n] (device)
    // optional: the decoded-tensor row (0 .. rows-1, within its image) every kept record came from, [n*max_out], -1 in unused slots;
    // needs the workspace srow [n*rows].  nullptr: not reported
    int *rows_out; int *srow;
    unsigned *zero_word;                // optional: a device word this launch resets to 0 (the lean decode's list counter)
};
hipError_t launch_postprocess(const PostArgs &a, hipStream_t s);
hipError_t launch_letterbox_chw(const float *img, int iw, int ih, int S, void *out, int out_dt, int out_stride, hipStream_t s);
hipError_t launch_nms_dets(const float4 *boxes, float *prob, float *objectness, int n, int classes, float thresh, int by_obj, hipStream_t s);
// darknet get_network_boxes on the device (post_ops.hip): ordered compaction + letterbox correction of one image's decoded rows
struct DnBoxesArgs {
    const float *det; int attrs;          // decoded rows of ONE image [rows][attrs]
    int nheads, kind[8], grid[8], na[8], off[8];   // per head: 0 yolo / 1 region, grid size, anchors, first row
    float thresh; int w, h, netw, neth, relative;
    // kind 2, a [detection] head (get_detection_detections, DN/detection_layer.c:225-254): every one of the side * side * num boxes,
    // straight from the layer's input vector (= its output in inference): raw [classes | confidences | boxes]
    const float *raw; int side, classes, sqr;
    float *rec;                           // [cap][attrs]: x, y, w, h, objectness, prob[classes]; nullptr = count only
    int *src;                             // workspace [>= cap] (row index of every kept box)
    int *count; int cap;
};
hipError_t launch_darknet_boxes(const DnBoxesArgs &a, hipStream_t s);
// last head's raw output -> darknet's layer-output layout (planar, activations applied); image 0
hipError_t launch_head_darknet_layout(const float *raw, int raw_stride, int cells, int na, int classes, int region, float *out, hipStream_t s);
// darknet letterbox_image / resize_image on a planar float image -> planar float canvas w x h (DN/image.c:960-981)
hipError_t launch_letterbox_planar(const float *img, int iw, int ih, int w, int h, int embed, float *out, hipStream_t s);
hipError_t launch_boxes_to_corners(const float *in, float *out, size_t nrows, int attrs, hipStream_t s);
