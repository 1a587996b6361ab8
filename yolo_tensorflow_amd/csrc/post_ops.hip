// Detection-head post-processing on the device (the reference has NO GPU kernel for this: darknet pulls
// the head tensor to the host, DN/yolo_layer.c:347-362, and runs DN/box.c on the CPU; the TF scripts run
// tf.boolean_mask / tf.image.non_max_suppression or pure-numpy loops, V3/yolo_v3.py:376-420).
//
//   k_decode_yolo / k_decode_region : rows D3 / D2 -- raw head conv output -> [n, rows, 5+C] fp32
//   k_score_rows                    : row S        -- score = max_k(obj*cls_k), label = first argmax
//   k_nms_image (1 workgroup/image) : rows S+N     -- order-preserving threshold compaction, sort by
//                                     score (ties: lower row first), greedy suppression, top max_out
//
// Built with -ffp-contract=off: every IoU is evaluated with the reference's operation order in fp32
// (or fp64 for the V2 numpy flavour) so the kept set is bit-identical to the oracle on equal inputs.
#include "kernels.h"
#include <math.h>

struct BoxOut { float x0, y0, x1, y1, score; int cls; };

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
// hardware-rate sigmoid for the 83 of 85 attributes that are probabilities: v_exp_f32 (2^x, 1 ulp) on x*log2(e) and
// v_rcp_f32 (1 ulp).  |rel err| <= ~1.2e-6 for |x| <= 20 (the rounding of x*log2e), inside the 3e-6 the decode is
// tested to; box sizes keep the full-precision expf.
__device__ __forceinline__ float sigmoid_fast(float x)
{
    const float e = __builtin_amdgcn_exp2f(-x * 1.44269504088896340736f);
    return __builtin_amdgcn_rcpf(1.0f + e);
}
void launch_score_rows(const float *det, size_t nrows, int attrs, float *scores, int *labels, hipStream_t s, int objectness_mode = 0);

// wave-wide (value, first index) arg-max over lanes; every lane returns the result
__device__ __forceinline__ void wave_argmax(float &v, int &idx)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        float ov = __shfl_xor(v, off);
        int oi = __shfl_xor(idx, off);
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
}

// ---- D3 (+S): `_detection_layer` (V3/yolo_v3.py:111-159) / `_ratio_detection_layer` (V3/YOLOV3.py:168-238),
//      fused with the row score of V3/YOLOV3.py:353-357.  One wave per candidate box: lanes run along the
//      5+C attributes (contiguous in the head tensor and in the decoded tensor -> fully coalesced), the
//      max/argmax over classes is a wave shuffle reduction. ----
__global__ __launch_bounds__(256) void k_decode_yolo(const DecodeArgs a, float *scores, int *labels)
{
    const int attrs = 5 + a.classes;
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    const long total = (long)a.n * a.g * a.g * a.na;
    const int stride = a.img_size / a.g;
    const float G = (float)a.g, S = (float)stride;
    const int gg = a.g * a.g;
    constexpr int U = 4;                       // boxes in flight per wave (memory-level parallelism)
    const bool two = attrs > 64;               // second 64-attribute pass needed (attrs <= 128 is enforced by the launcher)
    for (long b0 = wave * U; b0 < total; b0 += nwaves * U) {
        float v0[U], v1[U];
        const float *src[U]; float *dst[U]; size_t rowi[U]; int an_[U], cell_[U];
        // (image, cell, anchor) of the first box by 32-bit division, the next three by increment: the index math is
        // wave-uniform and must stay far cheaper than the 85 sigmoids it serves
        const unsigned ub = (unsigned)__builtin_amdgcn_readfirstlane((int)b0);
        unsigned t0 = ub / (unsigned)a.na; int an = (int)(ub - t0 * a.na);
        int b = (int)(t0 / (unsigned)gg); int cell = (int)(t0 - (unsigned)b * gg);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (u > 0 && b0 + u < total) { if (++an == a.na) { an = 0; if (++cell == gg) { cell = 0; ++b; } } }
            an_[u] = an; cell_[u] = cell;
            src[u] = a.raw + ((size_t)b * gg + cell) * a.raw_stride + an * attrs;
            rowi[u] = (size_t)b * a.rows_total + a.row_off + (size_t)cell * a.na + an;
            dst[u] = a.det + rowi[u] * attrs;
            v0[u] = lane < attrs ? src[u][lane] : 0.f;
            v1[u] = (two && lane + 64 < attrs) ? src[u][lane + 64] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (b0 + u >= total) break;
            const int an = an_[u], cell = cell_[u];
            float r0;
            if (lane < 2) {
                const float off = (float)(lane == 0 ? cell % a.g : cell / a.g);
                const float sg = sigmoidf_(v0[u]) + off;
                r0 = a.mode == 0 ? sg / G : sg * S;
            } else if (lane < 4) {
                const float e = expf(v0[u]) * a.anchors[2 * an + (lane - 2)];   // anchors pre-divided by stride on the host
                r0 = a.mode == 0 ? e / G : e * S;
            } else {
                r0 = sigmoid_fast(v0[u]);
            }
            const float r1 = sigmoid_fast(v1[u]);
            if (lane < attrs) dst[u][lane] = r0;
            if (two && lane + 64 < attrs) dst[u][lane + 64] = r1;
            const float obj = __shfl(r0, 4);
            float best = -INFINITY; int bi = 0x7fffffff;
            if (lane >= 5 && lane < attrs) { best = obj * r0; bi = lane - 5; }
            if (two && lane + 64 < attrs) { const float sc = obj * r1; if (sc > best) { best = sc; bi = lane + 59; } }
            wave_argmax(best, bi);
            if (lane == 0 && scores) { scores[rowi[u]] = best; labels[rowi[u]] = bi; }
        }
    }
}

// wave-wide maximum on the vector ALU only (DPP row operations, no LDS traffic): lane 63 ends up with the maximum of all
// 64 lanes, which is broadcast with a readlane
__device__ __forceinline__ float wave_max_dpp(float v)
{
#define DPP_MAX(ctrl, rmask)                                                                                      \
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), ctrl, rmask, 0xF, false)))
    DPP_MAX(0xB1, 0xF);      // quad_perm [1,0,3,2]
    DPP_MAX(0x4E, 0xF);      // quad_perm [2,3,0,1]
    DPP_MAX(0x124, 0xF);     // row_ror:4
    DPP_MAX(0x128, 0xF);     // row_ror:8   -> every lane holds its row's (16 lanes) maximum
    DPP_MAX(0x142, 0xA);     // row_bcast:15 into rows 1 and 3
    DPP_MAX(0x143, 0xC);     // row_bcast:31 into rows 2 and 3
#undef DPP_MAX
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// Same decode, one wave per grid CELL (all `na` boxes of a pixel = na*(5+C) <= 256 contiguous floats on both sides):
// lane l owns channels l, l+64, l+128, l+192, so every load and store instruction is one fully coalesced 256-B run, the
// index arithmetic is paid once per cell instead of once per box, and the per-box arg-max is a DPP max + one ballot.
// The wave-per-box form above spent ~150 instructions per box (index math, two half-empty passes, 12 bpermutes) and
// ran the 52x52 head at 2 TB/s.
template <int R>
__global__ __launch_bounds__(256) void k_decode_yolo_cell(const DecodeArgs a, float *scores, int *labels)
{
    const int attrs = 5 + a.classes, nch = a.na * attrs;
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    const int gg = a.g * a.g;
    const long total = (long)a.n * gg;
    const int stride = a.img_size / a.g;
    const float G = (float)a.g, S = (float)stride;
    // per-lane channel decomposition, constant over the cells
    int an_[R], k_[R]; bool ok_[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int ch = lane + 64 * r;
        ok_[r] = ch < nch;
        an_[r] = ch / attrs; k_[r] = ch - an_[r] * attrs;
    }
    bool spec_[R]; unsigned long long seen = 0; bool merged = true;      // geometry lanes; mergeable when no lane repeats
#pragma unroll
    for (int r = 0; r < R; ++r) {
        spec_[r] = ok_[r] && k_[r] < 4;
        const unsigned long long m = __ballot(spec_[r]);
        merged = merged && !(seen & m); seen |= m;
    }
    // the next cell's loads are issued before this cell is decoded (one cell of prefetch per wave: the head tensor was
    // written by the previous kernel and comes from the Infinity Cache / HBM, ~1-2 us away)
    float xn[R];
    if (wave < total) {
        const float *src = a.raw + (size_t)wave * a.raw_stride;
#pragma unroll
        for (int r = 0; r < R; ++r) xn[r] = ok_[r] ? src[lane + 64 * r] : 0.f;
    }
    for (long p = wave; p < total; p += nwaves) {
        const unsigned up = (unsigned)__builtin_amdgcn_readfirstlane((int)p);
        const int b = (int)(up / (unsigned)gg), cell = (int)(up - (unsigned)b * gg);
        const int cy = cell / a.g, cx = cell - cy * a.g;
        const size_t row0 = (size_t)b * a.rows_total + a.row_off + (size_t)cell * a.na;
        float *dst = a.det ? a.det + row0 * attrs : nullptr;
        float *dst4 = a.box4 ? a.box4 + row0 * 4 : nullptr;            // lean form: geometry only (the caller wants boxes, not the tensor)
        float x[R], v[R];
#pragma unroll
        for (int r = 0; r < R; ++r) x[r] = xn[r];
        if (p + nwaves < total) {
            const float *src = a.raw + (size_t)(p + nwaves) * a.raw_stride;
#pragma unroll
            for (int r = 0; r < R; ++r) xn[r] = ok_[r] ? src[lane + 64 * r] : 0.f;
        }
        // 4 of every (5+C) channels are box geometry and need the full-precision expf; the rest take the hardware-rate
        // sigmoid.  The geometry lanes of the R registers are (normally) disjoint lane sets, so they are merged into ONE
        // register and the expensive divergent branch runs once per cell instead of once per register.
        float xs = 0.f; int ks = 4, ans = 0;
        if (merged) {
#pragma unroll
            for (int r = 0; r < R; ++r) if (spec_[r]) { xs = x[r]; ks = k_[r]; ans = an_[r]; }
            float vs = 0.f;
            if (ks < 2) {
                const float sg = sigmoidf_(xs) + (float)(ks == 0 ? cx : cy);
                vs = a.mode == 0 ? sg / G : sg * S;
            } else if (ks < 4) {
                const float e = expf(xs) * a.anchors[2 * ans + (ks - 2)];       // anchors pre-divided by stride on the host
                vs = a.mode == 0 ? e / G : e * S;
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                v[r] = spec_[r] ? vs : sigmoid_fast(x[r]);
                if (dst) { if (ok_[r]) dst[lane + 64 * r] = v[r]; }
                else if (spec_[r]) dst4[an_[r] * 4 + k_[r]] = v[r];
            }
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int k = k_[r];
                if (k < 2) {
                    const float sg = sigmoidf_(x[r]) + (float)(k == 0 ? cx : cy);
                    v[r] = a.mode == 0 ? sg / G : sg * S;
                } else if (k < 4) {
                    const float e = expf(x[r]) * a.anchors[2 * an_[r] + (k - 2)];
                    v[r] = a.mode == 0 ? e / G : e * S;
                } else {
                    v[r] = sigmoid_fast(x[r]);
                }
                if (dst) { if (ok_[r]) dst[lane + 64 * r] = v[r]; }
                else if (spec_[r]) dst4[an_[r] * 4 + k_[r]] = v[r];
            }
        }
        if (!scores) continue;
        for (int an = 0; an < a.na; ++an) {
            // objectness of box `an` sits at channel an*attrs + 4
            const int oc = an * attrs + 4, ol = oc & 63, orr = oc >> 6;
            float ov = v[0];
#pragma unroll
            for (int r = 1; r < R; ++r) ov = orr == r ? v[r] : ov;
            const float obj = __shfl(ov, ol);
            if (obj < a.reject_below) {                  // wave-uniform: cannot pass the threshold whatever its classes say
                if (lane == 0) { scores[row0 + an] = obj; labels[row0 + an] = 0; }
                continue;
            }
            float sc[R]; float best = -INFINITY;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                sc[r] = (ok_[r] && an_[r] == an && k_[r] >= 5) ? obj * v[r] : -INFINITY;
                best = fmaxf(best, sc[r]);
            }
            best = wave_max_dpp(best);
            // first (lowest class index = lowest channel) position holding the maximum: registers in ascending order,
            // lowest lane inside a register
            int label = 0x7fffffff;
#pragma unroll
            for (int r = R - 1; r >= 0; --r) {
                const unsigned long long m = __ballot(sc[r] == best);
                if (m) label = (int)__builtin_ctzll(m) + 64 * r - an * attrs - 5;
            }
            if (lane == 0) { scores[row0 + an] = best; labels[row0 + an] = label; }
        }
    }
}

// Lean decode, objectness first.  When the caller only wants boxes above a score threshold (yolo_detect*: no decoded tensor), a box
// whose objectness is below the threshold cannot pass whatever its classes say (score = objectness x class probability), and with
// trained -- or the synthetic -- weights that is 95+ % of the 10647 boxes of an image.  Phase 1: one LANE per box reads just the
// objectness logit (one 128-byte line of the cell's 1020 bytes per box) and settles every box below the threshold with its score
// reported as the objectness itself, exactly as the cell-per-wave kernel does.  Phase 2: the lanes whose boxes remain decode them
// themselves -- the same arithmetic, the same first-maximum label.  The cell-per-wave kernel spent
// ~400 instructions per cell on boxes that were then thrown away (the 52x52 head: 45 us; this form: the objectness lines + a few
// per cent of the boxes).
__global__ __launch_bounds__(256) void k_decode_yolo_lean(const DecodeArgs a, float *scores, int *labels)
{
    const int attrs = 5 + a.classes;
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    const int gg = a.g * a.g;
    const long total = (long)a.n * gg * a.na;
    const int stride = a.img_size / a.g;
    const float G = (float)a.g, S = (float)stride;
    for (long base = wave * 64; base < total; base += nwaves * 64) {
        const long box = base + lane;
        const bool valid = box < total;
        const unsigned ubox = valid ? (unsigned)box : 0u;
        const unsigned cidx = ubox / (unsigned)a.na; const int an = (int)(ubox - cidx * (unsigned)a.na);      // cell index over n * g * g
        const unsigned b = cidx / (unsigned)gg; const int cell = (int)(cidx - b * (unsigned)gg);
        const size_t row = (size_t)b * a.rows_total + a.row_off + (size_t)cell * a.na + an;
        const float obj = sigmoid_fast(a.raw[(size_t)cidx * a.raw_stride + an * attrs + 4]);
        const bool pass = valid && !(obj < a.reject_below);
        if (valid && !pass) { scores[row] = obj; labels[row] = 0; }
        if (pass) {
            // this lane decodes its own box: the attribute loads of the passing lanes are independent of each other (memory-level
            // parallelism whatever the pass rate -- the synthetic weights let a quarter of the boxes through, trained ones far fewer);
            // a cooperative wave-per-box loop here was latency-bound at 1-2 us per passing box
            const float *src = a.raw + (size_t)cidx * a.raw_stride + an * attrs;
            const float g0 = src[0], g1 = src[1], g2 = src[2], g3 = src[3];
            float best = -INFINITY; int label = 0;
            for (int k = 0; k < a.classes; ++k) {                   // ascending: the first maximum wins, like the wave kernels' ballot
                const float sc = obj * sigmoid_fast(src[5 + k]);
                if (sc > best) { best = sc; label = k; }
            }
            const float sx = sigmoidf_(g0) + (float)(cell % a.g), sy = sigmoidf_(g1) + (float)(cell / a.g);
            const float ew = expf(g2) * a.anchors[2 * an], eh = expf(g3) * a.anchors[2 * an + 1];      // anchors pre-divided by stride on the host
            float4 o;
            o.x = a.mode == 0 ? sx / G : sx * S; o.y = a.mode == 0 ? sy / G : sy * S;
            o.z = a.mode == 0 ? ew / G : ew * S; o.w = a.mode == 0 ? eh / G : eh * S;
            *(float4 *)(a.box4 + row * 4) = o;
            scores[row] = best; labels[row] = label;
        }
    }
}

// The same for every head of the network in one launch, with a cooperative second phase.  Phase 1 as above (one lane per box reads
// the objectness logit).  Phase 2: the boxes that remain are taken four at a time, one per 16-lane row of the wave: the row reads the
// box's 5 + C attributes as coalesced 64-byte runs (the lane-per-box loop walked 340 bytes per lane, 85 dependent-address loads each,
// with the other lanes of the wave idle), scores them, and reduces maximum and first arg-max inside the row with DPP row operations.
// Same arithmetic per element as k_decode_yolo_lean, same first-maximum label: bit-identical scores, labels and boxes.
__device__ __forceinline__ float row_max_dpp(float v)         // maximum over the 16 lanes of a DPP row, in every lane of the row
{
#define DPP_MAXR(ctrl) v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, false)))
    DPP_MAXR(0xB1); DPP_MAXR(0x4E); DPP_MAXR(0x124); DPP_MAXR(0x128);
#undef DPP_MAXR
    return v;
}
__device__ __forceinline__ int row_min_dpp(int v)
{
#define DPP_MINR(ctrl) v = min(v, __builtin_amdgcn_update_dpp(v, v, ctrl, 0xF, 0xF, false))
    DPP_MINR(0xB1); DPP_MINR(0x4E); DPP_MINR(0x124); DPP_MINR(0x128);
#undef DPP_MINR
    return v;
}
// Phase 1: one lane per box.  Boxes below the threshold are settled here; the others are appended to a list (one atomic per wave) that
// phase 2 spreads over the whole chip -- the passing boxes cluster (an anchor, an image), and a wave that had to finish its own 64
// boxes four at a time was the kernel's tail: 49 us for 14 000 boxes, 5 us without them.
__global__ __launch_bounds__(1024) void k_decode_yolo_lean_p1(const LeanArgs a, float *scores, int *labels)
{
    __shared__ unsigned wcnt[16], s_base;
    const int attrs = 5 + a.classes;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (long base = (long)blockIdx.x * 1024; base < a.total; base += (long)gridDim.x * 1024) {      // (block-uniform trip count: barriers inside)
        const long box = base + threadIdx.x;
        const bool valid = box < a.total;
        int hd = 0;
#pragma unroll
        for (int k = 1; k < 4; ++k) if (k < a.nheads && box >= a.h[k].box_begin) hd = k;
        // per-lane head parameters (the four heads' descriptors sit in the kernel arguments: scalar selects)
        const float *raw = a.h[0].raw, *objp = a.h[0].obj; int raw_stride = a.h[0].raw_stride, g = a.h[0].g, na = a.h[0].na, row_off = a.h[0].row_off; long bb = a.h[0].box_begin;
#pragma unroll
        for (int k = 1; k < 4; ++k) if (hd == k) { raw = a.h[k].raw; objp = a.h[k].obj; raw_stride = a.h[k].raw_stride; g = a.h[k].g; na = a.h[k].na; row_off = a.h[k].row_off; bb = a.h[k].box_begin; }
        const int gg = g * g;
        const unsigned ubox = valid ? (unsigned)(box - bb) : 0u;
        const unsigned cidx = ubox / (unsigned)na; const int an = (int)(ubox - cidx * (unsigned)na);      // cell index over n * g * g
        const unsigned b = cidx / (unsigned)gg; const int cell = (int)(cidx - b * (unsigned)gg);
        const unsigned row = b * (unsigned)a.rows_total + (unsigned)row_off + (unsigned)cell * (unsigned)na + (unsigned)an;
        const unsigned eoff = cidx * (unsigned)raw_stride + (unsigned)(an * attrs);              // element offset of the box in its head tensor
        // (the head conv leaves the objectness logits in a compact plane as well: consecutive lanes read consecutive floats)
        const float obj = sigmoid_fast(objp ? objp[ubox] : raw[eoff + 4]);
        const bool pass = valid && !(obj < a.reject_below);
        if (valid && !pass) { scores[row] = obj; labels[row] = 0; }
        // one atomic per WORKGROUP and step (all of them hit one word: ~11 ns each, whoever issues them): wave counts through LDS
        const unsigned long long m = __ballot(pass);
        if (lane == 0) wcnt[wv] = (unsigned)__popcll(m);
        __syncthreads();
        unsigned pre = 0, tot = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) { const unsigned cnt = wcnt[k]; if (k < wv) pre += cnt; tot += cnt; }
        if (threadIdx.x == 0 && tot) s_base = atomicAdd(a.list_count, tot);
        __syncthreads();
        const unsigned slot = s_base + pre + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
        // cell < 2^24 (grid <= 4096), anchor < 16, head < 4
        if (pass && slot < a.list_cap) a.list[slot] = uint4{eoff, row, (unsigned)cell | ((unsigned)an << 24) | ((unsigned)hd << 28), __float_as_uint(obj)};
    }
}
// Phase 2: one 16-lane row per listed box, grid-stride.  The row reads the box's 5 + C attributes as coalesced 64-byte runs, all loads
// issued before the first use, scores them and reduces maximum and first arg-max inside the row with DPP row operations -- the same
// arithmetic per element as k_decode_yolo_lean, the same first-maximum label: bit-identical scores, labels and boxes.  
__global__ __launch_bounds__(256) void k_decode_yolo_lean_p2(const LeanArgs a, float *scores, int *labels)
{
    const int attrs = 5 + a.classes;
    const int l15 = threadIdx.x & 15;
    const unsigned rowid = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, nrows = (gridDim.x * blockDim.x) >> 4;
    unsigned count = __hip_atomic_load(a.list_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (count > a.list_cap) count = a.list_cap;
    for (unsigned base = 0; base < count; base += nrows) {               // (uniform trip count: the DPP reductions want whole rows)
        const unsigned idx = base + rowid;
        const bool act = idx < count;
        const uint4 d = act ? a.list[idx] : uint4{0, 0, 0, 0};
        const unsigned r_eoff = d.x, r_row = d.y; const int r_cell = (int)(d.z & 0xffffffu), r_an = (int)((d.z >> 24) & 15u), r_hd = (int)(d.z >> 28);
        const float r_obj = __uint_as_float(d.w);
        const float *r_raw = a.h[0].raw; int r_g = a.h[0].g;
#pragma unroll
        for (int k = 1; k < 4; ++k) if (r_hd == k) { r_raw = a.h[k].raw; r_g = a.h[k].g; }
        const float *src = r_raw + r_eoff;
        float vv[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) { const int k = l15 + 16 * t; vv[t] = (act && k < attrs) ? src[k] : 0.f; }
        const float g_raw = vv[0];
        float best = -INFINITY; int label = 0x7fffffff;
#pragma unroll
        for (int t = 0; t < 8; ++t) {                                    // ascending k inside a lane: a strict > keeps the first maximum
            const int k = l15 + 16 * t;
            if (act && k >= 5 && k < attrs) { const float sc = r_obj * sigmoid_fast(vv[t]); if (sc > best) { best = sc; label = k - 5; } }
        }
        const float rbest = row_max_dpp(best);
        const int rlabel = row_min_dpp(best == rbest ? label : 0x7fffffff);      // lowest class index holding the maximum
        if (act && l15 < 4) {
            const int stride = a.img_size / r_g;
            const float G = (float)r_g, S = (float)stride;
            float o;
            if (l15 < 2) { const float sg = sigmoidf_(g_raw) + (float)(l15 == 0 ? r_cell % r_g : r_cell / r_g); o = a.mode == 0 ? sg / G : sg * S; }
            else {
                float anc = a.h[0].anchors[0];
#pragma unroll
                for (int k = 0; k < 4; ++k) if (r_hd == k) anc = a.h[k].anchors[2 * r_an + (l15 - 2)];
                const float e = expf(g_raw) * anc;                   // anchors pre-divided by stride on the host
                o = a.mode == 0 ? e / G : e * S;
            }
            a.box4[(size_t)r_row * 4 + l15] = o;
            if (l15 == 0) { scores[r_row] = rbest; labels[r_row] = rlabel; }
        }
    }
    // (the list counter is zeroed by the NMS kernel that follows -- PostArgs::zero_word -- or by a memset in front of phase 1 when
    //  no NMS ran since: a last-workgroup-done reset here cost a thousand more atomics on one word)
}
hipError_t launch_decode_lean(const LeanArgs &a, float *scores, int *labels, hipStream_t s)
{
    if (a.nheads < 1 || a.nheads > 4 || !a.box4 || !scores || !labels || 5 + a.classes > 128 || !a.list || !a.list_count) return hipErrorInvalidValue;
    size_t blocks = ((size_t)a.total + 1023) / 1024; if (blocks > 256 * 2) blocks = 256 * 2;
    hipLaunchKernelGGL(k_decode_yolo_lean_p1, dim3((unsigned)blocks), dim3(1024), 0, s, a, scores, labels);
    hipLaunchKernelGGL(k_decode_yolo_lean_p2, dim3(1024), dim3(256), 0, s, a, scores, labels);
    return hipGetLastError();
}

// ---- D2: V2 `decode` (V2/decode.py:13-47): sigmoid xy/obj, exp wh, softmax classes; stored as
//      (bx, by, bw, bh, obj, cls...) normalised; corners are formed at selection time ----
__global__ void k_decode_region(const DecodeArgs a)
{
    const int attrs = 5 + a.classes;
    size_t box = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)a.n * a.g * a.g * a.na;
    if (box >= total) return;
    int an = (int)(box % a.na); size_t t = box / a.na;
    int cell = (int)(t % (a.g * a.g)); int b = (int)(t / (a.g * a.g));
    const float *p = a.raw + ((size_t)b * a.g * a.g + cell) * a.raw_stride + an * attrs;
    float *o = a.det + ((size_t)b * a.rows_total + a.row_off + (size_t)cell * a.na + an) * attrs;
    const float G = (float)a.g;
    o[0] = ((float)(cell % a.g) + sigmoidf_(p[0])) / G;
    o[1] = ((float)(cell / a.g) + sigmoidf_(p[1])) / G;
    o[2] = (a.anchors[2 * an] * expf(p[2])) / G;
    o[3] = (a.anchors[2 * an + 1] * expf(p[3])) / G;
    o[4] = sigmoidf_(p[4]);
    float mx = -INFINITY;
    for (int k = 0; k < a.classes; ++k) mx = fmaxf(mx, p[5 + k]);
    float sum = 0.f;
    for (int k = 0; k < a.classes; ++k) sum += expf(p[5 + k] - mx);
    for (int k = 0; k < a.classes; ++k) o[5 + k] = expf(p[5 + k] - mx) / sum;
}

// ---- D1: YOLOv1 `_build_detector` before the selection (V1/YOLO_V1_Inference.py:213-244; darknet twin get_detection_detections
//      DN/detection_layer.c:225-254): x = (bx + col) / S, y = (by + row) / S, w = bw^2, h = bh^2 (`tf.square`, darknet's sqrt=1),
//      class-specific score = conf * cls, label = first arg-max.  98 boxes per image: one thread per box. ----
__global__ void k_decode_v1(const float *raw, int raw_stride, int n, int S, int B, int C, int sqr, float *det, int rows_total, int row_off,
                            float *scores, int *labels)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int per = S * S * B;
    if (i >= n * per) return;
    const int img = i / per, r = i - img * per, cell = r / B, b = r - cell * B;
    const float *p = raw + (size_t)img * raw_stride;
    const float *cls = p + cell * C, *box = p + S * S * (C + B) + (cell * B + b) * 4;
    const float conf = p[S * S * C + cell * B + b];
    const int attrs = 5 + C;
    const size_t row = (size_t)img * rows_total + row_off + r;
    float *o = det + row * attrs;
    o[0] = (box[0] + (float)(cell % S)) / (float)S;
    o[1] = (box[1] + (float)(cell / S)) / (float)S;
    o[2] = sqr ? box[2] * box[2] : box[2];
    o[3] = sqr ? box[3] * box[3] : box[3];
    o[4] = conf;
    float best = -INFINITY; int bi = 0;
    for (int k = 0; k < C; ++k) { const float c = cls[k]; o[5 + k] = c; const float sc = conf * c; if (sc > best) { best = sc; bi = k; } }
    if (scores) { scores[row] = best; labels[row] = bi; }
}
hipError_t launch_decode_v1(const float *raw, int raw_stride, int n, int side, int num, int classes, int sqr, float *det, int rows_total,
                            int row_off, float *scores, int *labels, hipStream_t s)
{
    const int total = n * side * side * num;
    hipLaunchKernelGGL(k_decode_v1, dim3((total + 127) / 128), dim3(128), 0, s, raw, raw_stride, n, side, num, classes, sqr, det, rows_total, row_off, scores, labels);
    return hipGetLastError();
}

hipError_t launch_decode(const DecodeArgs &a, float *scores, int *labels, hipStream_t s)
{
    // the lean form (no decoded tensor) exists in the objectness-first and cell-per-wave kernels only
    if (!a.det && (a.region || !a.box4 || !scores || a.na * (5 + a.classes) > 256 || a.raw_stride < a.na * (5 + a.classes))) return hipErrorInvalidValue;
    if (!a.det && a.reject_below > 0.f && 5 + a.classes <= 128) {
        const size_t total = (size_t)a.n * a.g * a.g * a.na;
        size_t blocks = (total + 255) / 256; if (blocks > 256 * 8) blocks = 256 * 8;
        hipLaunchKernelGGL(k_decode_yolo_lean, dim3((unsigned)blocks), dim3(256), 0, s, a, scores, labels);
        return hipGetLastError();
    }
    if (a.region) {
        size_t total = (size_t)a.n * a.g * a.g * a.na;
        hipLaunchKernelGGL(k_decode_region, dim3((unsigned)((total + 127) / 128)), dim3(128), 0, s, a);
        if (scores) {
            // region heads are tiny (845 rows/image): score them with the generic row kernel
            const size_t rows = (size_t)a.g * a.g * a.na;
            for (int b = 0; b < a.n; ++b)
                launch_score_rows(a.det + ((size_t)b * a.rows_total + a.row_off) * (5 + a.classes), rows, 5 + a.classes,
                                  scores + (size_t)b * a.rows_total + a.row_off, labels + (size_t)b * a.rows_total + a.row_off, s);
        }
    } else {
        const int nch = a.na * (5 + a.classes);
        if (nch <= 256 && a.raw_stride >= nch) {                   // one wave per cell (every shipped topology)
            size_t cells = (size_t)a.n * a.g * a.g;
            size_t blocks = (cells + 3) / 4; if (blocks > 256 * 8) blocks = 256 * 8;    // persistent: 8 workgroups per CU
            if (nch <= 128) hipLaunchKernelGGL(k_decode_yolo_cell<2>, dim3((unsigned)blocks), dim3(256), 0, s, a, scores, labels);
            else hipLaunchKernelGGL(k_decode_yolo_cell<4>, dim3((unsigned)blocks), dim3(256), 0, s, a, scores, labels);
            return hipGetLastError();
        }
        size_t total = (size_t)a.n * a.g * a.g * a.na;             // four boxes per wave per step, grid-stride
        if (5 + a.classes > 128) return hipErrorInvalidValue;
        size_t blocks = (total + 15) / 16; if (blocks > (1u << 20)) blocks = 1u << 20;   // one step per wave: latency-bound otherwise
        hipLaunchKernelGGL(k_decode_yolo, dim3((unsigned)blocks), dim3(256), 0, s, a, scores, labels);
    }
    return hipGetLastError();
}

// ---- S: box_scores = confidence * class_prob; argmax / reduce_max (V3/YOLOV3.py:353-357).  One wave per row,
//      lanes along the attributes (coalesced), shuffle arg-max (first maximum, like argmax). ----
__global__ __launch_bounds__(256) void k_score_rows(const float *det, size_t nrows, int attrs, float *scores, int *labels, int objectness_mode)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t r = wave; r < nrows; r += nwaves) {
        const float *p = det + r * attrs;
        const float obj = p[4];
        float best = -INFINITY; int bi = 0x7fffffff;
        for (int k = lane; k < attrs - 5; k += 64) {
            // objectness_mode (V3/yolo_v3.py:385,397): gate on obj alone, class = argmax of the raw class scores
            float sc = objectness_mode ? p[5 + k] : obj * p[5 + k];
            if (sc > best) { best = sc; bi = k; }
        }
        wave_argmax(best, bi);
        if (lane == 0) { scores[r] = objectness_mode ? obj : best; labels[r] = bi; }
    }
}
void launch_score_rows(const float *det, size_t nrows, int attrs, float *scores, int *labels, hipStream_t s, int objectness_mode)
{
    size_t blocks = (nrows + 3) / 4; if (blocks > 8192) blocks = 8192; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_score_rows, dim3((unsigned)blocks), dim3(256), 0, s, det, nrows, attrs, scores, labels, objectness_mode);
}

// TF NonMaxSuppression IOU on [y0,x0,y1,x1] rows (min/max-normalised corners, 0 when an area <= 0)
__device__ __forceinline__ float iou_tf(float4 a, float4 b)   // a,b = (x0,y0,x1,y1)
{
    float ymin_i = fminf(a.y, a.w), xmin_i = fminf(a.x, a.z), ymax_i = fmaxf(a.y, a.w), xmax_i = fmaxf(a.x, a.z);
    float ymin_j = fminf(b.y, b.w), xmin_j = fminf(b.x, b.z), ymax_j = fmaxf(b.y, b.w), xmax_j = fmaxf(b.x, b.z);
    float area_i = (ymax_i - ymin_i) * (xmax_i - xmin_i);
    float area_j = (ymax_j - ymin_j) * (xmax_j - xmin_j);
    if (area_i <= 0.f || area_j <= 0.f) return 0.f;
    float iy0 = fmaxf(ymin_i, ymin_j), ix0 = fmaxf(xmin_i, xmin_j);
    float iy1 = fminf(ymax_i, ymax_j), ix1 = fminf(xmax_i, xmax_j);
    float inter = fmaxf(iy1 - iy0, 0.f) * fmaxf(ix1 - ix0, 0.f);
    return inter / ((area_i + area_j) - inter);
}
// V2/utils.py:155-174 on int32 pixel boxes: int arithmetic for the areas, float64 for the ratio
__device__ __forceinline__ double iou_v2np(int4 a, int4 b)     // (x0,y0,x1,y1) ints; the reference names them ymin.. but is symmetric
{
    int i0 = max(a.x, b.x), i1 = max(a.y, b.y), i2 = min(a.z, b.z), i3 = min(a.w, b.w);
    double ih = fmax((double)(i2 - i0), 0.), iw = fmax((double)(i3 - i1), 0.);
    double iv = ih * iw;
    int v1 = (a.z - a.x) * (a.w - a.y), v2 = (b.z - b.x) * (b.w - b.y);
    return iv / ((double)(v1 + v2) - iv);
}
// darknet box_iou (DN/box.c:152-182) on (cx,cy,w,h)
__device__ __forceinline__ float dn_overlap(float x1, float w1, float x2, float w2)
{
    float l1 = x1 - w1 / 2, l2 = x2 - w2 / 2;
    float left = l1 > l2 ? l1 : l2;
    float r1 = x1 + w1 / 2, r2 = x2 + w2 / 2;
    float right = r1 < r2 ? r1 : r2;
    return right - left;
}
__device__ __forceinline__ float iou_darknet(float4 a, float4 b)
{
    float w = dn_overlap(a.x, a.z, b.x, b.z), h = dn_overlap(a.y, a.w, b.y, b.w);
    float inter = (w < 0 || h < 0) ? 0.f : w * h;
    float uni = a.z * a.w + b.z * b.w - inter;
    return inter / uni;
}

// `_iou` of the reference's numpy NMS (V3/yolo_v3.py:350-373): float32, NO clamp of a negative overlap, +1e-05
__device__ __forceinline__ float iou_numpy_v3(float4 a, float4 b)
{
    float ix0 = fmaxf(a.x, b.x), iy0 = fmaxf(a.y, b.y), ix1 = fminf(a.z, b.z), iy1 = fminf(a.w, b.w);
    float inter = (ix1 - ix0) * (iy1 - iy0);
    float a1 = (a.z - a.x) * (a.w - a.y), a2 = (b.z - b.x) * (b.w - b.y);
    return inter / (a1 + a2 - inter + 1e-05f);
}

#define NMS_THREADS 1024
#define SORT_LDS 4096

// LDS of k_nms_image, one pool for both paths (u64 units).  General path: sort keys [SORT_LDS] + alive bitset.  Fast path (at most
// NMS_FAST candidates, the common case: a few hundred boxes pass a detection threshold): keys, the sorted candidates themselves and the
// NMS_FAST x NMS_FAST suppression bit matrix.
#define NMS_FAST 512
#define NMS_POOL (NMS_FAST /*keys*/ + 2 * NMS_FAST /*boxes*/ + NMS_FAST / 2 * 3 /*label, score, row*/ + NMS_FAST * (NMS_FAST / 64) /*matrix*/)
static_assert(NMS_POOL >= SORT_LDS + 512, "the general path's keys + alive bitset fit the pool");

__global__ __launch_bounds__(NMS_THREADS) void k_nms_image(const PostArgs a)
{
    float4 *sbox = a.sbox; int *slabel = a.slabel; float *sscore = a.sscore;
    const int img_h = a.img_h, img_w = a.img_w;
    __shared__ unsigned long long pool[NMS_POOL];
    unsigned long long *const skeys = pool;
    unsigned int *const alive = (unsigned int *)(pool + SORT_LDS);      // bitset for up to 32768 candidates (general path)
    __shared__ int wave_cnt[NMS_THREADS / 64];
    __shared__ int s_base, s_cur, s_kept;

    const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float *scores = a.scores + (size_t)img * a.rows;
    const int *labels = a.labels + (size_t)img * a.rows;
    int *cand = a.cand + (size_t)img * a.rows;
    unsigned long long *gkeys = a.keys + (size_t)img * a.rows_pow2;
    sbox += (size_t)img * a.rows; slabel += (size_t)img * a.rows; sscore += (size_t)img * a.rows;
    BoxOut *out = (BoxOut *)a.boxes_out + (size_t)img * a.max_out;

    // unused record slots read as zeros (this replaces a memset node in front of every launch); the kept boxes are written by thread
    // 0 after later barriers
    for (int k = tid; k < a.max_out * (int)(sizeof(BoxOut) / 4); k += NMS_THREADS) ((unsigned *)out)[k] = 0u;
    int *const rows_out = a.rows_out ? a.rows_out + (size_t)img * a.max_out : nullptr;
    int *const srow = a.rows_out ? a.srow + (size_t)img * a.rows : nullptr;      // row of every sorted candidate
    if (rows_out) for (int k = tid; k < a.max_out; k += NMS_THREADS) rows_out[k] = -1;
    if (a.zero_word && img == 0 && tid == 0) *a.zero_word = 0u;      // the lean decode's list counter, for the next step
    // (1) order-preserving compaction of rows whose score passes the threshold (tf.boolean_mask order).  Every wave owns a contiguous run
    // of rows and keeps the ballots of its 64-row steps in LDS: two barriers in all (one per 1024 rows made this phase a third of the
    // kernel: a barrier of sixteen waves costs ~0.4 us)
    const int per_wave = ((a.rows + NMS_THREADS - 1) / NMS_THREADS) * 64;          // rows per wave, a multiple of 64 (<= 2048)
    const int steps = per_wave / 64;                                               // <= 32
    unsigned long long *const wmask = pool + wv * 32;                              // this wave's ballots (the pool is free until the keys are built)
    {
        int cnt = 0;
        const int r_lo = wv * per_wave;
        // eight steps' loads in flight at a time (one load per step, each waited for by its ballot, was a chain of a dozen memory round trips)
        for (int s0 = 0; s0 < steps; s0 += 8) {
            float sv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int r = r_lo + (s0 + u) * 64 + lane; sv[u] = (s0 + u < steps && r < a.rows) ? scores[r] : -INFINITY; }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (s0 + u >= steps) break;
                const int r = r_lo + (s0 + u) * 64 + lane;
                const bool f = r < a.rows && (a.select_mode == 0 ? (sv[u] > a.score_thr) : (sv[u] >= a.score_thr));
                const unsigned long long m = __ballot(f);
                if (lane == 0) wmask[s0 + u] = m;
                cnt += __popcll(m);
            }
        }
        if (lane == 0) wave_cnt[wv] = cnt;
    }
    __syncthreads();
    {
        int pre = 0, tot = 0;
        for (int i = 0; i < NMS_THREADS / 64; ++i) { const int c = wave_cnt[i]; if (i < wv) pre += c; tot += c; }
        const int r_lo = wv * per_wave;
        for (int st = 0; st < steps; ++st) {
            const unsigned long long m = wmask[st];
            if ((m >> lane) & 1ull) cand[pre + __popcll(m & ((1ull << lane) - 1ull))] = r_lo + st * 64 + lane;
            pre += __popcll(m);
        }
        if (tid == 0) s_base = tot;
    }
    __syncthreads();
    int M = s_base;
    if (M <= NMS_FAST && a.nms_mode != 3) {
        // ---- fast path: everything in LDS, five barriers (the general path pays two per kept box and ~40 for its sort).  Phase times at
        //      ~150 candidates, 32 images (kernel trace, one MI355X): launch 4 us, compaction 2.5, keys + rank + gather 5, bit matrix 14, scan 5:
        //      the matrix evaluates every later pair although a kept box is rare -- a lazily evaluated row per kept box is the next step.
        //      Same keys, same gather arithmetic, same IoU functions and the same greedy
        //      order as the general path below: identical records. ----
        unsigned long long *const keys = pool;                                                  // [NMS_FAST]
        float4 *const lbox = (float4 *)(pool + NMS_FAST);                                       // [NMS_FAST] sorted candidates
        int *const llabel = (int *)(pool + 3 * NMS_FAST); float *const lscore = (float *)(llabel + NMS_FAST); int *const lrow = llabel + 2 * NMS_FAST;
        unsigned long long *const mat = pool + 3 * NMS_FAST + NMS_FAST / 2 * 3;                 // [M][W] suppression bits
        // (the barrier above also retired the ballots that shared the pool with the keys)
        if (tid < M) {
            unsigned int u = __float_as_uint(scores[cand[tid]]);
            u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);       // monotone map of float order
            keys[tid] = ((unsigned long long)(~u) << 32) | (unsigned int)tid;      // descending score, ascending candidate index
        }
        __syncthreads();
        // rank sort: the keys are distinct, so a key's position is the number of smaller keys.  Eight lanes share a candidate (each
        // counts an eighth of the keys; three shuffles add the parts up), two passes of the 1024 threads cover 256 candidates
        for (int c0 = 0; c0 < M; c0 += NMS_THREADS / 8) {
            const int ci = c0 + (tid >> 3), part = tid & 7;
            const unsigned long long mykey = ci < M ? keys[ci] : 0ull;
            int rank = 0;
            for (int j = part; j < M; j += 8) rank += keys[j] < mykey ? 1 : 0;
            rank += __shfl_xor(rank, 1); rank += __shfl_xor(rank, 2); rank += __shfl_xor(rank, 4);
            if (ci < M && part == 0) {
                const int row = cand[ci];
                const float *p = a.box4 ? a.box4 + ((size_t)img * a.rows + row) * 4 : a.det + ((size_t)img * a.rows + row) * a.attrs;
                float4 bx;
                if (a.nms_mode == 2 || a.corners_in) bx = float4{p[0], p[1], p[2], p[3]};      // (cx,cy,w,h) for darknet; given corners
                else if (a.nms_mode == 4) { float w2 = 0.5f * p[2], h2 = 0.5f * p[3]; bx = float4{p[0] - h2, p[1] - w2, p[0] + h2, p[1] + w2}; }      // YOLOv1's swapped extents, see below
                else { float w2 = p[2] * 0.5f, h2 = p[3] * 0.5f; bx = float4{p[0] - w2, p[1] - h2, p[0] + w2, p[1] + h2}; }                          // V3/YOLOV3.py:348-351
                if (a.nms_mode == 1 && img_w > 0) {            // V2/utils.py:32-43: scale to the image, truncate to int32, clip
                    int x0 = (int)(bx.x * (float)img_w), y0 = (int)(bx.y * (float)img_h);
                    int x1 = (int)(bx.z * (float)img_w), y1 = (int)(bx.w * (float)img_h);
                    x0 = max(x0, 0); y0 = max(y0, 0); x1 = min(x1, img_w - 1); y1 = min(y1, img_h - 1);
                    bx = float4{(float)x0, (float)y0, (float)x1, (float)y1};
                }
                lbox[rank] = bx; llabel[rank] = labels[row]; lscore[rank] = scores[row]; lrow[rank] = row;
            }
        }
        __syncthreads();
        if (a.nms_mode == 1 && M > 400) M = 400;           // V2 numpy flavour keeps only the 400 best before NMS (bboxes_sort top_k, V2/utils.py:146-151)
        const int W = (M + 63) >> 6;
        // The flavours that stop at max_out kept boxes (TF's, YOLOv1's) need the suppression row of a KEPT box only, and kept boxes are
        // few: one wave walks the candidates in order and forms the row of each box it keeps on the spot -- lane b evaluates the pair
        // (i, 64 w + b), the ballot is the word -- instead of all M W words up front (14 of the kernel's 30 us for ~150 candidates, of which
        // 20 rows were ever read).  Same pairs, same IoU function, same greedy order: identical records.
        if ((a.nms_mode == 0 || a.nms_mode == 4) && a.max_out <= 64) {
            if (wv == 0) {
                unsigned long long al = 0;                   // lane w holds word w of the alive set
                if (lane < W) { const int lo = lane * 64; al = (M - lo >= 64) ? ~0ull : ((1ull << (M - lo)) - 1ull); }
                int kept = 0;
                for (int w = 0; w < W && kept < a.max_out; ++w) {
                    while (kept < a.max_out) {
                        const unsigned long long cw = __shfl(al, w);      // (wave-uniform)
                        if (!cw) break;
                        const int i = 64 * w + (int)__builtin_ctzll(cw);
                        const float4 bc = lbox[i];
                        if (lane == 0) {
                            out[kept] = BoxOut{bc.x, bc.y, bc.z, bc.w, lscore[i], llabel[i]};
                            if (rows_out) rows_out[kept] = lrow[i];
                        }
                        ++kept;
                        for (int ww = w; ww < W; ++ww) {
                            const int j = 64 * ww + lane;
                            const bool kill = j > i && j < M && iou_tf(bc, lbox[j < M ? j : i]) > a.iou_thr;
                            unsigned long long bits = __ballot(kill);
                            if (ww == w) bits |= 1ull << (i & 63);
                            if (lane == ww) al &= ~bits;
                        }
                    }
                }
                if (lane == 0) a.counts_out[img] = kept;
            }
            return;
        }
        // suppression bits: word (i, w) bit b set <=> candidate i, when kept, removes the later candidate j = 64 w + b.  One wave per
        // word: lane b evaluates the pair (i, 64 w + b), the ballot IS the word
        for (int e = wv; e < M * W; e += NMS_THREADS / 64) {
            const int i = e / W, w = e - i * W;
            unsigned long long bits = 0;
            if (64 * w + 63 > i) {                           // (wave-uniform)
                const int j = 64 * w + lane;
                bool kill = false;
                if (j > i && j < M) {
                    const float4 bc = lbox[i], bj = lbox[j];
                    if (a.nms_mode == 0 || a.nms_mode == 4) kill = iou_tf(bc, bj) > a.iou_thr;
                    else if (a.nms_mode == 2) kill = llabel[j] == llabel[i] && iou_darknet(bc, bj) > a.iou_thr;
                    else {
                        const double v = iou_v2np(int4{(int)bc.x, (int)bc.y, (int)bc.z, (int)bc.w}, int4{(int)bj.x, (int)bj.y, (int)bj.z, (int)bj.w});
                        kill = llabel[j] == llabel[i] && !(v < (double)a.iou_thr);
                    }
                }
                bits = __ballot(kill);
            }
            if (lane == 0) mat[e] = bits;
        }
        __syncthreads();
        // greedy scan by ONE wave: lane w holds word w of the alive set
        if (wv == 0) {
            unsigned long long al = 0;
            if (lane < W) { const int lo = lane * 64; al = (M - lo >= 64) ? ~0ull : ((1ull << (M - lo)) - 1ull); }
            int kept = 0;
            const bool capped = a.nms_mode == 0 || a.nms_mode == 4;
            for (int w = 0; w < W; ++w) {
                while (true) {
                    const unsigned long long cw = __shfl(al, w);          // (wave-uniform)
                    if (!cw || (capped && kept >= a.max_out)) break;
                    const int i = 64 * w + (int)__builtin_ctzll(cw);
                    if (kept < a.max_out && lane == 0) {
                        const float4 bc = lbox[i];
                        out[kept] = BoxOut{bc.x, bc.y, bc.z, bc.w, lscore[i], llabel[i]};
                        if (rows_out) rows_out[kept] = lrow[i];
                    }
                    ++kept;
                    unsigned long long kill = lane < W ? mat[i * W + lane] : 0ull;
                    if (lane == w) kill |= 1ull << (i & 63);
                    al &= ~kill;
                }
                if (capped && kept >= a.max_out) break;
            }
            if (lane == 0) a.counts_out[img] = min(kept, a.max_out);
        }
        return;
    }
    // V2 numpy flavour keeps only the 400 best before NMS (bboxes_sort top_k, V2/utils.py:146-151)
    int P = 1; while (P < M) P <<= 1;
    const bool in_lds = P <= SORT_LDS;
    unsigned long long *keys = in_lds ? skeys : gkeys;

    // (2) keys: descending score, ascending candidate index
    for (int i = tid; i < P; i += NMS_THREADS) {
        unsigned long long k = ~0ull;
        if (i < M) {
            unsigned int u = __float_as_uint(scores[cand[i]]);
            u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);       // monotone map of float order
            k = ((unsigned long long)(~u) << 32) | (unsigned int)i;
            if (a.nms_mode == 3)       // per class (ascending), then objectness descending, then candidate index
                k = ((unsigned long long)(labels[cand[i]] & 0x3ff) << 47) | ((unsigned long long)(~u) << 15) | (unsigned int)(i & 0x7fff);
        }
        keys[i] = k;
    }
    __syncthreads();
    // (3) bitonic sort, ascending
    for (int k = 2; k <= P; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < P; i += NMS_THREADS) {
                int x = i ^ j;
                if (x > i) {
                    unsigned long long ki = keys[i], kx = keys[x];
                    bool up = (i & k) == 0;
                    if ((ki > kx) == up) { keys[i] = kx; keys[x] = ki; }
                }
            }
            __syncthreads();
        }
    if (a.nms_mode == 1 && M > 400) M = 400;
    // (4) gather candidates in sorted order
    for (int i = tid; i < M; i += NMS_THREADS) {
        int row = cand[(unsigned int)(keys[i] & (a.nms_mode == 3 ? 0x7fffu : 0xffffffffu))];
        const float *p = a.box4 ? a.box4 + ((size_t)img * a.rows + row) * 4 : a.det + ((size_t)img * a.rows + row) * a.attrs;
        float4 b;
        if (a.nms_mode == 2 || a.corners_in) b = float4{p[0], p[1], p[2], p[3]};       // (cx,cy,w,h) for darknet; given corners
        else if (a.nms_mode == 4) {
            // YOLOv1's quirk (V1/YOLO_V1_Inference.py:259-262): `_boxes` = [y - w/2, x - h/2, y + w/2, x + h/2] handed over as
            // [ymin, xmin, ymax, xmax] -- the vertical extent is built from the WIDTH and the horizontal one from the HEIGHT
            float w2 = 0.5f * p[2], h2 = 0.5f * p[3];
            b = float4{p[0] - h2, p[1] - w2, p[0] + h2, p[1] + w2};
        } else {
            float w2 = p[2] * 0.5f, h2 = p[3] * 0.5f;                                      // V3/YOLOV3.py:348-351
            b = float4{p[0] - w2, p[1] - h2, p[0] + w2, p[1] + h2};
        }
        if (a.nms_mode == 1 && img_w > 0) {
            // V2/utils.py:32-43: scale to the image, truncate to int32, clip to [0, w-1] x [0, h-1]
            int x0 = (int)(b.x * (float)img_w), y0 = (int)(b.y * (float)img_h);
            int x1 = (int)(b.z * (float)img_w), y1 = (int)(b.w * (float)img_h);
            x0 = max(x0, 0); y0 = max(y0, 0); x1 = min(x1, img_w - 1); y1 = min(y1, img_h - 1);
            b = float4{(float)x0, (float)y0, (float)x1, (float)y1};
        }
        sbox[i] = b; slabel[i] = labels[row]; sscore[i] = scores[row];
        if (srow) srow[i] = row;
    }
    for (int i = tid; i < 1024; i += NMS_THREADS) {
        int lo = i * 32;
        unsigned int bits = 0;
        if (lo < M) bits = (M - lo >= 32) ? 0xffffffffu : ((1u << (M - lo)) - 1u);
        alive[i] = bits;
    }
    if (tid == 0) { s_cur = -1; s_kept = 0; }
    __syncthreads();

    if (a.nms_mode == 3) {
        // Reference numpy NMS (V3/yolo_v3.py:376-420), class by class over the sorted list.  The reference filters
        // `cls_scores` with indices taken on `cls_boxes[1:]` (:414-418), so after every round each survivor
        // inherits the score of the element that preceded it in the current list -- reproduced here.
        __syncthreads();
        int *list = cand;                               // compacted alive positions (cand is free after the gather)
        float *tmp = (float *)gkeys;                    // scratch for the shifted scores (keys are free as well)
        __shared__ int s_seg_end;
        int seg = 0;
        while (seg < M) {
            if (tid == 0) { int e = seg + 1; const int lc = slabel[seg]; while (e < M && slabel[e] == lc) ++e; s_seg_end = e; }
            __syncthreads();
            const int seg_end = s_seg_end;
            while (true) {
                // compact the alive positions of this class segment, in order
                if (tid == 0) s_base = 0;
                __syncthreads();
                for (int r0 = seg; r0 < seg_end; r0 += NMS_THREADS) {
                    int j = r0 + tid;
                    bool f = j < seg_end && ((alive[j >> 5] >> (j & 31)) & 1u);
                    unsigned long long m = __ballot(f);
                    if (lane == 0) wave_cnt[wv] = __popcll(m);
                    __syncthreads();
                    int pre = 0, tot = 0;
                    for (int i = 0; i < NMS_THREADS / 64; ++i) { int c = wave_cnt[i]; if (i < wv) pre += c; tot += c; }
                    int base = s_base;
                    if (f) list[base + pre + __popcll(m & ((1ull << lane) - 1))] = j;
                    __syncthreads();
                    if (tid == 0) s_base = base + tot;
                    __syncthreads();
                }
                const int len = s_base;
                if (len == 0) break;
                const int head = list[0];
                const float4 bh = sbox[head];
                const int kept = s_kept;
                if (tid == 0) {
                    if (kept < a.max_out) { out[kept] = BoxOut{bh.x, bh.y, bh.z, bh.w, sscore[head], slabel[head]}; if (rows_out) rows_out[kept] = srow[head]; }
                    s_kept = kept + 1;
                    atomicAnd(&alive[head >> 5], ~(1u << (head & 31)));
                }
                for (int m2 = 1 + tid; m2 < len; m2 += NMS_THREADS) {
                    const int j = list[m2];
                    if (iou_numpy_v3(bh, sbox[j]) < a.iou_thr) tmp[j] = sscore[list[m2 - 1]];     // survivor: shifted score
                    else atomicAnd(&alive[j >> 5], ~(1u << (j & 31)));
                }
                __syncthreads();
                for (int m2 = 1 + tid; m2 < len; m2 += NMS_THREADS) {
                    const int j = list[m2];
                    if ((alive[j >> 5] >> (j & 31)) & 1u) sscore[j] = tmp[j];
                }
                __syncthreads();
            }
            seg = seg_end;
            __syncthreads();
        }
        if (tid == 0) a.counts_out[img] = min(s_kept, a.max_out);
        return;
    }

    // (5) greedy: the best alive candidate is kept and suppresses every later one it overlaps
    int pos = 0;
    while (true) {
        if (tid == 0) {
            int cur = -1;
            for (int wd = pos >> 5; wd < ((M + 31) >> 5); ++wd) {
                unsigned int bits = alive[wd];
                if (wd == (pos >> 5)) bits &= ~((1u << (pos & 31)) - 1u);
                if (bits) { cur = wd * 32 + __ffs(bits) - 1; break; }
            }
            s_cur = cur;
        }
        __syncthreads();
        const int cur = s_cur, kept = s_kept;
        if (cur < 0 || ((a.nms_mode == 0 || a.nms_mode == 4) && kept >= a.max_out)) break;
        const float4 bc = sbox[cur];
        const int lc = slabel[cur];
        if (tid == 0) {
            if (kept < a.max_out) { out[kept] = BoxOut{bc.x, bc.y, bc.z, bc.w, sscore[cur], lc}; if (rows_out) rows_out[kept] = srow[cur]; }
            s_kept = kept + 1;
        }
        for (int j = cur + 1 + tid; j < M; j += NMS_THREADS) {
            if (!((alive[j >> 5] >> (j & 31)) & 1u)) continue;
            bool kill;
            if (a.nms_mode == 0 || a.nms_mode == 4) kill = iou_tf(bc, sbox[j]) > a.iou_thr;
            else if (a.nms_mode == 2) kill = slabel[j] == lc && iou_darknet(bc, sbox[j]) > a.iou_thr;
            else {
                float4 bj = sbox[j];
                double v = iou_v2np(int4{(int)bc.x, (int)bc.y, (int)bc.z, (int)bc.w}, int4{(int)bj.x, (int)bj.y, (int)bj.z, (int)bj.w});
                kill = slabel[j] == lc && !(v < (double)a.iou_thr);
            }
            if (kill) atomicAnd(&alive[j >> 5], ~(1u << (j & 31)));
        }
        pos = cur + 1;
        __syncthreads();
    }
    if (tid == 0) a.counts_out[img] = min(s_kept, a.max_out);
}

hipError_t launch_postprocess(const PostArgs &a, hipStream_t s)
{
    size_t nrows = (size_t)a.n * a.rows;
    if (!a.scores_ready) launch_score_rows(a.det, nrows, a.attrs, a.scores, a.labels, s, a.nms_mode == 3);
    hipLaunchKernelGGL(k_nms_image, dim3(a.n), dim3(NMS_THREADS), 0, s, a);
    return hipGetLastError();
}

// ---- X: `detections_boxes` (V3/yolo_v3.py:329-347): (cx, cy, w, h, rest...) -> (x0, y0, x1, y1, rest...), w/2 form ----
__global__ void k_boxes_to_corners(const float *in, float *out, size_t nrows, int attrs)
{
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nrows * attrs) return;
    size_t r = idx / attrs; int k = (int)(idx - r * attrs);
    const float *p = in + r * attrs;
    float v;
    if (k == 0) v = p[0] - p[2] / 2.0f;
    else if (k == 1) v = p[1] - p[3] / 2.0f;
    else if (k == 2) v = p[0] + p[2] / 2.0f;
    else if (k == 3) v = p[1] + p[3] / 2.0f;
    else v = p[k];
    out[idx] = v;
}
hipError_t launch_boxes_to_corners(const float *in, float *out, size_t nrows, int attrs, hipStream_t s)
{
    size_t total = nrows * attrs;
    hipLaunchKernelGGL(k_boxes_to_corners, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, out, nrows, attrs);
    return hipGetLastError();
}

// ---- darknet do_nms_sort / do_nms_obj (DN/box.c:21-89) on caller-shaped arrays: boxes [n] (cx,cy,w,h), prob [n][classes],
//      objectness [n]; suppressed entries are zeroed in place.  One workgroup per class (sort) or one in all (obj):
//      bitonic sort of (score, index) keys in LDS (descending; ties: lower index first -- qsort leaves tie order
//      unspecified), then the reference's greedy pass with the inner loop spread over the threads.  Detections with
//      objectness == 0 do not take part (the reference moves them behind `total`). ----
#define NMSD_MAX 4096
__global__ __launch_bounds__(1024) void k_nms_dets(const float4 *boxes, float *prob, float *objectness, int n, int classes, float thresh, int by_obj)
{
    __shared__ unsigned long long key[NMSD_MAX];
    __shared__ unsigned char dead[NMSD_MAX];
    const int k = by_obj ? -1 : (int)blockIdx.x;
    const int tid = threadIdx.x, nt = blockDim.x;
    int np2 = 1; while (np2 < n) np2 <<= 1;
    for (int j = tid; j < np2; j += nt) {
        float sc = 0.f;
        if (j < n && objectness[j] != 0.f) sc = by_obj ? objectness[j] : prob[(size_t)j * classes + k];
        // descending order == ascending order of the complemented key; zero / padded entries sort last
        const unsigned bits = sc > 0.f ? __float_as_uint(sc) : 0u;
        key[j] = ((unsigned long long)(~bits) << 32) | (unsigned)j;
        dead[j] = !(j < n && objectness[j] != 0.f);
    }
    __syncthreads();
    for (int size = 2; size <= np2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < np2 / 2; t += nt) {
                const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
                const bool up = (lo & size) == 0;
                const unsigned long long a = key[lo], b = key[hi];
                if ((a > b) == up) { key[lo] = b; key[hi] = a; }
            }
            __syncthreads();
        }
    // greedy pass over the sorted order (every condition below is uniform across the workgroup: it only reads shared /
    // global state that was last written before a barrier).  `dead` is indexed by ORIGINAL detection index.
    for (int i = 0; i < n; ++i) {
        const unsigned ji = (unsigned)(key[i] & 0xffffffffu);
        if (ji >= (unsigned)n) break;                                   // only padding from here on
        const float si = dead[ji] ? 0.f : (by_obj ? objectness[ji] : prob[(size_t)ji * classes + k]);
        if (si != 0.f) {
            const float4 a = boxes[ji];
            for (int jj = i + 1 + tid; jj < n; jj += nt) {
                const unsigned j = (unsigned)(key[jj] & 0xffffffffu);
                if (j >= (unsigned)n || dead[j]) continue;
                if (by_obj && objectness[j] == 0.f) continue;
                if (iou_darknet(a, boxes[j]) > thresh) {
                    if (by_obj) { objectness[j] = 0.f; for (int c = 0; c < classes; ++c) prob[(size_t)j * classes + c] = 0.f; }
                    else prob[(size_t)j * classes + k] = 0.f;
                }
            }
        }
        __syncthreads();
    }
}
hipError_t launch_nms_dets(const float4 *boxes, float *prob, float *objectness, int n, int classes, float thresh, int by_obj, hipStream_t s)
{
    if (n < 1) return hipSuccess;
    if (n > NMSD_MAX) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_nms_dets, dim3(by_obj ? 1 : classes), dim3(1024), 0, s, boxes, prob, objectness, n, classes, thresh, by_obj);
    return hipGetLastError();
}

// ---- darknet's get_network_boxes on the device (DN/network.c:536-567): num_detections + fill_network_boxes over the decoded rows
//      of one image.  [yolo] heads (get_yolo_detections, DN/yolo_layer.c:316-343): rows with objectness > thresh, in head / cell /
//      anchor order, prob[j] = objectness * class_j gated by thresh.  [region] heads without a tree (get_region_detections,
//      DN/region_layer.c:364-437): every box, anchor-major, objectness and probabilities gated by thresh.  Then
//      correct_yolo_boxes / correct_region_boxes (DN/yolo_layer.c:247-273 = DN/region_layer.c:336-362: undo the letterbox).  The
//      box arithmetic below restates those lines operation for operation (the reference mixes float and double there, and the
//      veneer is compared with it to the last bits) -- this file is built with -ffp-contract=off.
//      One workgroup: an ordered compaction of ~10^4 rows is latency-, not bandwidth-bound. ----
__global__ __launch_bounds__(1024) void k_darknet_boxes(const DnBoxesArgs a)
{
    __shared__ int s_wave[16];
    __shared__ int s_base;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int hd = 0; hd < a.nheads; ++hd) {
        const int cells = a.grid[hd] * a.grid[hd], rows = cells * a.na[hd];
        if (a.kind[hd] == 2) {
            // [detection]: no selection, no reordering -- box i * num + n of the layer is record i * num + n
            const int base = s_base;
            for (int r = tid; r < rows; r += 1024) if (base + r < a.cap) a.src[base + r] = a.off[hd] + r;
            __syncthreads();
            if (tid == 0) s_base = base + rows;
            __syncthreads();
        } else if (a.kind[hd] == 1) {
            const int base = s_base;
            for (int r = tid; r < rows; r += 1024) {
                const int i = r / a.na[hd], n = r - i * a.na[hd];
                const int pos = base + n * cells + i;              // index = n * w * h + i (DN/region_layer.c:395)
                if (pos < a.cap) a.src[pos] = a.off[hd] + r;
            }
            __syncthreads();
            if (tid == 0) s_base = base + rows;
            __syncthreads();
        } else {
            for (int r0 = 0; r0 < rows; r0 += 1024) {
                const int r = r0 + tid;
                const bool keep = r < rows && a.det[(size_t)(a.off[hd] + r) * a.attrs + 4] > a.thresh;
                const unsigned long long m = __ballot(keep);
                if (lane == 0) s_wave[wv] = __popcll(m);
                __syncthreads();
                int before = s_base, total = 0;
                for (int k = 0; k < 16; ++k) { if (k < wv) before += s_wave[k]; total += s_wave[k]; }
                if (keep) { const int pos = before + __popcll(m & ((1ull << lane) - 1ull)); if (pos < a.cap) a.src[pos] = a.off[hd] + r; }
                __syncthreads();
                if (tid == 0) s_base += total;
                __syncthreads();
            }
        }
    }
    const int count = s_base;
    if (tid == 0) *a.count = count;
    if (!a.rec) return;
    __threadfence_block();
    const int n = count < a.cap ? count : a.cap;
    const int netw = a.netw, neth = a.neth, w = a.w, h = a.h;
    int new_w, new_h;
    if (((float)netw / w) < ((float)neth / h)) { new_w = netw; new_h = (h * netw) / w; } else { new_h = neth; new_w = (w * neth) / h; }
    // which head a row belongs to decides the gating
    for (long idx = tid; idx < (long)n * a.attrs; idx += 1024) {
        const int rec = (int)(idx / a.attrs), k = (int)(idx - (long)rec * a.attrs);
        const int row = a.src[rec];
        int kind = 0;
        for (int hd = 0; hd < a.nheads; ++hd) if (row >= a.off[hd]) kind = a.kind[hd];
        if (kind == 2) {
            // operation for operation get_detection_detections: (pred + col) / side * w in float, pow(pred, 2) * w through double
            const int hd0 = 0, r = row - a.off[hd0], nb = a.na[hd0], i = r / nb, n = r - i * nb, side = a.side;
            const float *pr = a.raw;
            const float scale = pr[side * side * a.classes + i * nb + n];
            const float *bx = pr + side * side * (a.classes + nb) + (i * nb + n) * 4;
            float v;
            if (k == 0) v = (bx[0] + (i % side)) / side * w;
            else if (k == 1) v = (bx[1] + (i / side)) / side * h;
            else if (k == 2) v = (float)(pow((double)bx[2], a.sqr ? 2 : 1) * w);
            else if (k == 3) v = (float)(pow((double)bx[3], a.sqr ? 2 : 1) * h);
            else if (k == 4) v = scale;
            else { const float prob = scale * pr[i * a.classes + (k - 5)]; v = prob > a.thresh ? prob : 0.f; }
            a.rec[idx] = v;
            continue;
        }
        const float *p = a.det + (size_t)row * a.attrs;
        const float obj_raw = p[4];
        const float objectness = kind == 1 ? (obj_raw > a.thresh ? obj_raw : 0.f) : obj_raw;
        float v;
        if (k == 0) { v = (p[0] - (netw - new_w) / 2. / netw) / ((float)new_w / netw); if (!a.relative) v *= w; }
        else if (k == 1) { v = (p[1] - (neth - new_h) / 2. / neth) / ((float)new_h / neth); if (!a.relative) v *= h; }
        else if (k == 2) { v = p[2]; v *= (float)netw / new_w; if (!a.relative) v *= w; }
        else if (k == 3) { v = p[3]; v *= (float)neth / new_h; if (!a.relative) v *= h; }
        else if (k == 4) v = objectness;
        else { const float prob = obj_raw * p[k]; v = (objectness != 0.f && prob > a.thresh) ? prob : 0.f; }
        a.rec[idx] = v;
    }
}
hipError_t launch_darknet_boxes(const DnBoxesArgs &a, hipStream_t s)
{
    if (a.nheads < 1 || a.nheads > 8) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_darknet_boxes, dim3(1), dim3(1024), 0, s, a);
    return hipGetLastError();
}

// ---- the last layer's output in darknet's own layout, what `network_predict` returns (DN/network.c:497-508 -> net->output):
//      a [yolo] layer's output is its input with the logistic applied to x, y, objectness and the class scores
//      (DN/yolo_layer.c:143-152), a [region] layer's (softmax form) the logistic on x, y, objectness and a softmax over the classes
//      (DN/region_layer.c:163-200), both planar [anchor * (5 + classes) + attr][cell].  logistic = 1./(1. + exp(-x)) evaluated in
//      double like DN/activations.h:38. ----
__global__ void k_head_darknet_layout(const float *raw, int raw_stride, int cells, int na, int classes, int region, float *out)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= cells * na) return;
    const int cell = r / na, n = r - cell * na, attrs = 5 + classes;
    const float *p = raw + (size_t)cell * raw_stride + n * attrs;
    float *o = out + (size_t)n * attrs * cells + cell;
    auto lg = [](float x) { return (float)(1. / (1. + exp(-(double)x))); };
    o[0] = lg(p[0]); o[(size_t)cells] = lg(p[1]); o[(size_t)2 * cells] = p[2]; o[(size_t)3 * cells] = p[3]; o[(size_t)4 * cells] = lg(p[4]);
    if (!region) {
        for (int k = 0; k < classes; ++k) o[(size_t)(5 + k) * cells] = lg(p[5 + k]);
    } else {
        // softmax (DN/blas.c:305-321, temperature 1): largest first, exp(x - largest), normalised by the running sum
        float largest = -3.402823466e+38f;
        for (int k = 0; k < classes; ++k) if (p[5 + k] > largest) largest = p[5 + k];
        float sum = 0.f;
        for (int k = 0; k < classes; ++k) { const float e = (float)exp((double)(p[5 + k] - largest)); sum += e; o[(size_t)(5 + k) * cells] = e; }
        for (int k = 0; k < classes; ++k) o[(size_t)(5 + k) * cells] /= sum;
    }
}
hipError_t launch_head_darknet_layout(const float *raw, int raw_stride, int cells, int na, int classes, int region, float *out, hipStream_t s)
{
    const int rows = cells * na;
    hipLaunchKernelGGL(k_head_darknet_layout, dim3((rows + 255) / 256), dim3(256), 0, s, raw, raw_stride, cells, na, classes, region, out);
    return hipGetLastError();
}
