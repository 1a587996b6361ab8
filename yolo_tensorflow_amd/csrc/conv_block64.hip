// Fused FIRST residual block of darknet-53 for gfx950 (V3/yolo_v3.py:54-60 `_darknet53_block(inputs, 32)`, DN cfg layers 2-4; shapes
// V3/yolov3.txt:4-9):
//     y = x + leaky(conv3x3(leaky(conv1x1(x))))          x, y: [N, H, W, 64],  1x1: 64 -> 32,  3x3: 32 -> 64        (H = W = 208 at 416 x 416)
// Run layer by layer at 416 x 416 x 32 the stem's 1x1 tail writes its 88 MB tensor, the 3x3 (conv_halo_c32_c64) reads it back, reads x again
// as the shortcut and writes y: 443 MB for 57 GFLOP, the one launch of the network that sits on HBM (round 4: 4.7 TB/s, 104.6 us).  Here the
// 32-channel tensor never exists: a workgroup owns one 13 x 13 block of output pixels at a time, the 15 x 15 x 64 halo tile of x comes into
// LDS by LDS-DMA, the 1x1 turns it into the 15 x 15 x 32 tile the 3x3 needs (rounded exactly as the unfused layer stores it, zeros where
// the 3x3 pads), the 3x3 contracts that with its filters held in registers, and the shortcut is the interior of the x tile that is still
// in LDS -- x in once, y out once: 354 MB.  Bit-identical to the separate launches (same roundings, same K order: one MFMA per tap).
//
// What is different from conv_resblock_c128 (conv_block.hip), and why: that kernel is issue-slot-bound with ONE 162 KB workgroup per CU --
// nothing runs under its staging, epilogue and barriers.  Half the channels make everything a quarter the size (x tile 30 KB, mid tile
// 23 KB, 36 filter registers per wave), so TWO workgroups fit a CU (61 KB of LDS and <= 128 VGPRs each) and one's fetch / staging / stores
// run under the other's MFMAs; the x tile is single-buffered (the co-resident workgroup is the latency cover) and stays valid to the end of
// the block, which is what lets the shortcut come from LDS.
// MEASURED (416 x 416 x 32, bf16, per-layer events): 130 us as first written, 118 us with the interior-block fast path below (addresses = per-lane
// constants + one scalar soffset per block), against 108 us for conv_halo_c32_c64, the launch it replaces; the stem is no faster without its
// 1x1 tail.  ~650-800 instructions per wave and block for 62 MFMAs, five barriers: instruction-issue-bound.  Hence OPT-IN (YOLO_RESBLOCK64=1).
//
//   waves    8 = 4 channel groups (16 output channels of the 3x3) x 2 pixel halves (sub-tiles 0-5 / 6-10 of the block's 11)
//   stage 1  wave u takes halo-pixel sub-tiles 2u, 2u+1: two K-steps of (2 pixel fragments, 2 filter fragments from LDS, 4 MFMAs)
//   stage 2  per (tap, half of the wave's sub-tiles) 3 ds_read_b128 at precomputed addresses (the tap is an immediate offset) and 3 MFMAs; the
//            fragments are not double-buffered: 128 registers per wave, the SIMD's other three waves cover the LDS latency
//   epilogue bias, leaky, rounding -> LDS (over the dead mid tile) -> 16-byte pieces + the x piece of the same pixel from the x tile,
//            rounded once more (as the separate shortcut kernel would) -> global; the next block's x tile is requested before the stores
// LDS layouts: x tile 128-byte pixel rows, 16-byte slot XOR (pixel & 7) (applied to the SOURCE chunk of the lane-linear LDS-DMA):
// conflict-free for stage 1's fragment reads; mid tile 96-byte rows (64 B of channels + 32 B pad: a ds_read_b128 lane group's two 16-byte
// columns fall on even / odd slots and 8 consecutive pixels on distinct ones); staged output 144-byte rows.
#include "kernels.h"
#include <type_traits>

namespace {
typedef __bf16 b6_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 b6_f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 b6_bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 b6_f16x2 __attribute__((ext_vector_type(2)));
typedef float b6_f32x2 __attribute__((ext_vector_type(2)));
typedef float b6_f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t b6_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void b6_lds_void;
typedef __attribute__((address_space(3))) char b6_lds_char;

template <bool H16> __device__ __forceinline__ uint32_t b6_pk(float lo, float hi)
{
    if constexpr (H16) return __builtin_bit_cast(uint32_t, __builtin_convertvector(b6_f32x2{lo, hi}, b6_f16x2));      // (MODE.FP16_OVFL: an overflowing conversion saturates at +-65504)
    else return __builtin_bit_cast(uint32_t, __builtin_convertvector(b6_f32x2{lo, hi}, b6_bf16x2));
}
template <bool H16> __device__ __forceinline__ float b6_lo(uint32_t w) { if constexpr (H16) return (float)__builtin_bit_cast(b6_f16x2, w)[0]; else return __builtin_bit_cast(float, w << 16); }
template <bool H16> __device__ __forceinline__ float b6_hi(uint32_t w) { if constexpr (H16) return (float)__builtin_bit_cast(b6_f16x2, w)[1]; else return __builtin_bit_cast(float, w & 0xffff0000u); }
__device__ __forceinline__ float b6_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// acc + bias, activation (slope 0.1: leaky as max(v, 0.1 v); slope 1: linear), rounded to the storage type: four channels as two packed words
template <bool H16> __device__ __forceinline__ uint2 b6_epi(const b6_f32x4 acc, const b6_f32x4 bias, const float slope)
{
    b6_f32x4 v = acc + bias;
    const b6_f32x4 t = v * slope;
    return uint2{b6_pk<H16>(b6_max(v[0], t[0]), b6_max(v[1], t[1])), b6_pk<H16>(b6_max(v[2], t[2]), b6_max(v[3], t[3]))};
}
template <bool H16> __device__ __forceinline__ b6_f32x4 b6_mma(const b6_bf16x8 a, const b6_bf16x8 b, const b6_f32x4 c)
{
    if constexpr (H16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(b6_f16x8, a), __builtin_bit_cast(b6_f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

constexpr int B6_B = 13;                               // output block edge
constexpr int B6_T = B6_B + 2;                         // halo tile edge: 15
constexpr int B6_TPIX = B6_T * B6_T;                   // 225
constexpr int B6_TSUB = (B6_TPIX + 15) / 16;           // 15 sub-tiles of 16 halo pixels (240 rows are allocated)
constexpr int B6_OPIX = B6_B * B6_B;                   // 169
constexpr int B6_OSUB = (B6_OPIX + 15) / 16;           // 11
constexpr int B6_C = 64, B6_M = 32;                    // block channels, mid channels
constexpr int B6_NW = 8;                               // waves per workgroup: 4 channel groups x 2 pixel halves
constexpr int B6_NJ = 6;                               // pixel sub-tiles of one wave in the 3x3 (half 0: 0..5, half 1: 6..10 and one idle slot)
constexpr int B6_XPITCH = B6_C * 2;                    // 128
constexpr int B6_X_BYTES = B6_TSUB * 16 * B6_XPITCH;   // 30720
constexpr int B6_MPITCH = 96;                          // mid tile pixel rows: 64 B of channels + 32 B of padding (see the header)
constexpr int B6_MID_BYTES = B6_TSUB * 16 * B6_MPITCH; // 23040
constexpr int B6_OPITCH = B6_C * 2 + 16;               // staged output rows: 144
constexpr int B6_OUT_BYTES = B6_OSUB * 16 * B6_OPITCH; // 25344, over the mid tile (dead once every wave has left stage 2)
constexpr int B6_MO_BYTES = B6_OUT_BYTES > B6_MID_BYTES ? B6_OUT_BYTES : B6_MID_BYTES;
constexpr int B6_W1PITCH = B6_C * 2 + 16;              // 1x1 filter rows in LDS, padded: 144
constexpr int B6_W1_BYTES = B6_M * B6_W1PITCH;         // 4608
constexpr int B6_B1_BYTES = B6_M * 4, B6_B2_BYTES = B6_C * 4;
constexpr int B6_LDS = B6_X_BYTES + B6_MO_BYTES + B6_W1_BYTES + B6_B1_BYTES + B6_B2_BYTES;      // 61056: two workgroups per CU
constexpr int B6_NPIECE = (B6_OPIX * 8 + 64 * B6_NW - 1) / (64 * B6_NW);                         // 16-byte pieces of the output block per thread: 3
static_assert(2 * B6_LDS <= 160 * 1024, "two workgroups per CU");
}

template <bool H16>
__global__ __launch_bounds__(64 * B6_NW, 4) void conv_resblock_c64(const BlockArgs a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if constexpr (H16) __builtin_amdgcn_s_setreg((0 << 11) | (23 << 6) | 1, 1);      // hwreg(HW_REG_MODE, 23, 1) = FP16_OVFL
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const int wc = wave & 3, wp = wave >> 2;             // channel group (16 output channels of the 3x3), pixel half
    char *const lx_ = smem, *const lmo_ = lx_ + B6_X_BYTES, *const lw1_ = lmo_ + B6_MO_BYTES, *const lb1_ = lw1_ + B6_W1_BYTES, *const lb2_ = lb1_ + B6_B1_BYTES;
    const float slope1 = a.act1 == ACT_LEAKY ? 0.1f : 1.f, slope2 = a.act2 == ACT_LEAKY ? 0.1f : 1.f;
    const int bx = (a.W + B6_B - 1) / B6_B, by = (a.H + B6_B - 1) / B6_B, per_img = bx * by, nblocks = a.N * per_img;      // (ragged blocks on the bottom / right edge: their pixels past the image are computed on zeros and never stored)
    const int nt = (nblocks - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;

    // ---- once per workgroup: 1x1 filters -> LDS; this wave's slice of the 3x3 filters -> registers; biases -> LDS ----
    for (int g = tid; g < B6_M * 8; g += 64 * B6_NW) {
        const int row = g >> 3, piece = g & 7;
        *(uint4 *)(lw1_ + row * B6_W1PITCH + piece * 16) = *(const uint4 *)((const bf16_t *)a.w1 + (size_t)row * a.Kpad1 + piece * 8);
    }
    b6_bf16x8 fw2[9];                                    // [tap]: K = tap * 32 + lq * 8 ..
#pragma unroll
    for (int t = 0; t < 9; ++t)
        fw2[t] = *(const b6_bf16x8 *)((const bf16_t *)a.w2 + (size_t)(wc * 16 + l15) * a.Kpad2 + t * 32 + lq * 8);
    if (tid < B6_M) *(float *)(lb1_ + tid * 4) = a.b1[tid];
    if (tid < B6_C) *(float *)(lb2_ + tid * 4) = a.b2[tid];

    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, 0x80000000u, 0x00020000);
    __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)a.out, 0, 0x80000000u, 0x00020000);

    struct Blk { int n, y0, x0; };                       // image, origin of the output block
    auto blk_of = [&](int j) { const int b = (int)blockIdx.x + j * (int)gridDim.x; Blk q; q.n = b / per_img; const int r = b - q.n * per_img; q.y0 = (r / bx) * B6_B; q.x0 = (r - (r / bx) * bx) * B6_B; return q; };
    // the 15 x 15 x 64 halo tile of block `q` -> lx: 1800 16-byte pieces (225 pixels x 8), lane-linear 1 KiB per instruction: piece id = 2 * row +
    // part covers tile pixels c = 8 * part + (lane >> 3) of that row (the 16th does not exist: those lanes stay out), chunk lane & 7 of each;
    // the piece at LDS slot `phys` of pixel p holds global chunk phys ^ (p & 7), (p & 7) = (c - row) & 7, 15 being -1 mod 8; pixels outside the
    // image are zero-filled by the range check.  (A rolled loop: see conv_block.hip.)
    auto fetch_x = [&](const Blk &q, b6_lds_char *dst) {
        const int cl = lane >> 3, chunk = lane & 7;
#pragma unroll 1
        for (int id = wave; id < B6_T * 2; id += B6_NW) {
            const int r = id >> 1, c = (id & 1) * 8 + cl;
            const int iy = q.y0 - 1 + r, ix = q.x0 - 1 + c;
            const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const unsigned off = ok ? (unsigned)((((q.n * a.H + iy) * a.W + ix) * a.x_stride + ((chunk ^ ((c - r) & 7)) * 8)) * 2) : 0x80000000u;
            if (c < B6_T) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (b6_lds_void *)(dst + (r * B6_T + (id & 1) * 8) * B6_XPITCH), 16, off, 0, 0, 0);
        }
    };

    // ---- per-lane constants of the 3x3's fragment reads: output pixel (j, l15) = raster index j * 16 + l15 of the 13 x 13 block ----
    typedef const __attribute__((address_space(3))) b6_bf16x8 *lds_frag_p;
    uint32_t mida[B6_NJ];
#pragma unroll
    for (int j = 0; j < B6_NJ; ++j) {
        int q = (wp * B6_NJ + j) * 16 + l15; if (q >= B6_OPIX) q = B6_OPIX - 1;     // (the last sub-tile's spare lanes repeat the last pixel: computed, not stored)
        const int oy = (q * 5042) >> 16, ox = q - oy * B6_B;         // q / 13 for q < 176
        mida[j] = (uint32_t)(uintptr_t)(b6_lds_char *)lmo_ + (uint32_t)((oy * B6_T + ox) * B6_MPITCH + lq * 16);
        asm volatile("" : "+v"(mida[j]));
    }
    // ---- INTERIOR blocks (the 15 x 15 halo window lies inside the image: 196 of the 256 blocks of a 208 x 208 image): every address is a
    //      per-lane constant plus one scalar per block, passed as the buffer instruction's soffset -- no per-block vector address arithmetic ----
    const int fcl = lane >> 3, fchunk = lane & 7;
    unsigned foff[4];                                   // this wave's halo pieces id = wave + 8 i: byte offset from the window's first pixel
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int id = wave + i * B6_NW, r = id >> 1, c = (id & 1) * 8 + fcl;
        foff[i] = (unsigned)(((r * a.W + c) * a.x_stride + ((fchunk ^ ((c - r) & 7)) * 8)) * 2);
    }
    unsigned soff[B6_NPIECE], xlds[B6_NPIECE];           // this thread's output pieces: byte offset from the block's first pixel; LDS address of the shortcut piece
    unsigned plds0;                                      // LDS address of its first staged piece (piece k: + k * 64 rows)
    {
#pragma unroll
        for (int k = 0; k < B6_NPIECE; ++k) {
            const int g = tid + k * 64 * B6_NW, px = g >> 3, piece = g & 7;
            const bool in = px < B6_OPIX;
            const int oy = in ? (px * 5042) >> 16 : 0, ox = in ? px - oy * B6_B : 0;
            const int hp = (oy + 1) * B6_T + ox + 1;
            soff[k] = in ? (unsigned)(((oy * a.W + ox) * a.out_stride + piece * 8) * 2) : 0x80000000u;
            xlds[k] = (unsigned)(uintptr_t)(b6_lds_char *)lx_ + (unsigned)(hp * B6_XPITCH + ((piece ^ (hp & 7)) << 4));
        }
        plds0 = (unsigned)(uintptr_t)(b6_lds_char *)lmo_ + (unsigned)((tid >> 3) * B6_OPITCH + (tid & 7) * 16);
    }
    auto interior_of = [&](const Blk &q) { return q.y0 >= 1 && q.x0 >= 1 && q.y0 + B6_B + 1 <= a.H && q.x0 + B6_B + 1 <= a.W; };
    auto fetch_x_interior = [&](const Blk &q, b6_lds_char *dst) {
        const unsigned base = (unsigned)(((q.n * a.H + q.y0 - 1) * a.W + q.x0 - 1) * a.x_stride * 2);        // scalar
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = wave + i * B6_NW;
            if (id < B6_T * 2 && (id & 1) * 8 + fcl < B6_T)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (b6_lds_void *)(dst + ((id >> 1) * B6_T + (id & 1) * 8) * B6_XPITCH), 16, foff[i], base, 0, 0);
        }
    };
    // One block.  Every LDS region is its own __restrict__ parameter (see conv_block.hip: hipcc cannot tell an LDS-DMA's target from any
    // other LDS access and would wait vmcnt(0) in front of the first LDS read after every fetch); the barriers order what the parameters hide.
    auto block = [&](int it, const b6_lds_char *__restrict__ lx, b6_lds_char *__restrict__ lx_dma, b6_lds_char *__restrict__ lmid, const b6_lds_char *__restrict__ mid_rd,
                     const b6_lds_char *__restrict__ lw1, const b6_lds_char *__restrict__ lb1, const b6_lds_char *__restrict__ lb2, b6_lds_char *__restrict__ lout,
                     const b6_lds_char *__restrict__ abs_rd) {
        const Blk q = blk_of(it);
        const bool interior = interior_of(q);             // (scalar)
        // this block's halo tile has landed (first pass: and the 1x1 filters are written): the vector-memory queue is in order, and behind the
        // tile's LDS-DMA this thread issued the previous block's B6_NPIECE stores (every lane issues every one: out-of-range offsets, never a
        // skipped instruction), which may still be in flight
        static_assert(B6_NPIECE == 3, "vmcnt below counts the stores of one thread");
        if (it > 0) __builtin_amdgcn_s_waitcnt(0x0073);  // vmcnt(3) lgkmcnt(0)  (the first tile was waited for in front of the loop)
        __builtin_amdgcn_s_barrier();
        // ================= stage 1: mid = act1(W1 . x + b1) on the 225 halo pixels, zero outside the image =================
        {
            const int u = wave;
            b6_f32x4 acc[2][2];
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) acc[s][ct] = b6_f32x4{0.f, 0.f, 0.f, 0.f};
            int p0 = (2 * u) * 16 + l15;
            asm volatile("" : "+v"(p0));
            const int p1 = p0 + 16;
            // (one K-step's fragments at a time: stage 1 is where the register peak would be)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                b6_bf16x8 xf[2], wf[2];
                xf[0] = *(lds_frag_p)(lx + p0 * B6_XPITCH + (((kk * 4 + lq) ^ (p0 & 7)) << 4));
                xf[1] = *(lds_frag_p)(lx + p1 * B6_XPITCH + (((kk * 4 + lq) ^ (p1 & 7)) << 4));
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) wf[ct] = *(lds_frag_p)(lw1 + (ct * 16 + l15) * B6_W1PITCH + (kk * 4 + lq) * 16);
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    acc[0][ct] = b6_mma<H16>(wf[ct], xf[0], acc[0][ct]);
                    acc[1][ct] = b6_mma<H16>(wf[ct], xf[1], acc[1][ct]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int p = s ? p1 : p0;
                if (2 * u + s >= B6_TSUB) continue;      // (wave-uniform)
                const int r = (p * 4370) >> 16, col = p - r * B6_T;
                const bool inside = p < B6_TPIX && (interior || ((unsigned)(q.y0 - 1 + r) < (unsigned)a.H && (unsigned)(q.x0 - 1 + col) < (unsigned)a.W));
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    uint2 pk = b6_epi<H16>(acc[s][ct], *(const __attribute__((address_space(3))) b6_f32x4 *)(lb1 + (ct * 16 + lq * 4) * 4), slope1);
                    if (!inside) pk = uint2{0u, 0u};
                    *(__attribute__((address_space(3))) uint2 *)(lmid + p * B6_MPITCH + (ct * 16 + lq * 4) * 2) = pk;
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();                    // the mid tile is complete
        // ================= stage 2: 3x3 over the mid tile, filters in registers: one MFMA per (tap, sub-tile) =================
        b6_f32x4 acc2[B6_NJ];
#pragma unroll
        for (int j = 0; j < B6_NJ; ++j) acc2[j] = b6_f32x4{0.f, 0.f, 0.f, 0.f};
        // step = (tap, half of the wave's six sub-tiles): three fragments, three MFMAs.  Registers are the constraint (two workgroups per CU
        // leave 128 per wave: 36 of filters, 24 accumulators, the per-lane address constants of the interior path): the fragments are not
        // double-buffered -- with four waves per SIMD the other waves' MFMAs cover a step's LDS latency
        auto koff = [](int t) { const int kh = t / 3, kw = t - kh * 3; return (kh * B6_T + kw) * B6_MPITCH; };
#pragma unroll
        for (int st = 0; st < 18; ++st) {
            const int t = st >> 1, g = st & 1;
            b6_bf16x8 fg[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) fg[j] = *(lds_frag_p)(mid_rd + mida[g * 3 + j] + koff(t));
#pragma unroll
            for (int j = 0; j < 3; ++j) acc2[g * 3 + j] = b6_mma<H16>(fw2[t], fg[j], acc2[g * 3 + j]);
        }
        __builtin_amdgcn_s_barrier();                    // every wave is done with the mid tile: the staged output goes over it
        // ================= epilogue =================
        {
            const b6_f32x4 bv = *(const __attribute__((address_space(3))) b6_f32x4 *)(lb2 + (wc * 16 + lq * 4) * 4);
#pragma unroll
            for (int j = 0; j < B6_NJ; ++j)
                if (wp * B6_NJ + j < B6_OSUB)                // (wave-uniform: the second half has one idle slot)
                    *(__attribute__((address_space(3))) uint2 *)(lout + ((wp * B6_NJ + j) * 16 + l15) * B6_OPITCH + (wc * 16 + lq * 4) * 2) = b6_epi<H16>(acc2[j], bv, slope2);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();                    // staged tile complete
        // 169 pixels x 8 pieces of 16 bytes; the shortcut is x at the same pixel and channels, still in the x tile
        // (piece g = tid + 512 k is 16-byte piece (g & 7) of block pixel (g >> 3): staged piece k sits 64 rows below piece 0)
        b6_u32x4 o[B6_NPIECE];
#pragma unroll
        for (int k = 0; k < B6_NPIECE; ++k) {
            const bool in = tid + k * 64 * B6_NW < B6_OPIX * 8;
            const b6_u32x4 v = *(const __attribute__((address_space(3))) b6_u32x4 *)(abs_rd + (in ? plds0 + (unsigned)(k * 64 * B6_OPITCH) : plds0));
            const b6_u32x4 r = *(const __attribute__((address_space(3))) b6_u32x4 *)(abs_rd + xlds[k]);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[k][e] = b6_pk<H16>(b6_lo<H16>(v[e]) + b6_lo<H16>(r[e]), b6_hi<H16>(v[e]) + b6_hi<H16>(r[e]));
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();                    // nobody reads the x tile (or the staged tile) any more
        if (it + 1 < nt) {                               // in flight under the stores below and the co-resident workgroup's block
            const Blk qn = blk_of(it + 1);
            if (interior_of(qn)) fetch_x_interior(qn, lx_dma); else fetch_x(qn, lx_dma);
        }
        if (interior) {
            const unsigned base = (unsigned)(((q.n * a.H + q.y0) * a.W + q.x0) * a.out_stride * 2);           // scalar
#pragma unroll
            for (int k = 0; k < B6_NPIECE; ++k) __builtin_amdgcn_raw_buffer_store_b128(o[k], ro, soff[k], base, OUT_STORE_AUX);
        } else {
#pragma unroll
            for (int k = 0; k < B6_NPIECE; ++k) {
                int g = tid + k * 64 * B6_NW;
                asm volatile("" : "+v"(g));
                const int px = g >> 3, piece = g & 7;
                const int oy = (px * 5042) >> 16, ox = px - oy * B6_B;
                const bool ok = px < B6_OPIX && q.y0 + oy < a.H && q.x0 + ox < a.W;
                const unsigned pix = (unsigned)((q.n * a.H + q.y0 + oy) * a.W + q.x0 + ox);
                __builtin_amdgcn_raw_buffer_store_b128(o[k], ro, ok ? (pix * a.out_stride + piece * 8) * 2 : 0x80000000u, 0, OUT_STORE_AUX);
            }
        }
    };
    if (nt > 0) { const Blk q0 = blk_of(0); if (interior_of(q0)) fetch_x_interior(q0, (b6_lds_char *)lx_); else fetch_x(q0, (b6_lds_char *)lx_); }
    __builtin_amdgcn_s_waitcnt(0x0070);                  // vmcnt(0) lgkmcnt(0): the first halo tile has landed, the 1x1 filters are written
    for (int it = 0; it < nt; ++it)
        block(it, (const b6_lds_char *)lx_, (b6_lds_char *)lx_, (b6_lds_char *)lmo_, (const b6_lds_char *)(uintptr_t)0, (const b6_lds_char *)lw1_, (const b6_lds_char *)lb1_,
              (const b6_lds_char *)lb2_, (b6_lds_char *)lmo_, (const b6_lds_char *)(uintptr_t)0);
#endif
}

bool conv_resblock64_ok(const BlockArgs &a)
{
    const double px = (double)a.N * a.H * a.W;
    if (px * a.x_stride * 2.0 >= 2147483648.0 || px * a.out_stride * 2.0 >= 2147483648.0) return false;      // 32-bit buffer offsets below the out-of-range sentinel
    // whole 13 x 13 blocks, or ragged ones on the bottom / right edge while they waste no more than 15 % (608 x 608: 304 = 24 * 13 - 8)
    const long cover = (long)((a.H + B6_B - 1) / B6_B) * ((a.W + B6_B - 1) / B6_B) * B6_B * B6_B;
    if (a.H <= 0 || a.W <= 0 || cover * 100 > (long)a.H * a.W * 115) return false;
    return (a.dt == DT_BF16 || a.dt == DT_F16) && a.C == B6_C && a.Cmid == B6_M &&
           a.Kpad1 >= B6_C && a.Kpad2 >= 9 * B6_M && (a.x_stride % 8) == 0 && a.x_stride >= B6_C && (a.out_stride % 8) == 0 && a.out_stride >= B6_C;
}

hipError_t launch_conv_resblock64(const BlockArgs &a, hipStream_t s)
{
    if (!conv_resblock64_ok(a)) return hipErrorInvalidValue;
    const bool h16 = a.dt == DT_F16;
    const void *k = h16 ? (const void *)conv_resblock_c64<true> : (const void *)conv_resblock_c64<false>;
    { hipError_t e = conv_opt_in_lds(k, B6_LDS); if (e != hipSuccess) return e; }
    long blocks = (long)a.N * ((a.H + B6_B - 1) / B6_B) * ((a.W + B6_B - 1) / B6_B);
    if (blocks > 512) blocks = 512;                      // persistent: two workgroups per CU
    if (h16) hipLaunchKernelGGL(conv_resblock_c64<true>, dim3((unsigned)blocks), dim3(64 * B6_NW), B6_LDS, s, a);
    else hipLaunchKernelGGL(conv_resblock_c64<false>, dim3((unsigned)blocks), dim3(64 * B6_NW), B6_LDS, s, a);
    return hipGetLastError();
}
