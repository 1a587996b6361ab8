// Fused residual block of darknet-53's 104 x 104 stage for gfx950 (V3/yolo_v3.py:26-38 `_darknet53_block`, DN cfg layers 6-8 and 9-11):
//     y = x + leaky(conv3x3(leaky(conv1x1(x))))          x, y: [N, H, W, 128],  1x1: 128 -> 64,  3x3: 64 -> 128
// Run layer by layer the block moves 352 MB at 416 x 416 x 32 (the 1x1 reads x and writes its 44 MB output, the 3x3 reads that back nine
// taps deep, reads x again as the shortcut and writes y) for 57 GFLOP: both launches sit on the memory system (29 + 63 us).  Here a
// workgroup owns one 13 x 13 block of output pixels at a time: the 15 x 15 x 128 halo tile of x comes into LDS by LDS-DMA (once, one block
// ahead), the 1x1 turns it into the 15 x 15 x 64 tile the 3x3 needs -- values rounded exactly as the unfused layer would have stored them,
// zeros where the 3x3 pads -- and the 3x3 contracts that tile with filters held in REGISTERS: a wave keeps the 32 output channels x 576 K
// of its filter slice (144 registers) for the life of the persistent workgroup.  The K loop then has no filter traffic at all; its only
// LDS traffic is the 6 pixel fragments per K-step that feed a wave's 12 MFMAs.  HBM sees x once (plus the 43 KB shortcut re-read of a
// block, an L2 hit) and y once: 176 MB, and the 1x1's tensor is never materialised.
//
//   waves    8 = 4 channel groups (32 output channels of the 3x3) x 2 pixel halves (sub-tiles 0-5 / 6-10 of the block's 11)
//   stage 1  wave w takes halo-pixel sub-tiles 2w, 2w+1: per K-step one x fragment and the four 16-channel filter fragments of the 1x1
//            (LDS), 4 MFMAs; bias, leaky, rounding, zero outside the image; ds_write_b64 into the mid tile.
//   stage 2  per tap and 32-channel half 6 ds_read_b128 at precomputed addresses (tap and half are immediate offsets) and 12 MFMAs,
//            reads issued one group of two fragments ahead.  Measured at the matrix-pipe floor (27.6 us of the launch's 78).
//   epilogue bias, leaky, rounding -> LDS -> 16-byte pieces; + the shortcut piece of x from global memory, rounded once more (as the
//            separate shortcut kernel would) -> global.
// LDS: the x tile has 256-byte pixel rows with the 16-byte slot XOR (pixel & 7) -- the LDS-DMA, which writes lane-linear 1 KiB pieces,
// gets the swizzle by permuting which global chunk each lane fetches; the mid tile and the staged output have padded rows (144 / 272 B):
// the 16 lanes of a fragment read fall on distinct bank slots and every tap is an immediate offset from one address per sub-tile.
// Registers are the constraint (256 per wave: 144 filters + 48 accumulators leave 64): a single spilled value is ruinous, because scratch
// accesses share vmcnt with the LDS-DMA and the stores -- every reload waits for whatever of those is in flight (measured: 78 -> 155 us
// with four spills).
// Same box, 416 x 416 x 32, bf16: 78 us against 26 + 59 for the two launches it replaces on a fast box, 89 against 91 on a slow one; the
// whole step gains 0.2 %: what the memory system no longer does, stage 1, the staging and the piece loop spend in issue slots (ablation:
// stage 1 9, DMA issue 3, piece loop 15, staging / barriers / waits 25 us).  HBM traffic of a forward falls by 350 MB (6 %).
#include "kernels.h"
#include <type_traits>

typedef __bf16 cb_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 cb_f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 cb_bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 cb_f16x2 __attribute__((ext_vector_type(2)));
typedef float cb_f32x2 __attribute__((ext_vector_type(2)));
typedef float cb_f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t cb_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void cb_lds_void;
typedef __attribute__((address_space(3))) char cb_lds_char;

template <bool H16> __device__ __forceinline__ uint32_t cb_pk(float lo, float hi)
{
    if constexpr (H16) return __builtin_bit_cast(uint32_t, __builtin_convertvector(cb_f32x2{lo, hi}, cb_f16x2));      // (MODE.FP16_OVFL: an overflowing conversion saturates at +-65504)
    else return __builtin_bit_cast(uint32_t, __builtin_convertvector(cb_f32x2{lo, hi}, cb_bf16x2));
}
template <bool H16> __device__ __forceinline__ float cb_lo(uint32_t w) { if constexpr (H16) return (float)__builtin_bit_cast(cb_f16x2, w)[0]; else return __builtin_bit_cast(float, w << 16); }
template <bool H16> __device__ __forceinline__ float cb_hi(uint32_t w) { if constexpr (H16) return (float)__builtin_bit_cast(cb_f16x2, w)[1]; else return __builtin_bit_cast(float, w & 0xffff0000u); }
__device__ __forceinline__ float cb_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// acc + bias, activation (slope 0.1: leaky as max(v, 0.1 v); slope 1: linear), rounded to the storage type: four channels as two packed words
template <bool H16> __device__ __forceinline__ uint2 cb_epi(const cb_f32x4 acc, const cb_f32x4 bias, const float slope)
{
    cb_f32x4 v = acc + bias;
    const cb_f32x4 t = v * slope;
    return uint2{cb_pk<H16>(cb_max(v[0], t[0]), cb_max(v[1], t[1])), cb_pk<H16>(cb_max(v[2], t[2]), cb_max(v[3], t[3]))};
}
template <bool H16> __device__ __forceinline__ cb_f32x4 cb_mma(const cb_bf16x8 a, const cb_bf16x8 b, const cb_f32x4 c)
{
    if constexpr (H16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(cb_f16x8, a), __builtin_bit_cast(cb_f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

constexpr int CB_B = 13;                               // output block edge
constexpr int CB_T = CB_B + 2;                         // halo tile edge: 15
constexpr int CB_TPIX = CB_T * CB_T;                   // 225
constexpr int CB_TSUB = (CB_TPIX + 15) / 16;           // 15 sub-tiles of 16 halo pixels (240 rows are allocated)
constexpr int CB_OPIX = CB_B * CB_B;                   // 169
constexpr int CB_OSUB = (CB_OPIX + 15) / 16;           // 11
constexpr int CB_C = 128, CB_M = 64;                   // block channels, mid channels
constexpr int CB_NW = 8;                               // waves per workgroup: 4 channel groups x 2 pixel halves
constexpr int CB_NJ = 6;                               // pixel sub-tiles of one wave in the 3x3 (half 0: 0..5, half 1: 6..10 and one idle slot)
constexpr int CB_X_BYTES = CB_TSUB * 16 * CB_C * 2;    // x halo tile, 256-byte pixel rows: 61440
constexpr int CB_MPITCH = CB_M * 2 + 16;               // mid tile pixel rows, padded: at 144 B the 16 lanes of a fragment read (pixel stride 1) fall on distinct 16-byte bank slots
constexpr int CB_MID_BYTES = CB_TSUB * 16 * CB_MPITCH; // 34560
constexpr int CB_W1PITCH = CB_C * 2 + 16;              // 1x1 filter rows in LDS, padded (272 B: 16 lanes of a fragment read on distinct slots)
constexpr int CB_W1_BYTES = CB_M * CB_W1PITCH;         // 17408
constexpr int CB_OPITCH = CB_C * 2 + 16;               // staged output rows
constexpr int CB_OUT_BYTES = CB_OSUB * 16 * CB_OPITCH; // 47872
constexpr int CB_B1_BYTES = CB_M * 4, CB_B2_BYTES = CB_C * 4;
constexpr int CB_LDS = CB_X_BYTES + CB_MID_BYTES + CB_W1_BYTES + CB_B1_BYTES + CB_B2_BYTES + CB_OUT_BYTES;      // 162048

template <bool H16>
__global__ __launch_bounds__(64 * CB_NW) void conv_resblock_c128(const BlockArgs a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if constexpr (H16) __builtin_amdgcn_s_setreg((0 << 11) | (23 << 6) | 1, 1);      // hwreg(HW_REG_MODE, 23, 1) = FP16_OVFL
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const int wc = wave & 3, wp = wave >> 2;             // channel group (32 output channels of the 3x3), pixel half
    char *const lx_ = smem, *const lmid_ = lx_ + CB_X_BYTES, *const lw1_ = lmid_ + CB_MID_BYTES, *const lb1_ = lw1_ + CB_W1_BYTES, *const lb2_ = lb1_ + CB_B1_BYTES, *const lout_ = lb2_ + CB_B2_BYTES;
    const float slope1 = a.act1 == ACT_LEAKY ? 0.1f : 1.f, slope2 = a.act2 == ACT_LEAKY ? 0.1f : 1.f;
    const int bx = (a.W + CB_B - 1) / CB_B, by = (a.H + CB_B - 1) / CB_B, per_img = bx * by, nblocks = a.N * per_img;      // (ragged blocks on the bottom / right edge: their pixels past the image are computed on zeros and never stored)
    const int nt = (nblocks - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;

    // ---- once per workgroup: 1x1 filters -> LDS; this wave's slice of the 3x3 filters and its biases -> registers ----
    for (int g = tid; g < CB_M * 16; g += 64 * CB_NW) {
        const int row = g >> 4, piece = g & 15;
        *(uint4 *)(lw1_ + row * CB_W1PITCH + piece * 16) = *(const uint4 *)((const bf16_t *)a.w1 + (size_t)row * a.Kpad1 + piece * 8);
    }
    cb_bf16x8 fw2[2][18];                                // [channel tile of this wave][tap * 2 + half]: K = tap * 64 + half * 32 + lq * 8 ..
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int ks = 0; ks < 18; ++ks)
            fw2[ct][ks] = *(const cb_bf16x8 *)((const bf16_t *)a.w2 + (size_t)(wc * 32 + ct * 16 + l15) * a.Kpad2 + ks * 32 + lq * 8);
    if (tid < CB_M) *(float *)(lb1_ + tid * 4) = a.b1[tid];
    if (tid < CB_C) *(float *)(lb2_ + tid * 4) = a.b2[tid];

    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, 0x80000000u, 0x00020000);
    __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)a.out, 0, 0x80000000u, 0x00020000);

    struct Blk { int n, y0, x0; };                       // image, origin of the output block
    auto blk_of = [&](int j) { const int b = (int)blockIdx.x + j * (int)gridDim.x; Blk q; q.n = b / per_img; const int r = b - q.n * per_img; q.y0 = (r / bx) * CB_B; q.x0 = (r - (r / bx) * bx) * CB_B; return q; };
    // the 15 x 15 x 128 halo tile of block `q` -> lx: 3600 16-byte pieces (225 pixels x 16), lane-linear 1 KiB per instruction; the piece at
    // LDS slot `phys` of pixel p holds global chunk phys ^ (p & 7); pixels outside the image are zero-filled by the range check
    auto fetch_x = [&](const Blk &q, cb_lds_char *dst) {
        // Row by row: a tile row is 15 pixels x 256 B, contiguous in global memory and in LDS; piece id = 4 * row + part covers tile pixels
        // c = 4 * part + (lane >> 4) of that row (the 16th does not exist: those lanes stay out), chunk lane & 15 of each.  Everything but the
        // column test and the swizzle key -- (pixel & 7) = (c - row) & 7, 15 being -1 mod 8 -- is scalar.
        // (a rolled loop: hipcc tracks the targets of only a handful of LDS-DMA instructions individually; with the pieces of an unrolled
        //  fetch it falls back to "an LDS-DMA may alias any LDS access" and waits vmcnt(0) in front of the next ds instruction)
        const int cl = lane >> 4, chunk = lane & 15;
#pragma unroll 1
        for (int id = wave; id < CB_T * 4; id += CB_NW) {
            const int r = id >> 2, c = (id & 3) * 4 + cl;
            const int iy = q.y0 - 1 + r, ix = q.x0 - 1 + c;
            const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const unsigned off = ok ? (unsigned)((((q.n * a.H + iy) * a.W + ix) * a.x_stride + ((chunk ^ ((c - r) & 7)) * 8)) * 2) : 0x80000000u;
            if (c < CB_T) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (cb_lds_void *)(dst + (r * CB_T + (id & 3) * 4) * 256), 16, off, 0, 0, 0);
        }
    };

    // ---- per-lane constants of the 3x3's fragment reads: output pixel (j, l15) = raster index j * 16 + l15 of the 13 x 13 block ----
    // address of mid pixel (oy + kh, ox + kw), 32-channel half h, this lane's 16 bytes: mida[j] + ((kh * 15 + kw) * CB_MPITCH + h * 64), the
    // second term an immediate
    typedef const __attribute__((address_space(3))) cb_bf16x8 *lds_frag_p;
    uint32_t mida[CB_NJ];
#pragma unroll
    for (int j = 0; j < CB_NJ; ++j) {
        int q = (wp * CB_NJ + j) * 16 + l15; if (q >= CB_OPIX) q = CB_OPIX - 1;     // (the last sub-tile's spare lanes repeat the last pixel: computed, not stored)
        const int oy = (q * 5042) >> 16, ox = q - oy * CB_B;         // q / 13 for q < 176
        mida[j] = (uint32_t)(uintptr_t)(cb_lds_char *)lmid_ + (uint32_t)((oy * CB_T + ox) * CB_MPITCH + lq * 16);
        asm volatile("" : "+v"(mida[j]));
    }

    // One block.  Every LDS region is its own __restrict__ parameter -- the x tile twice, as what stage 1 reads and as what the LDS-DMA of
    // the NEXT block fills; the mid tile twice, as what stage 1 writes and (null-based: `mida` holds absolute addresses) as what stage 2
    // reads -- or hipcc, which cannot tell an LDS-DMA's target from any other LDS access, waits vmcnt(0) in front of the first LDS read
    // after every fetch: the whole latency of the prefetch, exposed.  The barriers order what the parameters hide.
    auto block = [&](int it, const cb_lds_char *__restrict__ lx, cb_lds_char *__restrict__ lx_dma, cb_lds_char *__restrict__ lmid, const cb_lds_char *__restrict__ mid_rd,
                     const cb_lds_char *__restrict__ lw1, const cb_lds_char *__restrict__ lb1, const cb_lds_char *__restrict__ lb2, cb_lds_char *__restrict__ lout) {
        const Blk q = blk_of(it);
        // this block's halo tile has landed (first pass: and the 1x1 filters are written): the vector-memory queue is in order, and behind the
        // tile's LDS-DMA this thread issued the previous block's 6 shortcut loads (consumed since) and 6 stores, which may still be in flight
        static_assert((CB_OPIX * 16 + 64 * CB_NW - 1) / (64 * CB_NW) == 6, "vmcnt below counts the stores of one thread");
        if (it > 0) __builtin_amdgcn_s_waitcnt(0x0076);  // vmcnt(6) lgkmcnt(0)  (the first tile was waited for in front of the loop)
        __builtin_amdgcn_s_barrier();
        // ================= stage 1: mid = act1(W1 . x + b1) on the 225 halo pixels, zero outside the image =================
        // sub-tile pair (2u, 2u+1), u = wave: 8 pairs cover 16 sub-tiles (the 16th is padding rows: skipped)
        {
            const int u = wave;
            cb_f32x4 acc[2][4];
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) acc[s][ct] = cb_f32x4{0.f, 0.f, 0.f, 0.f};
            int p0 = (2 * u) * 16 + l15;
            asm volatile("" : "+v"(p0));
            const int p1 = p0 + 16;
            // the fragments of K-step kk + 1 (two of pixels, four of filters) are requested ahead of the MFMAs of K-step kk
            cb_bf16x8 xf[2][2], wf[2][4];
            auto frags = [&](int kk) {
                xf[kk & 1][0] = *(lds_frag_p)(lx + p0 * 256 + (((kk * 4 + lq) ^ (p0 & 7)) << 4));
                xf[kk & 1][1] = *(lds_frag_p)(lx + p1 * 256 + (((kk * 4 + lq) ^ (p1 & 7)) << 4));
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) wf[kk & 1][ct] = *(lds_frag_p)(lw1 + (ct * 16 + l15) * CB_W1PITCH + (kk * 4 + lq) * 16);
            };
            frags(0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                if (kk + 1 < 4) frags(kk + 1);
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    acc[0][ct] = cb_mma<H16>(wf[kk & 1][ct], xf[kk & 1][0], acc[0][ct]);
                    acc[1][ct] = cb_mma<H16>(wf[kk & 1][ct], xf[kk & 1][1], acc[1][ct]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int p = s ? p1 : p0;
                if (2 * u + s >= CB_TSUB) continue;      // (wave-uniform)
                const int r = (p * 4370) >> 16, col = p - r * CB_T;
                const bool inside = p < CB_TPIX && (unsigned)(q.y0 - 1 + r) < (unsigned)a.H && (unsigned)(q.x0 - 1 + col) < (unsigned)a.W;
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    uint2 pk = cb_epi<H16>(acc[s][ct], *(const __attribute__((address_space(3))) cb_f32x4 *)(lb1 + (ct * 16 + lq * 4) * 4), slope1);
                    if (!inside) pk = uint2{0u, 0u};
                    *(__attribute__((address_space(3))) uint2 *)(lmid + p * CB_MPITCH + (ct * 16 + lq * 4) * 2) = pk;
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();                    // the mid tile is complete; the x tile is free
        if (it + 1 < nt) fetch_x(blk_of(it + 1), lx_dma);        // lands during stage 2
        // ================= stage 2: 3x3 over the mid tile, filters in registers =================
        cb_f32x4 acc2[2][CB_NJ];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int j = 0; j < CB_NJ; ++j) acc2[ct][j] = cb_f32x4{0.f, 0.f, 0.f, 0.f};
        // K-step ks = tap * 2 + half.  The 6 pixel fragments of a K-step are read in three groups of 2, each one group ahead of the MFMAs that
        // consume it (the fences keep hipcc from re-ordering the software pipeline); the SIMD's other wave covers the rest of the LDS latency.
        // (Registers are the constraint: 144 of filters + 48 accumulators leave 64, and ONE spilled value is ruinous here -- scratch accesses
        //  share vmcnt with the LDS-DMA and the stores, so every reload waits for whatever of those is in flight.)
        cb_bf16x8 fg[3][2];
        auto koff = [](int ks) { const int t = ks >> 1, h = ks & 1, kh = t / 3, kw = t - kh * 3; return (kh * CB_T + kw) * CB_MPITCH + h * 64; };
#pragma unroll
        for (int j = 0; j < 2; ++j) fg[0][j] = *(lds_frag_p)(mid_rd + mida[j] + koff(0));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 18; ++ks)
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                const int gn = (g + 1) % 3, ksn = g == 2 ? ks + 1 : ks;      // the group read now, consumed by the next step
                if (ksn < 18) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) fg[gn][j] = *(lds_frag_p)(mid_rd + mida[gn * 2 + j] + koff(ksn));
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) { acc2[0][g * 2 + j] = cb_mma<H16>(fw2[0][ks], fg[g][j], acc2[0][g * 2 + j]); acc2[1][g * 2 + j] = cb_mma<H16>(fw2[1][ks], fg[g][j], acc2[1][g * 2 + j]); }
                __builtin_amdgcn_sched_barrier(0);
            }
        // ================= epilogue =================
        // bias, activation, rounding -> this wave's part of the staged tile, one channel tile at a time: the shortcut pieces of this thread
        // (x at the pixels and channels it stores below) are requested in between, as soon as half the accumulators are dead -- their
        // latency runs under the second half and the barrier
        auto stage_out = [&](int ct) {
            const cb_f32x4 bv = *(const __attribute__((address_space(3))) cb_f32x4 *)(lb2 + (wc * 32 + ct * 16 + lq * 4) * 4);
#pragma unroll
            for (int j = 0; j < CB_NJ; ++j)
                if (wp * CB_NJ + j < CB_OSUB)                // (wave-uniform: the second half has one idle slot)
                    *(__attribute__((address_space(3))) uint2 *)(lout + ((wp * CB_NJ + j) * 16 + l15) * CB_OPITCH + (wc * 32 + ct * 16 + lq * 4) * 2) = cb_epi<H16>(acc2[ct][j], bv, slope2);
        };
        stage_out(0);
        constexpr int NPIECE = (CB_OPIX * 16 + 64 * CB_NW - 1) / (64 * CB_NW);
        cb_u32x4 rsv[NPIECE];
#pragma unroll
        for (int k = 0; k < NPIECE; ++k) {
            int g = tid + k * 64 * CB_NW;
            asm volatile("" : "+v"(g));
            const int px = g >> 4, piece = g & 15;
            const int oy = (px * 5042) >> 16, ox = px - oy * CB_B;
            const unsigned pix = (unsigned)((q.n * a.H + q.y0 + oy) * a.W + q.x0 + ox);
            rsv[k] = __builtin_amdgcn_raw_buffer_load_b128(rx, (px < CB_OPIX && q.y0 + oy < a.H && q.x0 + ox < a.W) ? (pix * a.x_stride + piece * 8) * 2 : 0x80000000u, 0, 0);
        }
        stage_out(1);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();                    // staged tile complete (and every wave is done with the mid tile)
        // 169 pixels x 16 pieces of 16 bytes: piece g of thread tid + 256 k; the shortcut is x at the same pixel and channels
#pragma unroll
        for (int k = 0; k < NPIECE; ++k) {
            int g = tid + k * 64 * CB_NW;
            asm volatile("" : "+v"(g));
            const int px = g >> 4, piece = g & 15;
            const int oy = (px * 5042) >> 16, ox = px - oy * CB_B;
            const bool ok = px < CB_OPIX && q.y0 + oy < a.H && q.x0 + ox < a.W;
            const unsigned pix = (unsigned)((q.n * a.H + q.y0 + oy) * a.W + q.x0 + ox);
            const cb_u32x4 r = rsv[k];
            cb_u32x4 o = *(const __attribute__((address_space(3))) cb_u32x4 *)(lout + (px < CB_OPIX ? px : 0) * CB_OPITCH + piece * 16);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = cb_pk<H16>(cb_lo<H16>(o[e]) + cb_lo<H16>(r[e]), cb_hi<H16>(o[e]) + cb_hi<H16>(r[e]));
            __builtin_amdgcn_raw_buffer_store_b128(o, ro, ok ? (pix * a.out_stride + piece * 8) * 2 : 0x80000000u, 0, OUT_STORE_AUX);
        }
    };
    if (nt > 0) fetch_x(blk_of(0), (cb_lds_char *)lx_);
    __builtin_amdgcn_s_waitcnt(0x0070);                  // vmcnt(0) lgkmcnt(0): the first halo tile has landed, the 1x1 filters are written
    for (int it = 0; it < nt; ++it) block(it, (const cb_lds_char *)lx_, (cb_lds_char *)lx_, (cb_lds_char *)lmid_, (const cb_lds_char *)(uintptr_t)0, (const cb_lds_char *)lw1_, (const cb_lds_char *)lb1_, (const cb_lds_char *)lb2_, (cb_lds_char *)lout_);
#endif
}

bool conv_resblock_ok(const BlockArgs &a)
{
    const double px = (double)a.N * a.H * a.W;
    if (px * a.x_stride * 2.0 >= 2147483648.0 || px * a.out_stride * 2.0 >= 2147483648.0) return false;      // 32-bit buffer offsets below the out-of-range sentinel
    // whole 13 x 13 blocks, or ragged ones on the bottom / right edge while they waste no more than 15 % (608 x 608: 152 = 12 * 13 - 4)
    const long cover = (long)((a.H + CB_B - 1) / CB_B) * ((a.W + CB_B - 1) / CB_B) * CB_B * CB_B;
    if (a.H <= 0 || a.W <= 0 || cover * 100 > (long)a.H * a.W * 115) return false;
    return (a.dt == DT_BF16 || a.dt == DT_F16) && a.C == CB_C && a.Cmid == CB_M &&
           a.Kpad1 >= CB_C && a.Kpad2 >= 9 * CB_M && (a.x_stride % 8) == 0 && a.x_stride >= CB_C && (a.out_stride % 8) == 0 && a.out_stride >= CB_C;
}

hipError_t launch_conv_resblock(const BlockArgs &a, hipStream_t s)
{
    if (!conv_resblock_ok(a)) return hipErrorInvalidValue;
    const bool h16 = a.dt == DT_F16;
    const void *k = h16 ? (const void *)conv_resblock_c128<true> : (const void *)conv_resblock_c128<false>;
    { hipError_t e = conv_opt_in_lds(k, CB_LDS); if (e != hipSuccess) return e; }
    long blocks = (long)a.N * ((a.H + CB_B - 1) / CB_B) * ((a.W + CB_B - 1) / CB_B);
    if (blocks > 256) blocks = 256;                      // persistent: one workgroup per CU
    if (h16) hipLaunchKernelGGL(conv_resblock_c128<true>, dim3((unsigned)blocks), dim3(64 * CB_NW), CB_LDS, s, a);
    else hipLaunchKernelGGL(conv_resblock_c128<false>, dim3((unsigned)blocks), dim3(64 * CB_NW), CB_LDS, s, a);
    return hipGetLastError();
}
