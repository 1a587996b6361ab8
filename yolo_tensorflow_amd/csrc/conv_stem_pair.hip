// Fused first two convs of a darknet in the split-fp16 configuration (YOLO_FP16X2; round 6): conv0 (3x3 / stride 1, image -> C0 <= 32
// channels) and conv1 (3x3 / stride 2, 32 -> 64 channels) in ONE launch, conv0's tensor never materialised.  As separate launches they are
// the two largest memory movers of the network -- conv0 writes 709 MB of pairs at 416 x 416 x 32 that conv1 reads straight back (270 +
// 264 us, both at their traffic floors) -- exactly what the bf16 configuration's conv_stem_c32_c64 removes for 16-bit storage
// (conv_stem.hip; the reference's darknet-53 layers 0-1, V3/yolo_v3.py:22-24, DN cfg layers 0-1).
//
// One persistent 8-wave workgroup per CU walks 8 x 16 tiles of conv1 pixels.  Per tile:
//   A  the 19 x 35 image pixels the tile needs (three 16-byte pieces each: hi | lo | hi of the 8 padded channels, PAIR_B3) come into LDS --
//      fetched into registers during the previous tile's phase C, written after it; out-of-image pixels are zeros (conv0's padding);
//   B  conv0 on the 17 x 33 pixels conv1 reads: 36 groups of 16 pixels over the 8 waves, per group the direct kernel's arithmetic
//      (conv_c8_3x3_direct_pair: per K-group W_hi x_hi, W_lo x_hi, W_hi x_lo; bias, leaky, split) -> an LDS tile of interleaved pairs
//      (64 B hi | 64 B lo per pixel, pitch 144 B); pixels outside the image are conv1's zero padding: zeros;
//   C  conv1 from that tile: wave (cg, pq) owns 32 output channels (filters W1_hi / W1_lo register-resident, 144 VGPRs) x 2 output rows;
//      per tap and sub-tile the pair K loop's three products in its order (A, B, C per accumulator and K-step, taps ascending);
//   E  bias, leaky, split; each wave's 16 pixels x (32 hi | 32 lo) leave through a private LDS slab as whole 128-byte rows (sc1).
// Every intermediate keeps the rounding points and the K order of the separate launches (the direct first-layer kernel and the tiled pair
// kernel): the fused plan is bit-identical to the layer-by-layer plan (tests/test_gpu_fp16x2.py).
#include "conv_igemm_kernel.h"
#include <cstdlib>

namespace {

constexpr int SP_TH = 8, SP_TW = 16;                       // conv1 pixels per tile
constexpr int SP_IH = 2 * SP_TH + 3, SP_IW = 2 * SP_TW + 3; // 19 x 35 image pixels
constexpr int SP_MH = 2 * SP_TH + 1, SP_MW = 2 * SP_TW + 1; // 17 x 33 conv0 pixels
constexpr int SP_IPIX = 48;                                // bytes of an image pixel record (hi 8 | lo 8 | hi 8)
constexpr int SP_MPIX = 144;                               // pitch of a conv0 pixel in LDS (64 B hi | 64 B lo | 16 B pad)
constexpr int SP_IBYTES = SP_IH * SP_IW * SP_IPIX;          // 31 920
constexpr int SP_MBYTES = SP_MH * SP_MW * SP_MPIX;          // 80 784
constexpr int SP_SLAB = 16 * 144;                          // one wave's output slab
constexpr int SP_NPIECE = SP_IH * SP_IW * 3;               // 16-byte pieces of an image tile: 1995
constexpr int SP_PPT = (SP_NPIECE + 511) / 512;            // per thread: 4
constexpr int SP_W0 = 12 * 1024;                            // conv0's filter fragments (W_hi / W_lo x 2 channel tiles x 3 K-groups), one KiB each, in LDS
constexpr int SP_LDS = ((SP_IBYTES + 15) / 16 * 16) + SP_MBYTES + 8 * SP_SLAB + SP_W0;

}  // namespace

template <bool U8>
__global__ __launch_bounds__(512) void conv_stem_pair_c32_c64(const StemPairArgs a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    fp16_saturating_mode();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *const I0 = smem;                                          // image tile
    char *const M0 = smem + (SP_IBYTES + 15) / 16 * 16;             // conv0 tile (pairs)
    char *const slab = M0 + SP_MBYTES + (threadIdx.x >> 6) * SP_SLAB;
    char *const W0S = M0 + SP_MBYTES + 8 * SP_SLAB;                  // fragment f = (i * 3 + kk) * 2 + (0 hi | 1 lo): lane's 16 bytes at f * 1024 + lane * 16
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cg = wave & 1, pq = wave >> 1;                        // phase C: channel half (32 channels), output rows 2 pq, 2 pq + 1
    const int tx_n = (a.Wo + SP_TW - 1) / SP_TW, ty_n = (a.Ho + SP_TH - 1) / SP_TH;
    const int tiles = a.N * ty_n * tx_n;
    const bf16_t *__restrict__ in = (const bf16_t *)a.in;

    // conv1 filters of this wave's 32 channels, register-resident: A fragment (i, tap): row cg * 32 + i * 16 + l15, elements tap * 64 + lq * 8 (hi), + 32 (lo)
    bf16x8 w1h[2][9], w1l[2][9];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const bf16_t *row = (const bf16_t *)a.w1 + (size_t)(cg * 32 + i * 16 + l15) * a.Kpad1 + t * 64 + lq * 8;
            w1h[i][t] = *(const bf16x8 *)row; w1l[i][t] = *(const bf16x8 *)(row + 32);
        }
    const float slope0 = a.act0 == ACT_LEAKY ? 0.1f : 1.0f, slope1 = a.act1 == ACT_LEAKY ? 0.1f : 1.0f;

    auto tile_origin = [&](int tile, int &n, int &oy0, int &ox0) {
        n = tile / (ty_n * tx_n); const int r = tile - n * ty_n * tx_n; const int ty = r / tx_n;
        oy0 = ty * SP_TH; ox0 = (r - ty * tx_n) * SP_TW;
    };
    // phase A, first half: this thread's share of tile `tile`'s image window into registers (zeros outside the image).  Two forms: 16-byte
    // pieces of the staged pair image (hi | lo | hi of the 8 padded channels), or -- U8 -- the caller's uint8 [N,H,W,3] batch itself, two pixels
    // per thread, converted when they are written to LDS with the arithmetic of the two launches this retires (k_preprocess: x * scale
    // [* mul + add] in fp32; k_split_from_f32: hi = f16(v), lo = f16(v - hi)): 145 us per batch-32 step at 416 x 416
    uint4 pre[SP_PPT];
    constexpr int UPT = (SP_IH * SP_IW + 511) / 512;       // U8: pixels per thread (2)
    unsigned pre8[UPT];                                    // U8: r | g << 8 | b << 16 | valid << 24
    auto fetch = [&](int tile) {
        int n, oy0, ox0; tile_origin(tile, n, oy0, ox0);
        if constexpr (U8) {
#pragma unroll
            for (int k = 0; k < UPT; ++k) {
                const int px = tid + k * 512, r = px / SP_IW, c = px - r * SP_IW;
                const int y = 2 * oy0 - 2 + r, x = 2 * ox0 - 2 + c;
                const bool ok = px < SP_IH * SP_IW && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
                unsigned v = 0;
                if (ok) { const uint8_t *p = a.in_u8 + ((size_t)(n * a.H + y) * a.W + x) * 3; v = (unsigned)p[0] | ((unsigned)p[1] << 8) | ((unsigned)p[2] << 16) | (1u << 24); }
                pre8[k] = v;
            }
        } else {
#pragma unroll
            for (int k = 0; k < SP_PPT; ++k) {
                const int pc = tid + k * 512;
                const int px = pc / 3, part = pc - px * 3, r = px / SP_IW, c = px - r * SP_IW;
                const int y = 2 * oy0 - 2 + r, x = 2 * ox0 - 2 + c;
                const bool ok = pc < SP_NPIECE && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
                pre[k] = ok ? *(const uint4 *)(in + ((size_t)(n * a.H + y) * a.W + x) * 24 + part * 8) : uint4{0u, 0u, 0u, 0u};
            }
        }
    };
    auto stash = [&]() {
        if constexpr (U8) {
            const bool affine = a.in_mul != 1.0f || a.in_add != 0.0f;
#pragma unroll
            for (int k = 0; k < UPT; ++k) {
                const int px = tid + k * 512;
                if (px >= SP_IH * SP_IW) continue;
                uint4 H = uint4{0u, 0u, 0u, 0u}, L = H;
                if (pre8[k] >> 24) {
                    float v[3];
#pragma unroll
                    for (int e = 0; e < 3; ++e) { v[e] = (float)((pre8[k] >> (8 * e)) & 0xffu) * a.in_scale; if (affine) v[e] = v[e] * a.in_mul + a.in_add; }
                    H.x = pack16x2<true>(v[0], v[1]); H.y = pack16x2<true>(v[2], 0.f);
                    L.x = pack16x2<true>(v[0] - unpack16_lo<true>(H.x), v[1] - unpack16_hi<true>(H.x)); L.y = pack16x2<true>(v[2] - unpack16_lo<true>(H.y), 0.f);
                }
                *(uint4 *)(I0 + px * SP_IPIX) = H; *(uint4 *)(I0 + px * SP_IPIX + 16) = L; *(uint4 *)(I0 + px * SP_IPIX + 32) = H;
            }
        } else {
#pragma unroll
            for (int k = 0; k < SP_PPT; ++k) { const int pc = tid + k * 512; if (pc < SP_NPIECE) *(uint4 *)(I0 + pc * 16) = pre[k]; }
        }
    };

    // conv0's filters (32 x 216, pairs): too many registers beside conv1's 144 -- kept in LDS in fragment order, re-read per 16-pixel group
    for (int f = wave; f < 12; f += 8) {
        const int i = f / 6, kk = (f >> 1) % 3, lo = f & 1, tap = kk * 4 + lq;
        const bf16_t *row = (const bf16_t *)a.w0 + (size_t)(i * 16 + l15) * a.Kpad0 + tap * 24 + lo * 16;
        *(uint4 *)(W0S + f * 1024 + lane * 16) = tap < 9 ? *(const uint4 *)row : uint4{0u, 0u, 0u, 0u};
    }
    int tile = blockIdx.x;
    if (tile < tiles) { fetch(tile); stash(); }
    __syncthreads();
    for (; tile < tiles; tile += gridDim.x) {
        int n, oy0, ox0; tile_origin(tile, n, oy0, ox0);
        // ---- phase B: conv0 on the 17 x 33 pixels, 16 at a time ----
        {
            f32x4 b0v[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) b0v[i] = *(const f32x4 *)(a.b0 + i * 16 + lq * 4);
            constexpr int NG = (SP_MH * SP_MW + 15) / 16;          // 36 groups
            // (Tried and dropped: K-group outermost with the wave's five groups' accumulators live, so that a filter fragment is read from
            // LDS once per tile instead of once per group -- 58 spilled registers beside conv1's 144, 482 -> 760 us.)
            for (int g = wave; g < NG; g += 8) {
                const int q = g * 16 + l15;
                const bool qv = q < SP_MH * SP_MW;
                const int qq = qv ? q : 0, my = qq / SP_MW, mx = qq - my * SP_MW;
                bf16x8 xh[3], xl[3];
#pragma unroll
                for (int kk = 0; kk < 3; ++kk) {
                    const int tap = kk * 4 + lq;
                    const int kh = (tap * 11) >> 5, kw = tap - kh * 3;
                    if (tap < 9) {
                        const char *p = I0 + ((my + kh) * SP_IW + mx + kw) * SP_IPIX;
                        xh[kk] = *(const bf16x8 *)p; xl[kk] = *(const bf16x8 *)(p + 16);
                    } else { xh[kk] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; xl[kk] = xh[kk]; }
                }
                f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int kk = 0; kk < 3; ++kk) {
                    bf16x8 wh[2], wl[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) { wh[i] = *(const bf16x8 *)(W0S + ((i * 3 + kk) * 2) * 1024 + lane * 16); wl[i] = *(const bf16x8 *)(W0S + ((i * 3 + kk) * 2 + 1) * 1024 + lane * 16); }
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc[i] = mma16<true>(wh[i], xh[kk], acc[i]);
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc[i] = mma16<true>(wl[i], xh[kk], acc[i]);
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc[i] = mma16<true>(wh[i], xl[kk], acc[i]);
                }
                // conv0 pixel (y, x) of the image; outside it: conv1's zero padding
                const int y = 2 * oy0 - 1 + my, x = 2 * ox0 - 1 + mx;
                const bool inside = (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
                if (qv) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        float v[4] = {acc[i][0] + b0v[i][0], acc[i][1] + b0v[i][1], acc[i][2] + b0v[i][2], acc[i][3] + b0v[i][3]};
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v[e] = vmax_f32(v[e], v[e] * slope0); v[e] = inside && (i * 16 + lq * 4 + e) < a.C0 ? v[e] : 0.f; }
                        uint2 H, L;
                        H.x = pack16x2<true>(v[0], v[1]); H.y = pack16x2<true>(v[2], v[3]);
                        L.x = pack16x2<true>(v[0] - unpack16_lo<true>(H.x), v[1] - unpack16_hi<true>(H.x));
                        L.y = pack16x2<true>(v[2] - unpack16_lo<true>(H.y), v[3] - unpack16_hi<true>(H.y));
                        *(uint2 *)(M0 + q * SP_MPIX + i * 32 + lq * 8) = H;
                        *(uint2 *)(M0 + q * SP_MPIX + 64 + i * 32 + lq * 8) = L;
                    }
                }
            }
        }
        __syncthreads();                                   // conv0 tile complete; the image tile is no longer read
        const int next = tile + gridDim.x;
        if (next < tiles) fetch(next);                     // phase A of the next tile rides under phase C
        // ---- phase C: conv1, this wave's 32 channels x 2 output rows ----
        f32x4 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int kh = t / 3, kw = t - kh * 3;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int oy = 2 * pq + j;
                const char *p = M0 + ((2 * oy + kh) * SP_MW + 2 * l15 + kw) * SP_MPIX + lq * 16;
                const bf16x8 xh = *(const bf16x8 *)p, xl = *(const bf16x8 *)(p + 64);
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[i][j] = mma16<true>(w1h[i][t], xh, acc[i][j]);
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[i][j] = mma16<true>(w1l[i][t], xh, acc[i][j]);
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[i][j] = mma16<true>(w1h[i][t], xl, acc[i][j]);
            }
        }
        // ---- epilogue: bias, leaky, split; 16 pixels x (32 hi | 32 lo) per sub-tile through the wave's slab as whole 128-byte rows ----
        f32x4 b1v[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) b1v[i] = *(const f32x4 *)(a.b1 + cg * 32 + i * 16 + lq * 4);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int oy = oy0 + 2 * pq + j;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float v[4] = {acc[i][j][0] + b1v[i][0], acc[i][j][1] + b1v[i][1], acc[i][j][2] + b1v[i][2], acc[i][j][3] + b1v[i][3]};
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = vmax_f32(v[e], v[e] * slope1);
                uint2 H, L;
                H.x = pack16x2<true>(v[0], v[1]); H.y = pack16x2<true>(v[2], v[3]);
                L.x = pack16x2<true>(v[0] - unpack16_lo<true>(H.x), v[1] - unpack16_hi<true>(H.x));
                L.y = pack16x2<true>(v[2] - unpack16_lo<true>(H.y), v[3] - unpack16_hi<true>(H.y));
                *(uint2 *)(slab + l15 * 144 + i * 32 + lq * 8) = H;
                *(uint2 *)(slab + l15 * 144 + 64 + i * 32 + lq * 8) = L;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (oy < a.Ho) {
                const char *obase = (const char *)a.out + ((size_t)(n * a.Ho + oy) * a.Wo + ox0) * a.out_stride * 2 + cg * 128;      // wave-uniform
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int pc = lane + 64 * k, px = pc >> 3, c16 = pc & 7;
                    const uint4 o = *(const uint4 *)(slab + px * 144 + c16 * 16);
                    if (ox0 + px < a.Wo) out_store16_at(obase, (unsigned)(px * a.out_stride * 2 + c16 * 16), o.x, o.y, o.z, o.w);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the slab is rewritten by the next sub-tile
        }
        if (next < tiles) stash();                         // (the image tile was free since the barrier above)
        __syncthreads();                                   // next image tile in LDS; the conv0 tile is free again
    }
#endif
}

// ---- producer / consumer form (round 6, second step) -----------------------------------------------------------------------------
// In the kernel above the phases of a tile run one after the other on all eight waves (probe builds: conv0 135 us, conv1 70 us at its MFMA
// floor, fetch / convert / stores / barriers 111 us, and they ADD UP to the launch's 310 us).  Here the roles are split as in the bf16 stem
// (conv_stem.hip): waves 0-3 PRODUCE -- fetch the image window of tile k + 2, run conv0 of tile k + 1 (filters W0_hi / W0_lo in THEIR
// registers: no re-reads from LDS) into one half of a double-buffered conv0 tile, convert and stash window k + 2 -- while waves 4-7 CONSUME:
// conv1 of tile k from the other half with their register-resident W1, epilogue, stores.  Every SIMD hosts one wave of each kind, ONE
// barrier per tile moves the pipeline on.  Tiles are 4 x 16 conv1 pixels so that two conv0 tiles (9 x 33 pixels x 144 B) and two image
// windows (11 x 35 x 48 B) fit LDS: 132 KB.  Same arithmetic, same order per accumulator: bit-identical to the kernel above and to the
// separate launches.
namespace {
constexpr int PC_TH = 4, PC_TW = 16;
constexpr int PC_IH = 2 * PC_TH + 3, PC_IW = 2 * PC_TW + 3;   // 11 x 35
constexpr int PC_MH = 2 * PC_TH + 1, PC_MW = 2 * PC_TW + 1;   // 9 x 33
constexpr int PC_IBYTES = PC_IH * PC_IW * SP_IPIX;            // 18 480
constexpr int PC_MBYTES = PC_MH * PC_MW * SP_MPIX;            // 42 768
constexpr int PC_LDS = 2 * PC_IBYTES + 2 * PC_MBYTES + 4 * SP_SLAB;
constexpr int PC_IPT = (PC_IH * PC_IW + 255) / 256;           // image pixels per producer thread: 2
constexpr int PC_NPIECE = PC_IH * PC_IW * 3, PC_PPT = (PC_NPIECE + 255) / 256;      // staged form: 16-byte pieces per producer thread: 5
}  // namespace

template <bool U8>
__global__ __launch_bounds__(512) void conv_stem_pair_pc(const StemPairArgs a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    fp16_saturating_mode();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *const I0 = smem;                                          // two image windows
    char *const M0 = smem + 2 * PC_IBYTES;                          // two conv0 tiles (pairs)
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave < 4;
    const int tx_n = (a.Wo + PC_TW - 1) / PC_TW, ty_n = (a.Ho + PC_TH - 1) / PC_TH;
    const int tiles = a.N * ty_n * tx_n;
    const int G = gridDim.x;
    auto tile_origin = [&](int tile, int &n, int &oy0, int &ox0) {
        n = tile / (ty_n * tx_n); const int r = tile - n * ty_n * tx_n; const int ty = r / tx_n;
        oy0 = ty * PC_TH; ox0 = (r - ty * tx_n) * PC_TW;
    };
    const float slope0 = a.act0 == ACT_LEAKY ? 0.1f : 1.0f, slope1 = a.act1 == ACT_LEAKY ? 0.1f : 1.0f;
    const int first = blockIdx.x;
    const int count = first < tiles ? (tiles - first + G - 1) / G : 0;       // this workgroup's tiles: first, first + G, ...

    if (producer) {
        const int ptid = tid;                                        // 0 .. 255
        const bf16_t *__restrict__ in = (const bf16_t *)a.in;
        // conv0 filters in registers: A fragment (i, kk): row i * 16 + l15, tap kk * 4 + lq
        bf16x8 w0h[2][3], w0l[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int kk = 0; kk < 3; ++kk) {
                const int tap = kk * 4 + lq;
                const bf16_t *row = (const bf16_t *)a.w0 + (size_t)(i * 16 + l15) * a.Kpad0 + tap * 24;
                if (tap < 9) { w0h[i][kk] = *(const bf16x8 *)row; w0l[i][kk] = *(const bf16x8 *)(row + 16); }
                else { w0h[i][kk] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; w0l[i][kk] = w0h[i][kk]; }
            }
        f32x4 b0v[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) b0v[i] = *(const f32x4 *)(a.b0 + i * 16 + lq * 4);
        unsigned pre8[PC_IPT]; uint4 pre[U8 ? 1 : PC_PPT];
        auto fetch = [&](int tile) {
            int n, oy0, ox0; tile_origin(tile, n, oy0, ox0);
            if constexpr (U8) {
#pragma unroll
                for (int k = 0; k < PC_IPT; ++k) {
                    const int px = ptid + k * 256, r = px / PC_IW, c = px - r * PC_IW;
                    const int y = 2 * oy0 - 2 + r, x = 2 * ox0 - 2 + c;
                    const bool ok = px < PC_IH * PC_IW && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
                    unsigned v = 0;
                    if (ok) { const uint8_t *p = a.in_u8 + ((size_t)(n * a.H + y) * a.W + x) * 3; v = (unsigned)p[0] | ((unsigned)p[1] << 8) | ((unsigned)p[2] << 16) | (1u << 24); }
                    pre8[k] = v;
                }
            } else {
#pragma unroll
                for (int k = 0; k < PC_PPT; ++k) {
                    const int pc = ptid + k * 256;
                    const int px = pc / 3, part = pc - px * 3, r = px / PC_IW, c = px - r * PC_IW;
                    const int y = 2 * oy0 - 2 + r, x = 2 * ox0 - 2 + c;
                    const bool ok = pc < PC_NPIECE && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
                    pre[k] = ok ? *(const uint4 *)(in + ((size_t)(n * a.H + y) * a.W + x) * 24 + part * 8) : uint4{0u, 0u, 0u, 0u};
                }
            }
        };
        auto stash = [&](char *dst) {
            if constexpr (U8) {
                const bool affine = a.in_mul != 1.0f || a.in_add != 0.0f;
#pragma unroll
                for (int k = 0; k < PC_IPT; ++k) {
                    const int px = ptid + k * 256;
                    if (px >= PC_IH * PC_IW) continue;
                    uint4 H = uint4{0u, 0u, 0u, 0u}, L = H;
                    if (pre8[k] >> 24) {
                        float v[3];
#pragma unroll
                        for (int e = 0; e < 3; ++e) { v[e] = (float)((pre8[k] >> (8 * e)) & 0xffu) * a.in_scale; if (affine) v[e] = v[e] * a.in_mul + a.in_add; }
                        H.x = pack16x2<true>(v[0], v[1]); H.y = pack16x2<true>(v[2], 0.f);
                        L.x = pack16x2<true>(v[0] - unpack16_lo<true>(H.x), v[1] - unpack16_hi<true>(H.x)); L.y = pack16x2<true>(v[2] - unpack16_lo<true>(H.y), 0.f);
                    }
                    *(uint4 *)(dst + px * SP_IPIX) = H; *(uint4 *)(dst + px * SP_IPIX + 16) = L; *(uint4 *)(dst + px * SP_IPIX + 32) = H;
                }
            } else {
#pragma unroll
                for (int k = 0; k < PC_PPT; ++k) { const int pc = ptid + k * 256; if (pc < PC_NPIECE) *(uint4 *)(dst + pc * 16) = pre[k]; }
            }
        };
        // conv0 of the tile whose window is in `img` into `mid`: 19 groups of 16 pixels over the four producer waves
        auto conv0 = [&](int tile, const char *__restrict__ img, char *__restrict__ mid) {
            int n, oy0, ox0; tile_origin(tile, n, oy0, ox0);
            constexpr int NG = (PC_MH * PC_MW + 15) / 16, GW = (NG + 3) / 4;     // 19 groups, up to 5 per producer wave
            // Two passes over the wave's groups instead of group by group: a lone producer wave per SIMD has nobody to hide its latencies
            // behind, so the LDS reads and the (independent) MFMA chains of all its groups are issued together, then the splits.
            f32x4 acc[GW][2];
#pragma unroll
            for (int u = 0; u < GW; ++u) {
                acc[u][0] = acc[u][1] = f32x4{0.f, 0.f, 0.f, 0.f};
                const int g = wave + 4 * u;
                if (g >= NG) continue;
                const int q = g * 16 + l15;
                const int qq = q < PC_MH * PC_MW ? q : 0, my = qq / PC_MW, mx = qq - my * PC_MW;
                bf16x8 xh[3], xl[3];
#pragma unroll
                for (int kk = 0; kk < 3; ++kk) {
                    const int tap = kk * 4 + lq;
                    const int kh = (tap * 11) >> 5, kw = tap - kh * 3;
                    if (tap < 9) {
                        const char *p = img + ((my + kh) * PC_IW + mx + kw) * SP_IPIX;
                        xh[kk] = *(const bf16x8 *)p; xl[kk] = *(const bf16x8 *)(p + 16);
                    } else { xh[kk] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; xl[kk] = xh[kk]; }
                }
#pragma unroll
                for (int kk = 0; kk < 3; ++kk) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc[u][i] = mma16<true>(w0h[i][kk], xh[kk], acc[u][i]);
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc[u][i] = mma16<true>(w0l[i][kk], xh[kk], acc[u][i]);
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc[u][i] = mma16<true>(w0h[i][kk], xl[kk], acc[u][i]);
                }
            }
#pragma unroll
            for (int u = 0; u < GW; ++u) {
                const int g = wave + 4 * u, q = g * 16 + l15;
                if (g >= NG || q >= PC_MH * PC_MW) continue;
                const int my = q / PC_MW, mx = q - my * PC_MW;
                const int y = 2 * oy0 - 1 + my, x = 2 * ox0 - 1 + mx;
                const bool inside = (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    float v[4] = {acc[u][i][0] + b0v[i][0], acc[u][i][1] + b0v[i][1], acc[u][i][2] + b0v[i][2], acc[u][i][3] + b0v[i][3]};
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] = vmax_f32(v[e], v[e] * slope0); v[e] = inside && (i * 16 + lq * 4 + e) < a.C0 ? v[e] : 0.f; }
                    uint2 H, L;
                    H.x = pack16x2<true>(v[0], v[1]); H.y = pack16x2<true>(v[2], v[3]);
                    L.x = pack16x2<true>(v[0] - unpack16_lo<true>(H.x), v[1] - unpack16_hi<true>(H.x));
                    L.y = pack16x2<true>(v[2] - unpack16_lo<true>(H.y), v[3] - unpack16_hi<true>(H.y));
                    *(uint2 *)(mid + q * SP_MPIX + i * 32 + lq * 8) = H;
                    *(uint2 *)(mid + q * SP_MPIX + 64 + i * 32 + lq * 8) = L;
                }
            }
        };
        // prologue: window 0 -> I0[0], conv0(0) -> M0[0], window 1 -> I0[1]
        if (count > 0) { fetch(first); stash(I0); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();                                               // (a) window 0 visible to all producers
        if (count > 1) fetch(first + G);
        if (count > 0) conv0(first, I0, M0);
        if (count > 1) stash(I0 + PC_IBYTES);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();                                               // (b) conv0 tile 0 and window 1 in LDS
        for (int k = 0; k < count; ++k) {
            // iteration k: consumers run conv1 of tile k from M0[k & 1]; producers make tile k + 1 and bring window k + 2 in
            if (k + 2 < count) fetch(first + (k + 2) * G);
            if (k + 1 < count) conv0(first + (k + 1) * G, I0 + ((k + 1) & 1) * PC_IBYTES, M0 + ((k + 1) & 1) * PC_MBYTES);
            if (k + 2 < count) stash(I0 + (k & 1) * PC_IBYTES);      // window k's buffer: conv0(k) read it in the previous iteration
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __syncthreads();
        }
    } else {
        const int cw = wave - 4, cg = cw & 1, rp = cw >> 1;           // 32 output channels (cg) x output rows 2 rp, 2 rp + 1
        char *const slab = M0 + 2 * PC_MBYTES + cw * SP_SLAB;
        bf16x8 w1h[2][9], w1l[2][9];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const bf16_t *row = (const bf16_t *)a.w1 + (size_t)(cg * 32 + i * 16 + l15) * a.Kpad1 + t * 64 + lq * 8;
                w1h[i][t] = *(const bf16x8 *)row; w1l[i][t] = *(const bf16x8 *)(row + 32);
            }
        f32x4 b1v[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) b1v[i] = *(const f32x4 *)(a.b1 + cg * 32 + i * 16 + lq * 4);
        __syncthreads();                                               // (a)
        __syncthreads();                                               // (b)
        for (int k = 0; k < count; ++k) {
            int n, oy0, ox0; tile_origin(first + k * G, n, oy0, ox0);
            const char *mid = M0 + (k & 1) * PC_MBYTES;
            f32x4 acc[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int kh = t / 3, kw = t - kh * 3;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int oy = 2 * rp + j;
                    const char *p = mid + ((2 * oy + kh) * PC_MW + 2 * l15 + kw) * SP_MPIX + lq * 16;
                    const bf16x8 xh = *(const bf16x8 *)p, xl = *(const bf16x8 *)(p + 64);
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc[i][j] = mma16<true>(w1h[i][t], xh, acc[i][j]);
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc[i][j] = mma16<true>(w1l[i][t], xh, acc[i][j]);
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc[i][j] = mma16<true>(w1h[i][t], xl, acc[i][j]);
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int oy = oy0 + 2 * rp + j;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    float v[4] = {acc[i][j][0] + b1v[i][0], acc[i][j][1] + b1v[i][1], acc[i][j][2] + b1v[i][2], acc[i][j][3] + b1v[i][3]};
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = vmax_f32(v[e], v[e] * slope1);
                    uint2 H, L;
                    H.x = pack16x2<true>(v[0], v[1]); H.y = pack16x2<true>(v[2], v[3]);
                    L.x = pack16x2<true>(v[0] - unpack16_lo<true>(H.x), v[1] - unpack16_hi<true>(H.x));
                    L.y = pack16x2<true>(v[2] - unpack16_lo<true>(H.y), v[3] - unpack16_hi<true>(H.y));
                    *(uint2 *)(slab + l15 * 144 + i * 32 + lq * 8) = H;
                    *(uint2 *)(slab + l15 * 144 + 64 + i * 32 + lq * 8) = L;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (oy < a.Ho) {
                    const char *obase = (const char *)a.out + ((size_t)(n * a.Ho + oy) * a.Wo + ox0) * a.out_stride * 2 + cg * 128;      // wave-uniform
#pragma unroll
                    for (int kq = 0; kq < 2; ++kq) {
                        const int pc = lane + 64 * kq, px = pc >> 3, c16 = pc & 7;
                        const uint4 o = *(const uint4 *)(slab + px * 144 + c16 * 16);
                        if (ox0 + px < a.Wo) out_store16_at(obase, (unsigned)(px * a.out_stride * 2 + c16 * 16), o.x, o.y, o.z, o.w);
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            __syncthreads();
        }
    }
#endif
}

bool conv_stem_pair_ok(const StemPairArgs &a)
{
    return (a.in || a.in_u8) && a.w0 && a.w1 && a.out && (a.C0 == 16 || a.C0 == 32) && a.Kpad0 >= 216 && a.Kpad1 == 576 && a.out_stride >= 128 && a.H % 2 == 0 && a.W % 2 == 0 &&
           a.Ho == a.H / 2 && a.Wo == a.W / 2 && !getenv("YOLO_NO_PAIR_STEM");
}

hipError_t launch_conv_stem_pair(const StemPairArgs &a, hipStream_t s)
{
    if (!conv_stem_pair_ok(a)) return hipErrorInvalidValue;
    static int cus = 0;             // (one device model per process: MI355X)
    if (!cus) { int dev = 0; hipDeviceProp_t p; cus = hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess ? p.multiProcessorCount : 256; }
    if (!getenv("YOLO_PAIR_STEM_V1")) {          // the producer / consumer form (4 x 16 tiles); the variable: A/B against the phase-serial kernel
        hipError_t e = conv_opt_in_lds(a.in_u8 ? (const void *)conv_stem_pair_pc<true> : (const void *)conv_stem_pair_pc<false>, PC_LDS);
        if (e != hipSuccess) return e;
        const long tiles = (long)a.N * ((a.Ho + PC_TH - 1) / PC_TH) * ((a.Wo + PC_TW - 1) / PC_TW);
        const long grid = tiles < cus ? tiles : cus;
        if (a.in_u8) hipLaunchKernelGGL(conv_stem_pair_pc<true>, dim3((unsigned)grid), dim3(512), PC_LDS, s, a);
        else hipLaunchKernelGGL(conv_stem_pair_pc<false>, dim3((unsigned)grid), dim3(512), PC_LDS, s, a);
        return hipGetLastError();
    }
    hipError_t e = conv_opt_in_lds(a.in_u8 ? (const void *)conv_stem_pair_c32_c64<true> : (const void *)conv_stem_pair_c32_c64<false>, SP_LDS);
    if (e != hipSuccess) return e;
    const long tiles = (long)a.N * ((a.Ho + SP_TH - 1) / SP_TH) * ((a.Wo + SP_TW - 1) / SP_TW);
    const long grid = tiles < cus ? tiles : cus;
    if (a.in_u8) hipLaunchKernelGGL(conv_stem_pair_c32_c64<true>, dim3((unsigned)grid), dim3(512), SP_LDS, s, a);
    else hipLaunchKernelGGL(conv_stem_pair_c32_c64<false>, dim3((unsigned)grid), dim3(512), SP_LDS, s, a);
    return hipGetLastError();
}
