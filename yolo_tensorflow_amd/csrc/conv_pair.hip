// Pair-K-loop instantiations of the implicit-GEMM conv kernel (conv_igemm_kernel.h, PAIRK = true; round 6): convs of a split-fp16
// (YOLO_FP16X2) network whose INPUT is an interleaved pair tensor -- per 32-channel group 64 bytes of hi, 64 bytes of lo -- against filter rows
// packed W_hi 32 | W_lo 32.  One K-step row pair feeds three MFMA products (W_hi x_hi + W_lo x_hi + W_hi x_lo) where the plain fp16 loop
// forms two, so the 3 x matrix work of the configuration costs 2 x (not 3 x) the staging, LDS traffic, barriers and waits.
// Replaces the same reference chain as conv_igemm.hip (DN/convolutional_layer.c:445-485; slim.conv2d V3/yolo_v3.py:47-60) at the fp32
// reference's accuracy (22 significant bits per operand).  Its own translation unit so that it compiles beside the other tile tables.
#include "conv_igemm_kernel.h"
#include <cstdlib>

// tile shapes (ids and columns: conv_igemm.hip's table) instantiated with the pair K loop and the SPLIT epilogue (pairs in, pairs or an
// fp32 head out) ...
#define CONV_CFGS_PAIRK(X)                                                                             \
    X(0, 2, 2, 4, 4, 2, 64, 0)  X(2, 2, 2, 2, 4, 2, 64, 0)  X(3, 2, 2, 2, 4, 3, 64, 0)  X(4, 4, 1, 4, 2, 2, 64, 0)    \
    X(6, 2, 2, 4, 2, 2, 64, 0)  X(7, 2, 2, 4, 2, 3, 64, 0)  X(8, 4, 1, 4, 4, 2, 64, 0)  X(14, 2, 2, 2, 2, 2, 64, 0)   \
    X(15, 2, 2, 2, 2, 4, 64, 0) X(16, 1, 4, 11, 2, 2, 64, 0) X(23, 1, 4, 6, 2, 2, 64, 0) X(33, 1, 4, 11, 2, 3, 64, 0) \
    X(34, 1, 4, 6, 2, 3, 64, 0) X(49, 2, 4, 4, 2, 3, 64, 0) X(45, 1, 8, 6, 2, 3, 64, 0) X(52, 2, 4, 3, 2, 3, 64, 0)
// ... and with the ordinary fp16 epilogue (pairs in, PLAIN fp16 out: the boundaries of a mixed plan)
#define CONV_CFGS_PAIRK_PLAIN(X)                                                                       \
    X(0, 2, 2, 4, 4, 2, 64, 0)  X(2, 2, 2, 2, 4, 2, 64, 0)  X(4, 4, 1, 4, 2, 2, 64, 0)  X(6, 2, 2, 4, 2, 2, 64, 0)    \
    X(8, 4, 1, 4, 4, 2, 64, 0)  X(14, 2, 2, 2, 2, 2, 64, 0) X(16, 1, 4, 11, 2, 2, 64, 0) X(33, 1, 4, 11, 2, 3, 64, 0)      /* (every id split_default_cfg can return is here) */

bool conv_cfg_pairk_ok(int cfg, bool split_out)
{
    if (split_out) {
        if (cfg == 40 || cfg == 41 || cfg == 43 || cfg == 57 || cfg == 58) return true;      // the free-running halo forms (conv_halo13.hip)
        switch (cfg) {
#define X(id, wp, wc, tp, tc, ns, bk, nl) case id: return true;
            CONV_CFGS_PAIRK(X)
#undef X
        default: return false;
        }
    }
    switch (cfg) {
#define X(id, wp, wc, tp, tc, ns, bk, nl) case id: return true;
        CONV_CFGS_PAIRK_PLAIN(X)
#undef X
    default: return false;
    }
}

template <int WP, int WC, int TP, int TC, int NS, bool SPLIT>
static hipError_t launch_p(const ConvArgs &a, hipStream_t s)
{
    constexpr int BK = 64, BP = WP * TP * 16, BC = WC * TC * 16;
    // 32-bit buffer offsets: the activation window must stay below 2 GiB; whole 32-channel groups
    if (((double)a.N * a.H * a.W * a.in_stride + 2.0 * (a.W + 1) * a.in_stride) * 2 >= 2147483648.0) return hipErrorInvalidValue;
    if (a.Kpad % 64 || a.Cin_pad % 64 || a.kchunk != 64) return hipErrorInvalidValue;
    const long M = (long)a.N * a.Ho * a.Wo;
    const long tiles = ((M + BP - 1) / BP) * ((a.Cout + BC - 1) / BC);
    constexpr size_t lds = conv_lds_bytes<WP, WC, TP, TC, NS, BK, 0>();
    dim3 grid((unsigned)((tiles + 7) / 8 * 8)), block(64 * WP * WC);   // multiple of 8: see the XCD mapping
    const void *k = (const void *)conv_igemm<WP, WC, TP, TC, NS, BK, true, 0, false, 2, false, false, true, SPLIT, HALO_B, HALO_B, false, true>;
    if (lds > 65536) { hipError_t e = conv_opt_in_lds(k, lds); if (e != hipSuccess) return e; }
    hipLaunchKernelGGL((conv_igemm<WP, WC, TP, TC, NS, BK, true, 0, false, 2, false, false, true, SPLIT, HALO_B, HALO_B, false, true>), grid, block, lds, s, conv_tile_magic(a, BC, 0));
    return hipGetLastError();
}

hipError_t launch_conv_pair(const ConvArgs &a, int cfg, hipStream_t s)
{
    if (!a.pairk || a.in_dt != DT_F16 || (a.out_dt != DT_F16 && a.out_dt != DT_F32) || a.w2) return hipErrorInvalidValue;
    if (a.split) {
        if (a.res && a.out_dt == DT_F32) return hipErrorInvalidValue;
        if (conv_cfg_is_halo(cfg)) return launch_conv_halo13(a, cfg, s);
        switch (cfg) {
#define X(id, wp, wc, tp, tc, ns, bk, nl) case id: return launch_p<wp, wc, tp, tc, ns, true>(a, s);
            CONV_CFGS_PAIRK(X)
#undef X
        default: return hipErrorInvalidValue;
        }
    }
    if (a.out_dt != DT_F16) return hipErrorInvalidValue;          // (an fp32 head of a split network runs on the SPLIT instantiations: a.split is set for it)
    switch (cfg) {
#define X(id, wp, wc, tp, tc, ns, bk, nl) case id: return launch_p<wp, wc, tp, tc, ns, false>(a, s);
        CONV_CFGS_PAIRK_PLAIN(X)
#undef X
    default: return hipErrorInvalidValue;
    }
}

// ---------------------------------------------------------------------------------------------
// First layer of a split-fp16 network (3x3 / stride 1 / pad 1 on the image, 3 real channels padded to 8, stored as the three blocks
// hi | lo | hi): the pair counterpart of conv_c8_3x3_direct (conv_igemm.hip).  The tiled kernel reads every input pixel nine times through
// L2 -> LDS (K = 216 'channels' per output pixel) and took 391 us for 0.98 GB of compulsory traffic at 416 x 416 x 32; here a lane's MFMA B
// fragment for K-group (kk, lq) is the 16-byte hi (or lo) channel vector of ONE input pixel (tap kk * 4 + lq) straight from global memory,
// the filters (W_hi and W_lo, Cout x 96 each) live in registers, and the three products W_hi x_hi + W_lo x_hi + W_hi x_lo are formed per
// 16 pixels x 16 channels.  Output: interleaved pairs (Cout <= 32: one group, hi at element c, lo at 32 + c of the pixel's 64).
// K order = the tiled kernel's for this layer (tap-major over hi | lo | hi), three products per tap group -- the result differs from the
// tiled form in fp32 summation order only (tested against it and the emulation).
template <int TC>
__global__ __launch_bounds__(256) void conv_c8_3x3_direct_pair(const ConvArgs a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    fp16_saturating_mode();
    __shared__ __attribute__((aligned(16))) char lds[4 * 16 * 144];
    const int lane = threadIdx.x & 63, l15 = lane & 15, lq = lane >> 4;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    const int M = a.N * a.Ho * a.Wo;
    const int tiles = (M + 15) / 16;
    const bf16_t *__restrict__ in = (const bf16_t *)a.in;
    const bf16_t *__restrict__ wt = (const bf16_t *)a.wt;
    // filters: row [tap][hi 8 | hi 8 | lo 8] (yolo_pack.cpp, the image's three blocks): A fragment (i, kk) = W[channel i*16 + l15][tap kk*4 + lq]
    bf16x8 fwh[TC][3], fwl[TC][3];
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) {
            const int tap = kk * 4 + lq;
            const bf16_t *row = wt + (size_t)(i * 16 + l15) * a.Kpad + tap * 24;
            if (tap < 9) { fwh[i][kk] = *(const bf16x8 *)row; fwl[i][kk] = *(const bf16x8 *)(row + 16); }
            else { fwh[i][kk] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; fwl[i][kk] = fwh[i][kk]; }
        }
    float4 bv[TC];
#pragma unroll
    for (int i = 0; i < TC; ++i) bv[i] = *(const float4 *)(a.bias + i * 16 + lq * 4);
    const float slope = a.act == ACT_LEAKY ? 0.1f : 1.0f;
    const int HoWo = a.Ho * a.Wo;
    constexpr int U = 2;                        // 16-pixel tiles in flight per wave
    const long groups = (tiles + U - 1) / U;
    for (long g = wave; g < groups; g += nwaves) {
        bf16x8 fxh[U][3], fxl[U][3];
        int mrow[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int m = (int)(g * U + u) * 16 + l15;
            const bool mv = m < M;
            mrow[u] = mv ? m : -1;
            const int mm = mv ? m : M - 1;
            const int n = fast_div(mm, a.howo_mul, a.howo_shift), rem = mm - n * HoWo;
            const int oy = fast_div(rem, a.wo_mul, a.wo_shift), ox = rem - oy * a.Wo;
#pragma unroll
            for (int kk = 0; kk < 3; ++kk) {
                const int tap = kk * 4 + lq;
                const int kh = (tap * 11) >> 5, kw = tap - kh * 3;
                const int iy = oy + kh - 1, ix = ox + kw - 1;
                const bool ok = mv && tap < 9 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                const bf16_t *p = ok ? in + ((size_t)(n * a.H + iy) * a.W + ix) * a.in_stride : (const bf16_t *)a.zeros;
                fxh[u][kk] = *(const bf16x8 *)p; fxl[u][kk] = *(const bf16x8 *)(p + (ok ? 8 : 0));
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            f32x4 acc[TC];
#pragma unroll
            for (int i = 0; i < TC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 3; ++kk) {
#pragma unroll
                for (int i = 0; i < TC; ++i) acc[i] = mma16<true>(fwh[i][kk], fxh[u][kk], acc[i]);
#pragma unroll
                for (int i = 0; i < TC; ++i) acc[i] = mma16<true>(fwl[i][kk], fxh[u][kk], acc[i]);
#pragma unroll
                for (int i = 0; i < TC; ++i) acc[i] = mma16<true>(fwh[i][kk], fxl[u][kk], acc[i]);
            }
            // the 16 x (32 hi | 32 lo) tile leaves through a wave-private LDS slab (pixel pitch 144 B: conflict-free for the 8-byte writes of the
            // accumulator layout) as whole 128-byte pixel rows, 16 bytes per lane, written through (sc1) like every other tensor store:
            // straight from the accumulators a store instruction would touch sixteen lines 32 bytes at a time
            char *slab = lds + (threadIdx.x >> 6) * (16 * 144);
#pragma unroll
            for (int i = 0; i < TC; ++i) {
                float v[4] = {acc[i][0] + bv[i].x, acc[i][1] + bv[i].y, acc[i][2] + bv[i].z, acc[i][3] + bv[i].w};
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = vmax_f32(v[q], v[q] * slope);
                uint2 H, L;
                H.x = pack16x2<true>(v[0], v[1]); H.y = pack16x2<true>(v[2], v[3]);
                L.x = pack16x2<true>(v[0] - unpack16_lo<true>(H.x), v[1] - unpack16_hi<true>(H.x));
                L.y = pack16x2<true>(v[2] - unpack16_lo<true>(H.y), v[3] - unpack16_hi<true>(H.y));
                *(uint2 *)(slab + l15 * 144 + i * 32 + lq * 8) = H;
                *(uint2 *)(slab + l15 * 144 + 64 + i * 32 + lq * 8) = L;
            }
            if (TC == 1) {          // 16 filters: the group's other 16 channels are zeros (the consumer reads whole 32-channel groups)
                *(uint2 *)(slab + l15 * 144 + 32 + lq * 8) = uint2{0u, 0u};
                *(uint2 *)(slab + l15 * 144 + 96 + lq * 8) = uint2{0u, 0u};
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const long m0 = (long)(g * U + u) * 16;                       // wave-uniform
            const char *obase = (const char *)a.out + (size_t)m0 * a.out_stride * 2;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int pc = lane + 64 * k, px = pc >> 3, c16 = pc & 7;
                const uint4 o = *(const uint4 *)(slab + px * 144 + c16 * 16);
                if (m0 + px < M) out_store16_at(obase, (unsigned)(px * a.out_stride * 2 + c16 * 16), o.x, o.y, o.z, o.w);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the slab is rewritten by the next tile
        }
    }
#endif
}

bool conv_c8_direct_pair_ok(const ConvArgs &a)
{
    return a.in_dt == DT_F16 && a.out_dt == DT_F16 && a.split && !a.pairk && a.ksize == 3 && a.stride == 1 && a.pad == 1 && a.Cin_pad == 24 && a.in_stride == 24 &&
           a.kchunk == 24 && !a.res && !a.w2 && a.Kpad >= 216 && (a.Cout == 16 || a.Cout == 32) && a.out_stride >= 64 && !getenv("YOLO_NO_PAIR_DIRECT");      // (the variable: A/B against the tiled kernel)
}

hipError_t launch_conv_c8_direct_pair(const ConvArgs &a, hipStream_t s)
{
    if (!conv_c8_direct_pair_ok(a)) return hipErrorInvalidValue;
    const long M = (long)a.N * a.Ho * a.Wo;
    long waves = (M + 31) / 32;
    long blocks = (waves + 3) / 4; if (blocks > 256 * 8) blocks = 256 * 8;
    dim3 grid((unsigned)blocks), block(256);
    if (a.Cout == 16) hipLaunchKernelGGL((conv_c8_3x3_direct_pair<1>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((conv_c8_3x3_direct_pair<2>), grid, block, 0, s, a);
    return hipGetLastError();
}
