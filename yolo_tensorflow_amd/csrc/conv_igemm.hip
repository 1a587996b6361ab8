// Fused convolution for gfx950: im2col-free implicit GEMM on bf16 MFMA, NHWC activations.
//
//   out[n,oy,ox,co] = act( sum_{kh,kw,ci} in[n, oy*s+kh-p, ox*s+kw-p, ci] * W[co][kh][kw][ci] + bias[co] ) (+ residual)
//
// Replaces, in one launch, the reference's per-layer chain fill -> im2col -> sgemm -> normalize ->
// scale_bias -> add_bias -> activate (-> shortcut)  (DN/convolutional_layer.c:445-485,
// DN/convolutional_kernels.cu:73-135, DN/blas_kernels.cu:12-75,194-201,711-745) and TF's
// Conv2D + FusedBatchNorm + LeakyRelu (+ Add) behind slim.conv2d (V3/yolo_v3.py:47-60).
//
// GEMM view: D[channel][pixel] = sum_k Wt[channel][k] * X[pixel][k], k = (kh*K+kw)*Cin_pad + ci.
//   * MFMA A operand = filter rows, B operand = activation rows  -> each lane ends up holding 4
//     consecutive output channels of ONE pixel (D row = 4*(lane>>4)+reg, col = lane&15), i.e. an 8-byte
//     (bf16) or 16-byte (fp32) contiguous NHWC store.
//   * K is walked in 64-wide steps; both operand tiles are staged in LDS as [rows][64] bf16 (128-B rows),
//     16-B chunk index XOR-swizzled with (row & 7) so the ds_read_b128 fragment reads are conflict free.
//     LDS-DMA destinations are lane-linear, so the swizzle is applied to the SOURCE offset and to the read.
//   * Activation rows are gathered straight from the NHWC tensor (no im2col buffer) with
//     `buffer_load_dwordx4 ... offen lds`: per lane a CONSTANT 32-bit byte offset (its pixel, its 16-B chunk),
//     per K-step one SCALAR offset (tap and channel base) -- the address arithmetic of a K-step is a handful of
//     SALU instructions, not ~20 VALU per load (which had made the loop VALU-bound).  Padding taps and rows
//     past the last pixel use an out-of-range offset: the buffer range check makes the DMA write zeros
//     (verified on MI355X by tools/probe/lds_dma_oob.hip).
//   * NS-stage LDS ring: the loads of K-step t+NS-1 are issued before the MFMAs of step t; a counted
//     s_waitcnt vmcnt + one raw s_barrier per K-step keep them in flight across the barrier.
#include "kernels.h"
#include <cstdio>
#include <mutex>
#include <set>
#include <utility>

hipError_t conv_opt_in_lds(const void *kernel, size_t lds_bytes)
{
    static std::mutex mu;
    static std::set<std::pair<int, const void *>> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev); if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> g(mu);
    if (done.count({dev, kernel})) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e == hipSuccess) done.insert({dev, kernel});
    return e;
}

#include "conv_igemm_kernel.h"


template <int EB>
static hipError_t launch_conv_diag_t(const ConvArgs &a, hipStream_t s)
{
    // diagnostic instantiation of ONE configuration (p176c128_s2, uniform tap)
    constexpr int WP = 1, WC = 4, TP = 11, TC = 2, NS = 2, BK = 64;
    constexpr int BP = WP * TP * 16, BC = WC * TC * 16, NW = WP * WC;
    const long M = (long)a.N * a.Ho * a.Wo;
    const long tiles = ((M + BP - 1) / BP) * ((a.Cout + BC - 1) / BC);
    constexpr size_t lds = conv_lds_bytes<WP, WC, TP, TC, NS, BK>();
    { hipError_t e = conv_opt_in_lds((const void *)conv_igemm<WP, WC, TP, TC, NS, BK, true, 0, true, EB>, lds); if (e != hipSuccess) return e; }
    hipLaunchKernelGGL((conv_igemm<WP, WC, TP, TC, NS, BK, true, 0, true, EB>), dim3((unsigned)((tiles + 7) / 8 * 8)), dim3(64 * NW), lds, s, conv_tile_magic(a, WC * TC * 16, 0));
    return hipGetLastError();
}
hipError_t launch_conv_diag(const ConvArgs &a, hipStream_t s) { return a.in_dt == DT_FP8 ? launch_conv_diag_t<1>(a, s) : launch_conv_diag_t<2>(a, s); }

// ---------------------------------------------------------------------------------------------
// First layer (3x3, stride 1, 3 real input channels padded to 8): HBM-bound (it writes N*H*W*Cout bf16), K is only
// 9 taps x 8 channels = 72 (padded to 96).  No LDS: a lane's MFMA B fragment for K-group (kk, lq) is exactly the
// 16-byte channel vector of ONE input pixel (tap kk*4+lq), so it is a single global_load_dwordx4; the filters
// (Cout x 96 bf16) live in registers for the whole wave.  One wave computes 16 pixels x Cout per step.
template <int TC, bool H16 = false>
__global__ __launch_bounds__(256) void conv_c8_3x3_direct(const ConvArgs a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (H16) fp16_saturating_mode();
    const int lane = threadIdx.x & 63, l15 = lane & 15, lq = lane >> 4;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    const int M = a.N * a.Ho * a.Wo;
    const int tiles = (M + 15) / 16;
    const bf16_t *__restrict__ in = (const bf16_t *)a.in;
    const bf16_t *__restrict__ wt = (const bf16_t *)a.wt;

    // filters: A fragment (i, kk) = W[channel i*16 + l15][k = (kk*4 + lq)*8 .. +7]; Kpad >= 96 holds zeros past k = 72
    bf16x8 fw[TC][3];
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int kk = 0; kk < 3; ++kk)
            fw[i][kk] = *(const bf16x8 *)(wt + (size_t)(i * 16 + l15) * a.Kpad + (kk * 4 + lq) * 8);
    float4 bv[TC];
#pragma unroll
    for (int i = 0; i < TC; ++i) bv[i] = *(const float4 *)(a.bias + i * 16 + lq * 4);

    const int HoWo = a.Ho * a.Wo;
    constexpr int U = 4;                        // 16-pixel tiles in flight per wave (memory-level parallelism)
    const long groups = (tiles + U - 1) / U;
    for (long g = wave; g < groups; g += nwaves) {
        bf16x8 fx[U][3];
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
                fx[u][kk] = *(const bf16x8 *)p;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            f32x4 acc[TC];
#pragma unroll
            for (int i = 0; i < TC; ++i) {
                acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 3; ++kk) acc[i] = mma16<H16>(fw[i][kk], fx[u][kk], acc[i]);
            }
            if (mrow[u] >= 0) {
#pragma unroll
                for (int i = 0; i < TC; ++i) {
                    float v[4] = {acc[i][0] + bv[i].x, acc[i][1] + bv[i].y, acc[i][2] + bv[i].z, acc[i][3] + bv[i].w};
                    if (a.act == ACT_LEAKY)
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.1f * v[q]);     // == v > 0 ? v : 0.1 v
                    uint2 pk;
                    pk.x = pack16x2<H16>(v[0], v[1]);
                    pk.y = pack16x2<H16>(v[2], v[3]);
                    if (!H16 && a.out_dt == DT_FP8) {
                        // same value chain as the tiled kernel: bf16 rounding, then e4m3 of value / scale_out
                        const float s8 = a.out_inv_scale;
                        const uint32_t w8 = f32x2_to_fp8<true>(bf16_bits_to_f32(pk.y & 0xffff) * s8, bf16_bits_to_f32(pk.y >> 16) * s8,
                                                               f32x2_to_fp8<false>(bf16_bits_to_f32(pk.x & 0xffff) * s8, bf16_bits_to_f32(pk.x >> 16) * s8, 0));
                        *(uint32_t *)((char *)a.out + (size_t)mrow[u] * a.out_stride + i * 16 + lq * 4) = w8;
                    } else
                        *(uint2 *)((bf16_t *)a.out + (size_t)mrow[u] * a.out_stride + i * 16 + lq * 4) = pk;
                }
            }
        }
    }
#endif
}

bool conv_c8_direct_ok(const ConvArgs &a)
{
    return (a.in_dt == DT_BF16 || (a.in_dt == DT_F16 && a.out_dt == DT_F16)) && a.ksize == 3 && a.stride == 1 && a.pad == 1 && a.Cin_pad == 8 && a.out_dt != DT_F32 && !a.res && a.Kpad >= 96 &&
           (a.Cout == 16 || a.Cout == 32 || a.Cout == 64);
}

hipError_t launch_conv_c8_direct(const ConvArgs &a, hipStream_t s)
{
    const long M = (long)a.N * a.Ho * a.Wo;
    long waves = (M + 63) / 64;
    long blocks = (waves + 3) / 4; if (blocks > 256 * 8) blocks = 256 * 8;
    dim3 grid((unsigned)blocks), block(256);
    if (a.in_dt == DT_F16) {
        if (a.Cout == 16) hipLaunchKernelGGL((conv_c8_3x3_direct<1, true>), grid, block, 0, s, a);
        else if (a.Cout == 32) hipLaunchKernelGGL((conv_c8_3x3_direct<2, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((conv_c8_3x3_direct<4, true>), grid, block, 0, s, a);
    }
    else if (a.Cout == 16) hipLaunchKernelGGL((conv_c8_3x3_direct<1>), grid, block, 0, s, a);
    else if (a.Cout == 32) hipLaunchKernelGGL((conv_c8_3x3_direct<2>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((conv_c8_3x3_direct<4>), grid, block, 0, s, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Tile configurations: (waves along pixels, waves along channels, 16-px tiles per wave, 16-ch tiles per wave, LDS
// stages, K-step).  Pixel-tile heights that are not powers of two exist so the autotuner can make the tile count a
// near multiple of 256 CUs x resident workgroups (wave quantisation), e.g. 176 px for M = 32 * 26 * 26.
// BK = 32 halves the staging LDS (three or four workgroups per CU) and makes Cin = 32 layers uniform-tap; measured it
// only pays on the early, short-K layers -- on the deep 3x3 layers the extra barriers cost more than the occupancy
// buys (0.066 vs 0.053 ms) -- so only a few BK = 32 shapes are kept.
#define CONV_CFGS(X)                                                                                   \
    X(0, 2, 2, 4, 4, 2, 64, 0)  X(1, 2, 2, 4, 4, 3, 64, 0)  X(2, 2, 2, 2, 4, 2, 64, 0)  X(3, 2, 2, 2, 4, 3, 64, 0)    \
    X(4, 4, 1, 4, 2, 2, 64, 0)  X(5, 4, 1, 4, 2, 3, 64, 0)  X(6, 2, 2, 4, 2, 2, 64, 0)  X(7, 2, 2, 4, 2, 3, 64, 0)    \
    X(8, 4, 1, 4, 4, 2, 64, 0)  X(9, 4, 1, 4, 4, 3, 64, 0)  X(10, 4, 2, 4, 4, 2, 64, 0) X(11, 4, 2, 4, 4, 3, 64, 0)   \
    X(12, 2, 4, 4, 4, 2, 64, 0) X(13, 2, 4, 4, 4, 3, 64, 0) X(14, 2, 2, 2, 2, 2, 64, 0) X(15, 2, 2, 2, 2, 4, 64, 0)   \
    X(16, 1, 4, 11, 2, 2, 64, 0) X(17, 1, 4, 11, 4, 2, 64, 0) X(18, 1, 4, 11, 1, 2, 64, 0) X(19, 1, 4, 10, 2, 2, 64, 0) \
    X(20, 1, 4, 12, 2, 2, 64, 0) X(21, 1, 4, 9, 2, 2, 64, 0) X(22, 1, 4, 13, 2, 2, 64, 0) X(23, 1, 4, 6, 2, 2, 64, 0)  \
    X(24, 1, 4, 7, 2, 2, 64, 0)                                                                            \
    X(25, 1, 4, 11, 2, 2, 32, 0) X(26, 2, 2, 4, 2, 2, 32, 0)  X(27, 4, 1, 4, 2, 2, 32, 0)  X(28, 4, 1, 4, 4, 2, 32, 0)  \
    X(29, 2, 2, 2, 2, 2, 32, 0)  X(30, 2, 2, 2, 4, 2, 32, 0)                                                \
    X(31, 1, 8, 11, 2, 2, 64, 4) X(32, 1, 8, 11, 2, 2, 64, 0)                                              \
    X(33, 1, 4, 11, 2, 3, 64, 0) X(34, 1, 4, 6, 2, 3, 64, 0)                                              \
    X(35, 2, 4, 3, 4, 2, 64, 0)
// halo-staged 3x3 configurations (conv_halo13.hip; same columns): one 13x13 pixel block x (wc * tc * 16) channels per workgroup
#define CONV_CFGS_HALO(X)                                                                              \
    X(36, 1, 8, 11, 2, 2, 64, 0) X(37, 1, 8, 11, 2, 2, 64, 4) X(38, 1, 4, 11, 2, 2, 64, 4) X(39, 1, 4, 11, 2, 2, 64, 0)   \
    X(40, 1, 8, 11, 2, 2, 64, 0) X(41, 1, 8, 11, 1, 2, 64, 0) X(42, 1, 4, 11, 2, 2, 64, 0)        /* 40-: free-running waves */ \
    X(43, 1, 8, 11, 1, 3, 64, 0)
// round 4 (ids follow the halo block: a configuration's id is its index in kCfgs): whole-Cout tiles for the stand-alone 1x1 layers
// (every activation row enters ONE CU) and small-batch shapes.  (No tile wider than 256 channels: filters and bias are padded to
// multiples of 256 rows, cout_pad.)
#define CONV_CFGS_B(X)                                                                                 \
    X(44, 1, 8, 6, 2, 2, 64, 0) X(45, 1, 8, 6, 2, 3, 64, 0) X(46, 2, 4, 3, 4, 2, 64, 0) X(47, 2, 4, 2, 4, 2, 64, 0)    \
    X(48, 2, 4, 3, 2, 2, 64, 0) X(49, 2, 4, 4, 2, 3, 64, 0) X(50, 1, 8, 4, 2, 3, 64, 0) X(51, 2, 4, 2, 2, 3, 64, 0)    \
    X(52, 2, 4, 3, 2, 3, 64, 0) X(53, 2, 4, 3, 2, 4, 64, 0)
// (Tried and dropped, round 4: the free-running halo form with ONE wave per SIMD -- four waves of 176 x 64, 40 % fewer LDS bytes per FLOP
// than eight of 176 x 32, whose stamped K loop needs 1 708 cycles per K-step against 1 862.  In the network it LOSES: 26x26 layers 0.412 ms
// against 0.387 for the eleven of them, 52x52 0.495 against 0.460, 13x13 0.307 against 0.289 -- set-up and epilogue are serial in a wave,
// and with nobody else on the SIMD nothing runs under them.)

// split fp16 storage (YOLO_FP16X2): the tile shapes instantiated with the two-pass epilogue
#define CONV_CFGS_SPLIT(X)                                                                             \
    X(0, 2, 2, 4, 4, 2, 64, 0)  X(2, 2, 2, 2, 4, 2, 64, 0)  X(3, 2, 2, 2, 4, 3, 64, 0)  X(4, 4, 1, 4, 2, 2, 64, 0)    \
    X(6, 2, 2, 4, 2, 2, 64, 0)  X(7, 2, 2, 4, 2, 3, 64, 0)  X(8, 4, 1, 4, 4, 2, 64, 0)  X(14, 2, 2, 2, 2, 2, 64, 0)   \
    X(15, 2, 2, 2, 2, 4, 64, 0) X(16, 1, 4, 11, 2, 2, 64, 0) X(23, 1, 4, 6, 2, 2, 64, 0) X(33, 1, 4, 11, 2, 3, 64, 0) \
    X(34, 1, 4, 6, 2, 3, 64, 0) X(49, 2, 4, 4, 2, 3, 64, 0) X(45, 1, 8, 6, 2, 3, 64, 0) X(52, 2, 4, 3, 2, 3, 64, 0)                    \
    X(25, 1, 4, 11, 2, 2, 32, 0) X(26, 2, 2, 4, 2, 2, 32, 0) X(27, 4, 1, 4, 2, 2, 32, 0) X(28, 4, 1, 4, 4, 2, 32, 0)    /* 64-byte rows: 3 x 32 = 96 input 'channels' are uniform-tap */
#define CONV_CFGS_SPLIT_HALO(X) X(40, 1, 8, 11, 2, 2, 64, 0) X(41, 1, 8, 11, 1, 2, 64, 0) X(43, 1, 8, 11, 1, 3, 64, 0)
bool conv_cfg_split_ok(int cfg)
{
    switch (cfg) {
#define X(id, wp, wc, tp, tc, ns, bk, nl) case id: return true;
        CONV_CFGS_SPLIT(X) CONV_CFGS_SPLIT_HALO(X)
#undef X
    default: return false;
    }
}

// round 5: free-running halo forms on rectangular blocks (conv_halo13.hip): 10 x 19 (12 sub-tiles) x 256 / 128 channels, 5 x 19 (6 sub-tiles) x 128
// -- the 608 x 608 network's 76 / 38 / 19 grids tile into them exactly, 256 workgroups each at 8 images per GPU
#define CONV_CFGS_HALO_R(X) X(54, 1, 8, 12, 2, 2, 64, 0) X(55, 1, 8, 12, 1, 3, 64, 0) X(56, 1, 8, 6, 1, 3, 64, 0)
struct CfgDesc { int id, wp, wc, tp, tc, ns, bk, nl, halo; };
#define X(id, wp, wc, tp, tc, ns, bk, nl) {id, wp, wc, tp, tc, ns, bk, nl, 0},
#define XH(id, wp, wc, tp, tc, ns, bk, nl) {id, wp, wc, tp, tc, ns, bk, nl, 1},
// round 6: free-running 13 x 13-block halo forms with ONE wave per SIMD, for the pair K loop (conv_halo13.hip, split-fp16 only): four waves of
// 176 x 64 (57) / 176 x 32 (58) read every pixel fragment four times per K-step instead of eight -- 42 % fewer LDS bytes per MFMA in a loop whose
// LDS reads (1 952 cycles per K-step) sit just under its MFMAs (2 112)
#define CONV_CFGS_HALO_P(X) X(57, 1, 4, 11, 4, 2, 64, 0) X(58, 1, 4, 11, 2, 2, 64, 0)
static const CfgDesc kCfgs[] = {CONV_CFGS(X) CONV_CFGS_HALO(XH) CONV_CFGS_B(X) CONV_CFGS_HALO_R(XH) CONV_CFGS_HALO_P(XH)};
#undef X
#undef XH
int conv_num_cfgs() { return (int)(sizeof(kCfgs) / sizeof(kCfgs[0])); }
bool conv_cfg_is_halo(int cfg) { return cfg >= 0 && cfg < conv_num_cfgs() && kCfgs[cfg].halo; }
bool conv_cfg_tail_ok(int cfg, int cout, bool fp8, bool head)
{
    if (cfg < 0 || cfg >= conv_num_cfgs()) return false;
    const CfgDesc &c = kCfgs[cfg];
    const int bc = c.wc * c.tc * 16;
    if (head) return !fp8 && (cfg == 40 || cfg == 54) && cout == 256;      // a head as the tail: the free-running 176 x 256 / 10 x 19 x 256 halo forms have a HEADT instantiation
    return c.wp == 1 && c.wc == 8 && (fp8 ? bc == 256 : c.nl == 0 && (bc == 256 || bc == 128)) && bc == cout;      // (as TAIL_OK in the kernel)
}
const char *conv_cfg_name(int cfg)
{
    static char names[64][32];
    if (cfg < 0 || cfg >= conv_num_cfgs()) return cfg == CONV_CFG_DIRECT ? "direct_c8" : "?";
    const CfgDesc &c = kCfgs[cfg];
    if (c.id >= 57) { snprintf(names[cfg], sizeof names[cfg], "f176c%d_w4_pair", c.wc * c.tc * 16); return names[cfg]; }
    if (c.id >= 54) { snprintf(names[cfg], sizeof names[cfg], "f%sc%d_s%d", c.tp == 12 ? "10x19" : "5x19", c.wc * c.tc * 16, c.ns); return names[cfg]; }
    snprintf(names[cfg], sizeof names[cfg], "%s%dc%d_s%d_k%d%s%d", c.halo ? (c.id >= 40 ? "f" : "h") : "p", c.wp * c.tp * 16, c.wc * c.tc * 16, c.ns, c.bk, c.nl ? "_L" : "_w", c.nl ? c.nl : c.wp * c.wc);
    return names[cfg];
}

int conv_pick_cfg(const ConvArgs &a)
{
    if (conv_c8_direct_ok(a)) return CONV_CFG_DIRECT;
    const long M = (long)a.N * a.Ho * a.Wo;
    if (a.Cout <= 32) return 4;
    if (a.Cout <= 64) return M >= 65536 ? 8 : 6;
    const long tiles128 = ((M + 127) / 128) * ((a.Cout + 127) / 128);
    if (tiles128 < 512) return M < 8192 && tiles128 < 128 ? 14 : 2;
    return 0;
}

template <int WP, int WC, int TP, int TC, int NS, int BK, int NL, bool UNI, int EB, bool H16 = false, bool SPLIT = false>
static hipError_t launch_u(const ConvArgs &a, hipStream_t s)
{
    constexpr int BP = WP * TP * 16, BC = WC * TC * 16;
    const long M = (long)a.N * a.Ho * a.Wo;
    const long tiles = ((M + BP - 1) / BP) * ((a.Cout + BC - 1) / BC);
    constexpr size_t lds = conv_lds_bytes<WP, WC, TP, TC, NS, BK, NL>();
    dim3 grid((unsigned)((tiles + 7) / 8 * 8)), block(64 * (WP * WC + NL));   // multiple of 8: see the XCD mapping
    if (lds > 65536) {
        hipError_t e = conv_opt_in_lds((const void *)conv_igemm<WP, WC, TP, TC, NS, BK, UNI, NL, false, EB, false, false, H16, SPLIT>, lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((conv_igemm<WP, WC, TP, TC, NS, BK, UNI, NL, false, EB, false, false, H16, SPLIT>), grid, block, lds, s, conv_tile_magic(a, BC, 0));
    return hipGetLastError();
}

template <int WP, int WC, int TP, int TC, int NS, int BK, int NL, int EB, bool H16 = false, bool SPLIT = false>
static hipError_t launch_t(const ConvArgs &a, hipStream_t s)
{
    // 32-bit buffer offsets: the activation window must stay below 2 GiB
    if (((double)a.N * a.H * a.W * a.in_stride + 2.0 * (a.W + 1) * a.in_stride) * EB >= 2147483648.0) return hipErrorInvalidValue;
    constexpr int BKE = BK * 2 / EB;
    if (a.Kpad % BKE) return hipErrorInvalidValue;
    return (a.Cin_pad % BKE) == 0 ? launch_u<WP, WC, TP, TC, NS, BK, NL, true, EB, H16, SPLIT>(a, s) : launch_u<WP, WC, TP, TC, NS, BK, NL, false, EB, H16, SPLIT>(a, s);
}

hipError_t launch_conv_bf16(const ConvArgs &a, int cfg, hipStream_t s)
{
    // (16-bit storage: bf16, or fp16 -- the same tile table, the same kernels with the other MFMA and conversions)
    if (a.in_dt != DT_BF16 && a.in_dt != DT_F16) return hipErrorInvalidValue;
    if (a.in_dt == DT_F16 && a.out_dt != DT_F16 && a.out_dt != DT_F32) return hipErrorInvalidValue;
    if (cfg == CONV_CFG_DIRECT && a.split) return launch_conv_c8_direct_pair(a, s);      // the first layer of a split-fp16 network: direct kernel (conv_pair.hip)
    if (a.pairk) return launch_conv_pair(a, cfg, s);          // the input is an interleaved pair tensor: the pair K loop (conv_pair.hip, conv_halo13.hip)
    if (a.split) {
        // split fp16 storage (YOLO_FP16X2), PLAIN input (or the network input's three blocks): fp16 operands, the ordinary K loop, 16-bit
        // outputs written as interleaved pairs (fp32 head outputs as ever); a subset of the tile table is instantiated
        if (a.in_dt != DT_F16 || (a.out_dt != DT_F16 && a.out_dt != DT_F32) || a.w2 || (a.res && a.out_dt == DT_F32)) return hipErrorInvalidValue;
        switch (cfg) {
#define X(id, wp, wc, tp, tc, ns, bk, nl) case id: return launch_t<wp, wc, tp, tc, ns, bk, nl, 2, true, true>(a, s);
            CONV_CFGS_SPLIT(X)
#undef X
#define X(id, wp, wc, tp, tc, ns, bk, nl) case id: return launch_conv_halo13(a, id, s);
            CONV_CFGS_SPLIT_HALO(X)
#undef X
        default: return hipErrorInvalidValue;
        }
    }
    if (cfg == CONV_CFG_DIRECT) return conv_c8_direct_ok(a) ? launch_conv_c8_direct(a, s) : hipErrorInvalidValue;
    if (a.in_dt == DT_F16)
        switch (cfg) {
#define X(id, wp, wc, tp, tc, ns, bk, nl) case id: return launch_t<wp, wc, tp, tc, ns, bk, nl, 2, true>(a, s);
            CONV_CFGS(X) CONV_CFGS_B(X)
#undef X
        default: break;
        }
    switch (cfg) {
#define X(id, wp, wc, tp, tc, ns, bk, nl) case id: return launch_t<wp, wc, tp, tc, ns, bk, nl, 2>(a, s);
        CONV_CFGS(X) CONV_CFGS_B(X)
#undef X
#define X(id, wp, wc, tp, tc, ns, bk, nl) case id: return launch_conv_halo13(a, id, s);
        CONV_CFGS_HALO(X) CONV_CFGS_HALO_R(X) CONV_CFGS_HALO_P(X)
#undef X
    default: return hipErrorInvalidValue;
    }
}

// fp8 operands: the same tile ids, restricted to the two-stage 128-B-row symmetric shapes that the bf16 tuning kept
#define CONV_CFGS_FP8(X)                                                                                     \
    X(0, 2, 2, 4, 4, 2, 64, 0)  X(2, 2, 2, 2, 4, 2, 64, 0)  X(4, 4, 1, 4, 2, 2, 64, 0)  X(6, 2, 2, 4, 2, 2, 64, 0)    \
    X(8, 4, 1, 4, 4, 2, 64, 0)  X(12, 2, 4, 4, 4, 2, 64, 0) X(14, 2, 2, 2, 2, 2, 64, 0) X(16, 1, 4, 11, 2, 2, 64, 0)  \
    X(17, 1, 4, 11, 4, 2, 64, 0) X(19, 1, 4, 10, 2, 2, 64, 0) X(20, 1, 4, 12, 2, 2, 64, 0) X(23, 1, 4, 6, 2, 2, 64, 0) \
    X(32, 1, 8, 11, 2, 2, 64, 0) X(31, 1, 8, 11, 2, 2, 64, 4) X(33, 1, 4, 11, 2, 3, 64, 0) X(34, 1, 4, 6, 2, 3, 64, 0) \
    X(45, 1, 8, 6, 2, 3, 64, 0) X(48, 2, 4, 3, 2, 2, 64, 0) X(49, 2, 4, 4, 2, 3, 64, 0) X(51, 2, 4, 2, 2, 3, 64, 0)      /* round 4: the 8-wave / 3-stage shapes */
bool conv_cfg_fp8_ok(int cfg)
{
    switch (cfg) {
#define X(id, wp, wc, tp, tc, ns, bk, nl) case id: return true;
        CONV_CFGS_FP8(X)
        CONV_CFGS_HALO(X)
#undef X
    default: return false;
    }
}
hipError_t launch_conv_fp8(const ConvArgs &a, int cfg, hipStream_t s)
{
    if (a.in_dt != DT_FP8) return hipErrorInvalidValue;
    switch (cfg) {
#define X(id, wp, wc, tp, tc, ns, bk, nl) case id: return launch_t<wp, wc, tp, tc, ns, bk, nl, 1>(a, s);
        CONV_CFGS_FP8(X)
#undef X
#define X(id, wp, wc, tp, tc, ns, bk, nl) case id: return launch_conv_halo13(a, id, s);
        CONV_CFGS_HALO(X)
#undef X
    default: return hipErrorInvalidValue;
    }
}
