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
//   * Activation rows are gathered straight from the NHWC tensor (no im2col buffer): each 16-B chunk of
//     a K-step is one (tap, 8-channel) slice of one input pixel; padding taps read a zero line.
//   * Double-buffered LDS, global loads for step t+1 are issued before the MFMAs of step t and written
//     to LDS after them (one barrier per K-step).
#include "kernels.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t f32_to_bf16_rn(float f)
{
    __bf16 b = (__bf16)f;                       // v_cvt_pk_bf16_f32: RNE, NaN preserved
    return (uint32_t)__builtin_bit_cast(uint16_t, b);
}
__device__ __forceinline__ float bf16_bits_to_f32(uint32_t b) { return __builtin_bit_cast(float, b << 16); }

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

template <int WP, int WC, int TP, int TC, bool OUT_F32>
__global__ __launch_bounds__(64 * WP * WC) void conv_igemm_bf16(const ConvArgs a)
{
    constexpr int NT = 64 * WP * WC;
    constexpr int BP = WP * TP * 16;           // output pixels per workgroup
    constexpr int BC = WC * TC * 16;           // output channels per workgroup
    constexpr int RPP = NT / 8;                // tile rows covered by one pass of 16-B loads
    constexpr int LA = BP / RPP;
    constexpr int LB = BC / RPP;
    static_assert(BP % RPP == 0 && BC % RPP == 0, "tile/threads mismatch");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *sX = smem;                           // [2][BP][128 B]
    char *sW = smem + 2 * BP * 128;            // [2][BC][128 B]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wpi = wid % WP, wci = wid / WP;

    // XCD-aware tile assignment: workgroups b, b+8, b+16.. share an XCD (and its L2); give each XCD a
    // contiguous run of tiles ordered pixel-tile-major so that its resident workgroups re-use the same
    // activation rows (all channel tiles of a pixel tile) and walk the filter slices together.
    const int M = a.N * a.Ho * a.Wo;
    const int tilesC = (a.Cout + BC - 1) / BC;
    const int tilesP = (M + BP - 1) / BP;
    const int per_xcd = gridDim.x >> 3;
    const int tile = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (tile >= tilesP * tilesC) return;
    const int ct = tile % tilesC;
    const int pt = tile / tilesC;

    const bf16_t *__restrict__ in = (const bf16_t *)a.in;
    const bf16_t *__restrict__ wt = (const bf16_t *)a.wt;

    // ---- per-thread staging geometry: this thread always fetches LDS slot (row r0 + i*RPP, physical
    //      chunk tid&7), i.e. logical K-chunk (tid&7) ^ (row&7) of that row (rule: swizzle the SOURCE) ----
    const int r0 = tid >> 3;
    const int chunk = (tid & 7) ^ (r0 & 7);
    int pixbase[LA], iy0[LA], ix0[LA];
    const int HoWo = a.Ho * a.Wo;
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        int m = pt * BP + r0 + i * RPP;
        if (m < M) {
            int n = m / HoWo;
            int rem = m - n * HoWo;
            int oy = rem / a.Wo;
            int ox = rem - oy * a.Wo;
            pixbase[i] = n * a.H * a.W;
            iy0[i] = oy * a.stride - a.pad;
            ix0[i] = ox * a.stride - a.pad;
        } else {
            pixbase[i] = 0; iy0[i] = -(1 << 20); ix0[i] = 0;
        }
    }
    const bf16_t *wrow[LB];
#pragma unroll
    for (int i = 0; i < LB; ++i)
        wrow[i] = wt + (size_t)(ct * BC + r0 + i * RPP) * a.Kpad + chunk * 8;

    const int KK = a.ksize * a.ksize;
    int kc = chunk * 8, tap = 0;
    while (kc >= a.Cin_pad) { kc -= a.Cin_pad; ++tap; }

    // global -> LDS direct (global_load_lds_dwordx4): one wave instruction fills 8 tile rows (1 KiB)
    auto stage = [&](int kt, int buf) {
        int kh = 0, kw = 0;
        if (a.ksize == 3) { kh = (tap * 11) >> 5; kw = tap - kh * 3; }
        else if (a.ksize != 1) { kh = tap / a.ksize; kw = tap - kh * a.ksize; }
        const bool tap_ok = tap < KK;
        char *dx = sX + buf * (BP * 128) + wid * 1024;
        char *dw = sW + buf * (BC * 128) + wid * 1024;
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            int iy = iy0[i] + kh, ix = ix0[i] + kw;
            bool ok = tap_ok && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const bf16_t *p = ok ? in + ((size_t)(pixbase[i] + iy * a.W + ix) * a.in_stride + kc)
                                 : (const bf16_t *)a.zeros;
            __builtin_amdgcn_global_load_lds((glb_void *)p, (lds_void *)(dx + i * RPP * 128), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < LB; ++i)
            __builtin_amdgcn_global_load_lds((glb_void *)(wrow[i] + (size_t)kt * 64), (lds_void *)(dw + i * RPP * 128), 16, 0, 0);
        kc += 64;
        while (kc >= a.Cin_pad) { kc -= a.Cin_pad; ++tap; }
    };

    f32x4 acc[TC][TP];
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int KT = a.Kpad / 64;
    stage(0, 0);
    __syncthreads();

    const int l15 = lane & 15, lq = lane >> 4;
    for (int kt = 0; kt < KT; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < KT) stage(kt + 1, buf ^ 1);
        const char *bx = sX + buf * (BP * 128) + (wpi * TP * 16 + l15) * 128;
        const char *bw = sW + buf * (BC * 128) + (wci * TC * 16 + l15) * 128;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int sw = ((kk * 4 + lq) ^ (l15 & 7)) << 4;
            bf16x8 fw[TC], fx[TP];
#pragma unroll
            for (int i = 0; i < TC; ++i) fw[i] = *(const bf16x8 *)(bw + i * 16 * 128 + sw);
#pragma unroll
            for (int j = 0; j < TP; ++j) fx[j] = *(const bf16x8 *)(bx + j * 16 * 128 + sw);
#pragma unroll
            for (int i = 0; i < TC; ++i)
#pragma unroll
                for (int j = 0; j < TP; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[i], fx[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();      // drains this step's global_load_lds (vmcnt(0)) and orders LDS reuse
    }

    // ---- epilogue: bias + activation (+ residual), 4 consecutive channels per lane ----
    const bf16_t *__restrict__ res = (const bf16_t *)a.res;
#pragma unroll
    for (int i = 0; i < TC; ++i) {
        const int ch = ct * BC + (wci * TC + i) * 16 + lq * 4;
        if (ch >= a.Cout) continue;
        const float4 bv = *(const float4 *)(a.bias + ch);
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            const int m = pt * BP + (wpi * TP + j) * 16 + l15;
            if (m >= M) continue;
            float v[4] = {acc[i][j][0] + bv.x, acc[i][j][1] + bv.y, acc[i][j][2] + bv.z, acc[i][j][3] + bv.w};
            if (a.act == ACT_LEAKY) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = v[q] > 0.f ? v[q] : 0.1f * v[q];
            }
            if (OUT_F32) {
                float *o = (float *)a.out + (size_t)m * a.out_stride + ch;
                if (ch + 3 < a.Cout) *(float4 *)o = float4{v[0], v[1], v[2], v[3]};
                else
                    for (int q = 0; q < 4; ++q) if (ch + q < a.Cout) o[q] = v[q];
            } else {
                if (res) {
                    const uint2 rv = *(const uint2 *)(res + (size_t)m * a.res_stride + ch);
                    // the layer's own output is rounded to bf16 first, exactly as if it had been stored
                    // and re-read by a separate shortcut kernel, then the add is rounded once more
                    v[0] = bf16_bits_to_f32(f32_to_bf16_rn(v[0])) + bf16_bits_to_f32(rv.x & 0xffff);
                    v[1] = bf16_bits_to_f32(f32_to_bf16_rn(v[1])) + bf16_bits_to_f32(rv.x >> 16);
                    v[2] = bf16_bits_to_f32(f32_to_bf16_rn(v[2])) + bf16_bits_to_f32(rv.y & 0xffff);
                    v[3] = bf16_bits_to_f32(f32_to_bf16_rn(v[3])) + bf16_bits_to_f32(rv.y >> 16);
                }
                uint2 pk;
                pk.x = f32_to_bf16_rn(v[0]) | (f32_to_bf16_rn(v[1]) << 16);
                pk.y = f32_to_bf16_rn(v[2]) | (f32_to_bf16_rn(v[3]) << 16);
                *(uint2 *)((bf16_t *)a.out + (size_t)m * a.out_stride + ch) = pk;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
struct CfgDesc { const char *name; int wp, wc, tp, tc; };
static const CfgDesc kCfgs[] = {
    {"p128c128_w2x2", 2, 2, 4, 4},
    {"p64c128_w2x2", 2, 2, 2, 4},
    {"p256c32_w4x1", 4, 1, 4, 2},
    {"p128c64_w2x2", 2, 2, 4, 2},
    {"p256c64_w4x1", 4, 1, 4, 4},
    {"p256c128_w4x2", 4, 2, 4, 4},
    {"p128c256_w2x4", 2, 4, 4, 4},
    {"p64c64_w2x2", 2, 2, 2, 2},
    {"p256c256_w2x4", 2, 4, 8, 4},
    {"p256c256_w4x2", 4, 2, 4, 8},
};
int conv_num_cfgs() { return (int)(sizeof(kCfgs) / sizeof(kCfgs[0])); }
const char *conv_cfg_name(int cfg) { return (cfg >= 0 && cfg < conv_num_cfgs()) ? kCfgs[cfg].name : "?"; }

int conv_pick_cfg(const ConvArgs &a)
{
    const long M = (long)a.N * a.Ho * a.Wo;
    if (a.Cout <= 32) return 2;
    if (a.Cout <= 64) return M >= 65536 ? 4 : 3;
    const long tiles128 = ((M + 127) / 128) * ((a.Cout + 127) / 128);
    if (tiles128 < 512) return M < 8192 && tiles128 < 128 ? 7 : 1;
    return 0;
}

template <int WP, int WC, int TP, int TC>
static hipError_t launch_t(const ConvArgs &a, hipStream_t s)
{
    constexpr int BP = WP * TP * 16, BC = WC * TC * 16;
    const long M = (long)a.N * a.Ho * a.Wo;
    const long tiles = ((M + BP - 1) / BP) * ((a.Cout + BC - 1) / BC);
    const size_t lds = 2 * (size_t)(BP + BC) * 128;
    dim3 grid((unsigned)((tiles + 7) / 8 * 8)), block(64 * WP * WC);   // multiple of 8: see the XCD mapping
    if (a.out_f32) {
        auto k = conv_igemm_bf16<WP, WC, TP, TC, true>;
        if (lds > 65536) {
            hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(k, grid, block, lds, s, a);
    } else {
        auto k = conv_igemm_bf16<WP, WC, TP, TC, false>;
        if (lds > 65536) {
            hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(k, grid, block, lds, s, a);
    }
    return hipGetLastError();
}

hipError_t launch_conv_bf16(const ConvArgs &a, int cfg, hipStream_t s)
{
    switch (cfg) {
    case 0: return launch_t<2, 2, 4, 4>(a, s);
    case 1: return launch_t<2, 2, 2, 4>(a, s);
    case 2: return launch_t<4, 1, 4, 2>(a, s);
    case 3: return launch_t<2, 2, 4, 2>(a, s);
    case 4: return launch_t<4, 1, 4, 4>(a, s);
    case 5: return launch_t<4, 2, 4, 4>(a, s);
    case 6: return launch_t<2, 4, 4, 4>(a, s);
    case 7: return launch_t<2, 2, 2, 2>(a, s);
    case 8: return launch_t<2, 4, 8, 4>(a, s);
    case 9: return launch_t<4, 2, 4, 8>(a, s);
    default: return hipErrorInvalidValue;
    }
}
