// Exact-fp32 variant of the fused implicit-GEMM convolution (BASELINE config 2: "YOLOv2 416x416 batch=1
// fp32").  Same structure as conv_igemm.hip -- NHWC gather, [rows][128 B] XOR-swizzled LDS tiles,
// double buffering, bias/leaky/residual epilogue -- but the contraction runs on the f32-input MFMA
// v_mfma_f32_16x16x4_f32, which is bit-for-bit a k-ordered fmaf chain (no reduced-precision path exists
// on gfx950), so the result differs from the fp32 reference only by summation order.
// A K-step is 32 floats (128 B per tile row); lane l feeds A[row l&15][k = l>>4], B[k = l>>4][col l&15].
#include "kernels.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int WP, int WC, int TP, int TC>
__global__ __launch_bounds__(64 * WP * WC) void conv_igemm_f32(const ConvArgs a)
{
    constexpr int NT = 64 * WP * WC;
    constexpr int BP = WP * TP * 16, BC = WC * TC * 16;
    constexpr int RPP = NT / 8;
    constexpr int LA = BP / RPP, LB = BC / RPP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *sX = smem, *sW = smem + 2 * BP * 128;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wpi = wid % WP, wci = wid / WP;
    const int M = a.N * a.Ho * a.Wo;
    const int tilesC = (a.Cout + BC - 1) / BC;
    const int ct = blockIdx.x % tilesC, pt = blockIdx.x / tilesC;
    const float *__restrict__ in = (const float *)a.in;
    const float *__restrict__ wt = (const float *)a.wt;

    const int chunk = tid & 7, r0 = tid >> 3;
    int pixbase[LA], iy0[LA], ix0[LA];
    const int HoWo = a.Ho * a.Wo;
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        int m = pt * BP + r0 + i * RPP;
        if (m < M) {
            int n = m / HoWo, rem = m - n * HoWo, oy = rem / a.Wo, ox = rem - oy * a.Wo;
            pixbase[i] = n * a.H * a.W; iy0[i] = oy * a.stride - a.pad; ix0[i] = ox * a.stride - a.pad;
        } else { pixbase[i] = 0; iy0[i] = -(1 << 20); ix0[i] = 0; }
    }
    const float *wrow[LB];
#pragma unroll
    for (int i = 0; i < LB; ++i) wrow[i] = wt + (size_t)(ct * BC + r0 + i * RPP) * a.Kpad + chunk * 4;

    const int KK = a.ksize * a.ksize;
    int kc = chunk * 4, tap = 0;
    while (kc >= a.Cin_pad) { kc -= a.Cin_pad; ++tap; }
    uint4 ra[LA], rb[LB];
    auto load_global = [&](int kt) {
        int kh = 0, kw = 0;
        if (a.ksize == 3) { kh = (tap * 11) >> 5; kw = tap - kh * 3; }
        else if (a.ksize != 1) { kh = tap / a.ksize; kw = tap - kh * a.ksize; }
        const bool tap_ok = tap < KK;
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            int iy = iy0[i] + kh, ix = ix0[i] + kw;
            bool ok = tap_ok && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const float *p = ok ? in + ((size_t)(pixbase[i] + iy * a.W + ix) * a.in_stride + kc) : (const float *)a.zeros;
            ra[i] = *(const uint4 *)p;
        }
#pragma unroll
        for (int i = 0; i < LB; ++i) rb[i] = *(const uint4 *)(wrow[i] + (size_t)kt * 32);
        kc += 32;
        while (kc >= a.Cin_pad) { kc -= a.Cin_pad; ++tap; }
    };
    auto store_lds = [&](int buf) {
#pragma unroll
        for (int i = 0; i < LA; ++i) { int r = r0 + i * RPP; *(uint4 *)(sX + buf * (BP * 128) + r * 128 + ((chunk ^ (r & 7)) << 4)) = ra[i]; }
#pragma unroll
        for (int i = 0; i < LB; ++i) { int r = r0 + i * RPP; *(uint4 *)(sW + buf * (BC * 128) + r * 128 + ((chunk ^ (r & 7)) << 4)) = rb[i]; }
    };

    f32x4 acc[TC][TP];
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int KT = a.Kpad / 32;
    load_global(0); store_lds(0); __syncthreads();
    const int l15 = lane & 15, lq = lane >> 4;
    for (int kt = 0; kt < KT; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < KT) load_global(kt + 1);
        const char *bx = sX + buf * (BP * 128) + (wpi * TP * 16 + l15) * 128 + lq * 4;
        const char *bw = sW + buf * (BC * 128) + (wci * TC * 16 + l15) * 128 + lq * 4;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const int sw = (kk ^ (l15 & 7)) << 4;
            float fw[TC], fx[TP];
#pragma unroll
            for (int i = 0; i < TC; ++i) fw[i] = *(const float *)(bw + i * 16 * 128 + sw);
#pragma unroll
            for (int j = 0; j < TP; ++j) fx[j] = *(const float *)(bx + j * 16 * 128 + sw);
#pragma unroll
            for (int i = 0; i < TC; ++i)
#pragma unroll
                for (int j = 0; j < TP; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fw[i], fx[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < KT) store_lds(buf ^ 1);
        __syncthreads();
    }

    const float *__restrict__ res = (const float *)a.res;
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
            if (a.act == ACT_LEAKY)
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = v[q] > 0.f ? v[q] : 0.1f * v[q];
            if (res) {
                const float4 rv = *(const float4 *)(res + (size_t)m * a.res_stride + ch);
                v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
            }
            float *o = (float *)a.out + (size_t)m * a.out_stride + ch;
            if (ch + 3 < a.Cout) *(float4 *)o = float4{v[0], v[1], v[2], v[3]};
            else for (int q = 0; q < 4; ++q) if (ch + q < a.Cout) o[q] = v[q];
        }
    }
}

template <int WP, int WC, int TP, int TC>
static hipError_t launch_f(const ConvArgs &a, hipStream_t s)
{
    constexpr int BP = WP * TP * 16, BC = WC * TC * 16;
    const long M = (long)a.N * a.Ho * a.Wo;
    const long tiles = ((M + BP - 1) / BP) * ((a.Cout + BC - 1) / BC);
    const size_t lds = 2 * (size_t)(BP + BC) * 128;
    hipLaunchKernelGGL((conv_igemm_f32<WP, WC, TP, TC>), dim3((unsigned)tiles), dim3(64 * WP * WC), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_conv_f32(const ConvArgs &a, hipStream_t s)
{
    const long M = (long)a.N * a.Ho * a.Wo;
    if (a.Cout <= 32) return launch_f<4, 1, 4, 2>(a, s);
    if (a.Cout <= 64) return launch_f<2, 2, 4, 2>(a, s);
    const long tiles128 = ((M + 127) / 128) * ((a.Cout + 127) / 128);
    if (tiles128 < 512) return launch_f<2, 2, 2, 4>(a, s);
    return launch_f<2, 2, 4, 4>(a, s);
}
