// Halo-staged instantiations of the implicit-GEMM conv kernel (conv_igemm_kernel.h, HALO = true): 3x3 / stride 1 / pad 1 layers
// whose spatial size is a multiple of 13 -- every residual-block 3x3 conv of a 416x416 darknet-53 (104, 52, 26, 13).
// Replaces the same reference chain as conv_igemm.hip (DN/convolutional_layer.c:445-485; slim.conv2d V3/yolo_v3.py:47-60).
#include "conv_igemm_kernel.h"

bool conv_halo13_ok(const ConvArgs &a)
{
    const int row = a.in_dt == DT_FP8 ? 128 : 64;                  // channels of one 128-byte chunk
    if (a.in_dt != DT_BF16 && a.in_dt != DT_FP8 && a.in_dt != DT_F16) return false;
    if (a.ksize != 3 || a.stride != 1 || a.pad != 1 || a.Ho != a.H || a.Wo != a.W) return false;
    // whole 13 x 13 blocks (416 x 416 networks), or ragged ones on the bottom / right edge where they waste little (608 x 608: 38 = 3 * 13 - 1,
    // 76 = 6 * 13 - 2, 152 = 12 * 13 - 4: 5 % of the columns; 19 x 19 would compute 26 x 26: refused)
    const long cover = (long)((a.H + HALO_B - 1) / HALO_B) * ((a.W + HALO_B - 1) / HALO_B) * HALO_B * HALO_B;
    if (cover * 100 > (long)a.H * a.W * 115) return false;
    if (a.Cin_pad % row || a.kchunk != row || a.Kpad != 9 * a.Cin_pad) return false;
    if (a.out_dt == DT_F32) return false;
    // 32-bit buffer offsets below the out-of-range sentinel
    return (double)a.N * a.H * a.W * a.in_stride * dt_size(a.in_dt) < 2147483648.0;
}

template <int WC, int TC, int NL, int EB, bool FREE = false, int NS = 2, bool H16 = false, bool SPLIT = false>
static hipError_t launch_h(const ConvArgs &a, hipStream_t s)
{
    constexpr int WP = 1, TP = 11, BK = 64, BC = WC * TC * 16;
    const long blocks = (long)a.N * ((a.H + HALO_B - 1) / HALO_B) * ((a.W + HALO_B - 1) / HALO_B);
    const long tiles = blocks * ((a.Cout + BC - 1) / BC);
    constexpr size_t lds = conv_lds_bytes<WP, WC, TP, TC, NS, BK, NL, true>();
    static_assert(lds <= 160 * 1024, "halo form: LDS");
    hipError_t e = conv_opt_in_lds((const void *)conv_igemm<WP, WC, TP, TC, NS, BK, true, NL, false, EB, true, FREE, H16, SPLIT>, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((conv_igemm<WP, WC, TP, TC, NS, BK, true, NL, false, EB, true, FREE, H16, SPLIT>), dim3((unsigned)((tiles + 7) / 8 * 8)), dim3(64 * (WP * WC + NL)), lds, s, conv_tile_magic(a, BC, HALO_B));
    return hipGetLastError();
}

// stamped diagnostic builds of the free-running 176x256 form (tools only): per-wave cycle sums of the issue / wait / MFMA phases.
// variant 1 (wide): four waves of 176 x 64 (one per SIMD) instead of eight of 176 x 32 -- what a K-step costs a wave that shares its SIMD with
// nobody and reads 40 % fewer LDS bytes per FLOP
template <int WC, int TC>
static hipError_t launch_diag(const ConvArgs &a, hipStream_t s)
{
    constexpr int BC = WC * TC * 16;
    const long tiles = (long)a.N * ((a.H + HALO_B - 1) / HALO_B) * ((a.W + HALO_B - 1) / HALO_B) * ((a.Cout + BC - 1) / BC);
    constexpr size_t lds = conv_lds_bytes<1, WC, 11, TC, 2, 64, 0, true>();
    hipError_t e = conv_opt_in_lds((const void *)conv_igemm<1, WC, 11, TC, 2, 64, true, 0, true, 2, true, true>, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((conv_igemm<1, WC, 11, TC, 2, 64, true, 0, true, 2, true, true>), dim3((unsigned)((tiles + 7) / 8 * 8)), dim3(64 * WC), lds, s, conv_tile_magic(a, BC, HALO_B));
    return hipGetLastError();
}
hipError_t launch_conv_halo13_diag(const ConvArgs &a, hipStream_t s, int variant)
{
    if (!conv_halo13_ok(a) || a.in_dt != DT_BF16) return hipErrorInvalidValue;
    return variant == 1 ? launch_diag<4, 4>(a, s) : launch_diag<8, 2>(a, s);
}

hipError_t launch_conv_halo13(const ConvArgs &a, int cfg, hipStream_t s)
{
    if (!conv_halo13_ok(a)) return hipErrorInvalidValue;
    const bool f8 = a.in_dt == DT_FP8;
    if (a.split) {        // split fp16 storage (YOLO_FP16X2): the free-running forms with the two-pass epilogue
        if (a.in_dt != DT_F16 || a.out_dt != DT_F16 || a.w2) return hipErrorInvalidValue;
        switch (cfg) {
        case 40: return launch_h<8, 2, 0, 2, true, 2, true, true>(a, s);
        case 41: return launch_h<8, 1, 0, 2, true, 2, true, true>(a, s);
        case 43: return launch_h<8, 1, 0, 2, true, 3, true, true>(a, s);
        default: return hipErrorInvalidValue;
        }
    }
    if (a.in_dt == DT_F16) {
        if (a.out_dt != DT_F16) return hipErrorInvalidValue;
        switch (cfg) {
        case 36: return launch_h<8, 2, 0, 2, false, 2, true>(a, s);
        case 37: return launch_h<8, 2, 4, 2, false, 2, true>(a, s);
        case 38: return launch_h<4, 2, 4, 2, false, 2, true>(a, s);
        case 39: return launch_h<4, 2, 0, 2, false, 2, true>(a, s);
        case 40: return launch_h<8, 2, 0, 2, true, 2, true>(a, s);
        case 41: return launch_h<8, 1, 0, 2, true, 2, true>(a, s);
        case 42: return launch_h<4, 2, 0, 2, true, 2, true>(a, s);
        case 43: return launch_h<8, 1, 0, 2, true, 3, true>(a, s);
        default: return hipErrorInvalidValue;
        }
    }
    switch (cfg) {
    case 36: return f8 ? launch_h<8, 2, 0, 1>(a, s) : launch_h<8, 2, 0, 2>(a, s);
    case 37: return f8 ? launch_h<8, 2, 4, 1>(a, s) : launch_h<8, 2, 4, 2>(a, s);
    case 38: return f8 ? launch_h<4, 2, 4, 1>(a, s) : launch_h<4, 2, 4, 2>(a, s);
    case 39: return f8 ? launch_h<4, 2, 0, 1>(a, s) : launch_h<4, 2, 0, 2>(a, s);
    case 40: return f8 ? launch_h<8, 2, 0, 1, true>(a, s) : launch_h<8, 2, 0, 2, true>(a, s);
    case 41: return f8 ? launch_h<8, 1, 0, 1, true>(a, s) : launch_h<8, 1, 0, 2, true>(a, s);
    case 42: return f8 ? launch_h<4, 2, 0, 1, true>(a, s) : launch_h<4, 2, 0, 2, true>(a, s);
    case 43: return f8 ? launch_h<8, 1, 0, 1, true, 3>(a, s) : launch_h<8, 1, 0, 2, true, 3>(a, s);
    default: return hipErrorInvalidValue;
    }
}
