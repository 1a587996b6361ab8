// Halo-staged instantiations of the implicit-GEMM conv kernel (conv_igemm_kernel.h, HALO = true): 3x3 / stride 1 / pad 1 layers
// whose spatial size is a multiple of 13 -- every residual-block 3x3 conv of a 416x416 darknet-53 (104, 52, 26, 13) --, or (round 5) tiles
// into 10 x 19 blocks / 5 x 19 strips: the 76 / 38 / 19 grids of the 608x608 network (tile configurations 54..56).
// Replaces the same reference chain as conv_igemm.hip (DN/convolutional_layer.c:445-485; slim.conv2d V3/yolo_v3.py:47-60).
#include "conv_igemm_kernel.h"

static bool halo_ok(const ConvArgs &a, int bh, int bw);
bool conv_halo13_ok(const ConvArgs &a) { return halo_ok(a, HALO_B, HALO_B); }
// block shape of a halo configuration (ids: conv_igemm.hip's tables)
static void halo_cfg_block(int cfg, int &bh, int &bw) { bh = bw = HALO_B; if (cfg == 54 || cfg == 55) { bh = 10; bw = 19; } else if (cfg == 56) { bh = 5; bw = 19; } }
bool conv_halo_cfg_ok(const ConvArgs &a, int cfg)
{
    int bh, bw; halo_cfg_block(cfg, bh, bw);
    if (bh != HALO_B && (a.in_dt == DT_FP8 || a.split)) return false;      // the rectangular blocks are instantiated for bf16 / fp16 storage
    if (a.pairk && !(a.split && (cfg == 40 || cfg == 41 || cfg == 43 || cfg == 57 || cfg == 58))) return false;      // pair K loop: the free-running forms writing pairs
    if ((cfg == 57 || cfg == 58) && !a.pairk) return false;                    // (one wave per SIMD: instantiated for the pair K loop only)
    return halo_ok(a, bh, bw);
}
static bool halo_ok(const ConvArgs &a, int bh, int bw)
{
    const int row = a.in_dt == DT_FP8 ? 128 : 64;                  // channels of one 128-byte chunk
    if (a.in_dt != DT_BF16 && a.in_dt != DT_FP8 && a.in_dt != DT_F16) return false;
    if (a.ksize != 3 || a.stride != 1 || a.pad != 1 || a.Ho != a.H || a.Wo != a.W) return false;
    // whole 13 x 13 blocks (416 x 416 networks), or ragged ones on the bottom / right edge where they waste little (608 x 608: 38 = 3 * 13 - 1,
    // 76 = 6 * 13 - 2, 152 = 12 * 13 - 4: 5 % of the columns; 19 x 19 would compute 26 x 26: refused)
    // (round 5: 10 x 19 and 5 x 19 blocks -- 76 = 8 x 10 - 4 rows, 4 x 19 columns; 38 = 4 x 10 - 2; 19 = 4 x 5 - 1: at most 5 % of the rows)
    const long cover = (long)((a.H + bh - 1) / bh) * ((a.W + bw - 1) / bw) * bh * bw;
    if (cover * 100 > (long)a.H * a.W * 115) return false;
    if (a.Cin_pad % row || a.kchunk != row || a.Kpad != 9 * a.Cin_pad) return false;
    if (a.out_dt == DT_F32) return false;
    // 32-bit buffer offsets below the out-of-range sentinel
    return (double)a.N * a.H * a.W * a.in_stride * dt_size(a.in_dt) < 2147483648.0;
}

template <int WC, int TC, int NL, int EB, bool FREE = false, int NS = 2, bool H16 = false, bool SPLIT = false, int BH = HALO_B, int BW = HALO_B, bool HEADT = false, bool PAIRK = false>
static hipError_t launch_h(const ConvArgs &a, hipStream_t s)
{
    if (a.tail_f32 && !HEADT) return hipErrorInvalidValue;        // a head as the tail runs on the HEADT instantiations (tile configurations 40 and 54)
    constexpr int WP = 1, TP = (BH * BW + 15) / 16, BK = 64, BC = WC * TC * 16;
    const long blocks = (long)a.N * ((a.H + BH - 1) / BH) * ((a.W + BW - 1) / BW);
    const long tiles = blocks * ((a.Cout + BC - 1) / BC);
    constexpr size_t lds = conv_lds_bytes<WP, WC, TP, TC, NS, BK, NL, true, BH, BW>();
    static_assert(lds <= 160 * 1024, "halo form: LDS");
    hipError_t e = conv_opt_in_lds((const void *)conv_igemm<WP, WC, TP, TC, NS, BK, true, NL, false, EB, true, FREE, H16, SPLIT, BH, BW, HEADT, PAIRK>, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((conv_igemm<WP, WC, TP, TC, NS, BK, true, NL, false, EB, true, FREE, H16, SPLIT, BH, BW, HEADT, PAIRK>), dim3((unsigned)((tiles + 7) / 8 * 8)), dim3(64 * (WP * WC + NL)), lds, s, conv_tile_magic(a, BC, BH, BW));
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
    if (!conv_halo_cfg_ok(a, cfg)) return hipErrorInvalidValue;
    const bool f8 = a.in_dt == DT_FP8;
    // round 5: free-running forms on 10 x 19 blocks (12 sub-tiles: 76 x 76 and 38 x 38 grids) and 5 x 19 strips (6 sub-tiles: 19 x 19), bf16 / fp16
    if (cfg >= 54 && cfg <= 56) {
        if (f8 || a.split) return hipErrorInvalidValue;
        const bool h = a.in_dt == DT_F16;
        if (h && a.out_dt != DT_F16) return hipErrorInvalidValue;
        switch (cfg) {
        case 54: if (a.tail_f32) return h ? launch_h<8, 2, 0, 2, true, 2, true, false, 10, 19, true>(a, s) : launch_h<8, 2, 0, 2, true, 2, false, false, 10, 19, true>(a, s);
                 return h ? launch_h<8, 2, 0, 2, true, 2, true, false, 10, 19>(a, s) : launch_h<8, 2, 0, 2, true, 2, false, false, 10, 19>(a, s);
        case 55: return h ? launch_h<8, 1, 0, 2, true, 3, true, false, 10, 19>(a, s) : launch_h<8, 1, 0, 2, true, 3, false, false, 10, 19>(a, s);
        default: return h ? launch_h<8, 1, 0, 2, true, 3, true, false, 5, 19>(a, s) : launch_h<8, 1, 0, 2, true, 3, false, false, 5, 19>(a, s);
        }
    }
    if (a.split) {        // split fp16 storage (YOLO_FP16X2): the free-running forms with the two-pass epilogue
        if (a.in_dt != DT_F16 || a.out_dt != DT_F16 || a.w2) return hipErrorInvalidValue;
        if (a.pairk)          // pairs in, pairs out: the pair K loop (three products per K-step row pair)
            switch (cfg) {
            case 40: return launch_h<8, 2, 0, 2, true, 2, true, true, HALO_B, HALO_B, false, true>(a, s);
            case 41: return launch_h<8, 1, 0, 2, true, 2, true, true, HALO_B, HALO_B, false, true>(a, s);
            case 43: return launch_h<8, 1, 0, 2, true, 3, true, true, HALO_B, HALO_B, false, true>(a, s);
            case 57: return launch_h<4, 4, 0, 2, true, 2, true, true, HALO_B, HALO_B, false, true>(a, s);
            case 58: return launch_h<4, 2, 0, 2, true, 2, true, true, HALO_B, HALO_B, false, true>(a, s);
            default: return hipErrorInvalidValue;
            }
        switch (cfg) {          // plain fp16 in, pairs out (mixed plans)
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
        case 40: return a.tail_f32 ? launch_h<8, 2, 0, 2, true, 2, true, false, HALO_B, HALO_B, true>(a, s) : launch_h<8, 2, 0, 2, true, 2, true>(a, s);
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
    case 40: return f8 ? launch_h<8, 2, 0, 1, true>(a, s) : a.tail_f32 ? launch_h<8, 2, 0, 2, true, 2, false, false, HALO_B, HALO_B, true>(a, s) : launch_h<8, 2, 0, 2, true>(a, s);
    case 41: return f8 ? launch_h<8, 1, 0, 1, true>(a, s) : launch_h<8, 1, 0, 2, true>(a, s);
    case 42: return f8 ? launch_h<4, 2, 0, 1, true>(a, s) : launch_h<4, 2, 0, 2, true>(a, s);
    case 43: return f8 ? launch_h<8, 1, 0, 1, true, 3>(a, s) : launch_h<8, 1, 0, 2, true, 3>(a, s);
    default: return hipErrorInvalidValue;
    }
}
