// Pair-K-loop instantiations of the implicit-GEMM conv kernel (conv_igemm_kernel.h, PAIRK = true; round 6): convs of a split-fp16
// (YOLO_FP16X2) network whose INPUT is an interleaved pair tensor -- per 32-channel group 64 bytes of hi, 64 bytes of lo -- against filter rows
// packed W_hi 32 | W_lo 32.  One K-step row pair feeds three MFMA products (W_hi x_hi + W_lo x_hi + W_hi x_lo) where the plain fp16 loop
// forms two, so the 3 x matrix work of the configuration costs 2 x (not 3 x) the staging, LDS traffic, barriers and waits.
// Replaces the same reference chain as conv_igemm.hip (DN/convolutional_layer.c:445-485; slim.conv2d V3/yolo_v3.py:47-60) at the fp32
// reference's accuracy (22 significant bits per operand).  Its own translation unit so that it compiles beside the other tile tables.
#include "conv_igemm_kernel.h"

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
        if (cfg == 40 || cfg == 41 || cfg == 43) return true;      // the free-running halo forms (conv_halo13.hip)
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
