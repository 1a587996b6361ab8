// The implicit-GEMM conv kernel template (see conv_igemm.hip for the description); included by conv_igemm.hip (tiled
// configurations) and conv_halo13.hip (halo-staged 3x3 configurations) so the two sets of instantiations compile in parallel.
#pragma once
#include "kernels.h"
#include "halo_perm_tables.h"
#include <type_traits>
#include <utility>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ uint32_t f32_to_bf16_rn(float f)
{
    __bf16 b = (__bf16)f;                       // v_cvt_pk_bf16_f32: RNE, NaN preserved
    return (uint32_t)__builtin_bit_cast(uint16_t, b);
}
__device__ __forceinline__ float bf16_bits_to_f32(uint32_t b) { return __builtin_bit_cast(float, b << 16); }
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
// two floats -> packed bf16 pair (lo | hi << 16) in ONE v_cvt_pk_bf16_f32 (same RNE as f32_to_bf16_rn)
__device__ __forceinline__ uint32_t f32x2_to_bf16x2(float lo, float hi)
{
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t));
}
// ---- 16-bit storage type of the EB = 2 kernels: bf16 (8-bit significand) or, H16 = true, IEEE fp16 (11-bit significand; same MFMA
//      rate, v_mfma_f32_16x16x32_f16).  Values beyond fp16's range saturate at +-65504 on the way to memory. ----
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
#define F16_MAX 65504.0f
// The saturation is the hardware's: the fp16 kernels set MODE.FP16_OVFL (fp16_saturating_mode below), under which a conversion that
// overflows yields +-65504 instead of an infinity -- two v_med3_f32 per stored pair less than clamping in fp32 first (that clamp was the
// 3-4 % fp16 cost against bf16).
__device__ __forceinline__ void fp16_saturating_mode() { __builtin_amdgcn_s_setreg((0 << 11) | (23 << 6) | 1, 1); }      // hwreg(HW_REG_MODE, 23, 1) = FP16_OVFL
template <bool H16> __device__ __forceinline__ uint32_t pack16x2(float lo, float hi)
{
    if constexpr (H16) return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{lo, hi}, f16x2_t));       // RNE, saturating (FP16_OVFL)
    else return f32x2_to_bf16x2(lo, hi);
}
template <bool H16> __device__ __forceinline__ float unpack16_lo(uint32_t w)
{
    if constexpr (H16) return (float)__builtin_bit_cast(f16x2_t, w)[0]; else return __builtin_bit_cast(float, w << 16);
}
template <bool H16> __device__ __forceinline__ float unpack16_hi(uint32_t w)
{
    if constexpr (H16) return (float)__builtin_bit_cast(f16x2_t, w)[1]; else return __builtin_bit_cast(float, w & 0xffff0000u);
}
template <bool H16> __device__ __forceinline__ f32x4 mma16(const bf16x8 a, const bf16x8 b, const f32x4 c)
{
    if constexpr (H16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// Split-fp16 pairs in the epilogues, on the mixed-precision FMA (round 6): `v_fma_mix_f32` reads an f16 half as one of its operands, so
// hi + lo of a pair is ONE instruction per value (two conversions and an add before), and `v_fma_mixlo/hi_f16` forms f16(v - hi) into a half
// of the destination in one (conversion, subtraction, conversion before).  Same values bit for bit: fp32(hi) * 1 + fp32(lo) and v - fp32(hi)
// are exact in fp32 for a pair made by the split, the one rounding is the final one to f16 (RNE, as v_cvt_pk_f16_f32).
__device__ __forceinline__ void pair_join2(uint32_t H, uint32_t L, float &a, float &b)
{
#ifdef PAIR_NO_MIX          // (probe builds: the conversions and separate add / subtract of rounds 4-5, for the same-box A/B)
    a = unpack16_lo<true>(H) + unpack16_lo<true>(L); b = unpack16_hi<true>(H) + unpack16_hi<true>(L); return;
#endif
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,1]" : "=v"(a) : "v"(H), "v"(L));
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(b) : "v"(H), "v"(L));
}
__device__ __forceinline__ void pair_split2(float a, float b, uint32_t &H, uint32_t &L)
{
    H = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{a, b}, f16x2_t));       // RNE, saturating (FP16_OVFL)
#ifdef PAIR_NO_MIX
    L = pack16x2<true>(a - unpack16_lo<true>(H), b - unpack16_hi<true>(H)); return;
#endif
    uint32_t l;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(H), "v"(a));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(H), "v"(b));
    L = l;
}
// max(a, b) for finite operands without the sNaN-quieting v_max the compiler puts in front of fmaxf (one instruction, not two)
__device__ __forceinline__ float vmax_f32(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// two floats -> two OCP e4m3 codes (RNE, saturating at +-448) merged into the low / high half of `old`
template <bool HI> __device__ __forceinline__ uint32_t f32x2_to_fp8(float a, float b, uint32_t old)
{
    a = __builtin_amdgcn_fmed3f(a, -FP8_MAX, FP8_MAX); b = __builtin_amdgcn_fmed3f(b, -FP8_MAX, FP8_MAX);
    return (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, (int)old, HI);
}

__device__ __forceinline__ int fast_div(int n, uint32_t mul, uint32_t shift)
{
    return shift == 255 ? n : (int)(__umulhi((uint32_t)n, mul) >> shift);
}

// probe knobs of the pair K loop (tools/probe/ab): fragment read-ahead in first uses, stagger of a SIMD's two waves in 1/64 MFMA-steps
#ifndef PAIR_PD
#define PAIR_PD 4
#endif
#ifndef PAIR_SLEEP
#define PAIR_SLEEP 48
#endif
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// counted wait with a wave-uniform run-time count (the immediate has to be a constant: a scalar branch ladder)
template <int MAXN> __device__ __forceinline__ void wait_vmcnt_rt(int n)
{
    static_assert(MAXN <= 16, "ladder length");
    switch (n) {
#define W_(k) case k: if (k <= MAXN) { wait_vmcnt<k>(); } break;
        W_(1) W_(2) W_(3) W_(4) W_(5) W_(6) W_(7) W_(8) W_(9) W_(10) W_(11) W_(12) W_(13) W_(14) W_(15) W_(16)
#undef W_
    default: wait_vmcnt<0>(); break;
    }
}
__device__ __forceinline__ void block_barrier() { asm volatile("s_barrier" ::: "memory"); }

#define OOB_OFFSET 0x80000000u                  // >= num_records of every descriptor below -> DMA writes zeros
#define BUF_RECORDS 0x80000000u

// UNI: Cin_pad is a multiple of 64, so all 8 chunks of a K-step belong to one tap (scalar tap cursor).
// otherwise (Cin_pad = 8, 16, 32, ...): the chunks of one K-step span several taps, tap cursor is per lane.
// NL > 0: role split -- WP*WC consumer waves only read LDS and issue MFMAs, NL extra loader waves only issue the LDS-DMA
// (an LDS-DMA instruction blocks the issuing wave for ~66 cycles; in the symmetric NL = 0 form that is time the wave's own
// MFMAs cannot be issued).  All waves meet at the one s_barrier per K-step.
// EB: bytes per input element -- 2: bf16 operands (v_mfma_f32_16x16x32_bf16), 1: OCP e4m3 operands
// (v_mfma_f32_16x16x128_f8f6f4, twice the bf16 rate).  The byte geometry of the LDS tiles is the same for both: a
// 128-B row is 64 bf16 or 128 fp8 of K, so the fp8 form walks K twice as fast with the same loads.
template <int... I, class F> __device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F &&f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> __device__ __forceinline__ void static_for(F &&f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

// HALO: halo-staged form for 3x3 / stride 1 / pad 1 convs whose spatial size is a multiple of 13 (every 3x3/s1 layer of a
// 416x416 darknet: 104, 52, 26, 13).  A workgroup computes one 13x13 block of output pixels (169 of its 176 MFMA columns) for BC
// channels.  Per 64-channel chunk (EB = 1: 128-channel chunk) the 15x15 input halo of the block is DMA'd into LDS ONCE, as
// 160-byte pixel records (128 B of channels + 32 B of padding) without swizzle, and serves all nine taps: a tap is a constant
// byte offset ((kh * 15 + kw) * 160) on the fragment read, so the K-loop body is unrolled over the taps.  Per K-step only the
// filter tile is fetched (BC x 128 B) -- L2 -> LDS traffic drops from (176 + BC) x 128 B to (BC + 25) x 128 B per K-step and
// the nine-fold re-read of the activations through the vector L1 disappears.  The K order is the packed filters' own
// (chunk-major: k = (chunk * 9 + tap) * 64 + c), i.e. exactly the order of the tiled form: results are bit-identical.
// Bank conflicts: with 160-byte records a ds_read_b128 is conflict-free iff the halo indices y * 15 + x of lanes {0-3, 12-15}
// and of lanes {4-11} of a 16-lane MFMA column group are distinct mod 8; kHaloPerm13 (tools/gen/halo_perm.py) assigns the
// block's pixels to MFMA columns accordingly (raster order would conflict two-way in every group).
// lane -> pixel assignment tables: generated, halo_perm_tables.h (13 x 13: worst conflict degree 2, one group; 10 x 19 and 5 x 19: none)
// Round 5: the block is BH rows x BW columns (template parameters, 13 x 13 by default) -- 10 x 19 and 5 x 19 blocks tile the 608 x 608
// network's 76 / 38 / 19 grids exactly and fill 256 CUs at 8 images per GPU (BASELINE config 4's share), where 13 x 13 blocks are ragged.
// The same table as a lane reads it: 32 bytes per l15 = two 16-byte loads issued with the kernel's first instructions (bytes 0..TP-1: the
// pixel whose window sub-tile j's lane reads; bytes 16..16+TP-1: the row of the output tile it writes).  (Eleven byte loads per table
// sat behind the prologue's LDS-DMA in the in-order vmcnt queue: their first use waited for the whole prologue to land.)
struct HaloPermPk { unsigned w[16][8]; };
template <int N> constexpr HaloPermPk make_halo_perm_pk(const unsigned short (&tab)[N])
{
    HaloPermPk t{};
    for (int l = 0; l < 16; ++l)
        for (int j = 0; j < N / 16; ++j) {
            const unsigned v = tab[j * 16 + l];
            t.w[l][j / 4] |= (v & 0xffu) << (8 * (j % 4));
            t.w[l][4 + j / 4] |= (v >> 8) << (8 * (j % 4));
        }
    return t;
}
__device__ const HaloPermPk kHaloPermPk = make_halo_perm_pk(kHaloPerm13);
__device__ const HaloPermPk kHaloPermPk10x19 = make_halo_perm_pk(kHaloPerm10x19);
__device__ const HaloPermPk kHaloPermPk5x19 = make_halo_perm_pk(kHaloPerm5x19);
constexpr int HALO_B = 13;                         // default block edge (output pixels)
constexpr int HALO_APIX = 160;                     // bytes of one halo pixel record in LDS
constexpr int halo_apieces(int bh, int bw) { return ((bh + 2) * (bw + 2) * (HALO_APIX / 16) + 63) / 64; }      // 1-KiB LDS-DMA pieces of one halo tile: 36 for 13 x 13
constexpr int HALO_APIECES = halo_apieces(HALO_B, HALO_B);
constexpr int HALO_ACT_BYTES = HALO_APIECES * 1024;
// row / bw for row < 256 as a multiply and a shift (the epilogue's piece loop), checked exhaustively at compile time
constexpr unsigned halo_div_mul(int bw) { return (65536u + bw - 1) / bw; }
constexpr bool halo_div_ok(int bw) { for (unsigned r = 0; r < 256; ++r) if (((r * halo_div_mul(bw)) >> 16) != r / bw) return false; return true; }

// FREE (halo form only): free-running waves.  Wave w fetches exactly the filter rows it multiplies itself (its TC * 16 channels:
// LB = TC * 2 pieces per K-step into a private slice of the two filter stages), so inside a channel chunk nothing but the wave's
// own counted vmcnt orders its K-steps -- the one workgroup barrier per K-step of the other forms (where all eight waves issue
// their LDS-DMA, then all wait, then all multiply, and the matrix pipe idles through the first two) shrinks to one barrier per
// CHUNK (nine K-steps), which the shared halo tile needs.  The two waves of a SIMD (w and w + 4) are started half a K-step apart
// after every barrier, so that one's DMA-issue / wait / fragment-read startup runs under the other's MFMAs.
// (Tried on the free-running form and dropped, same-box A/B with tools/probe/kslope.py, YOLOv3-416 batch-32 layer shapes: three and four
// filter stages -- K-step 0.57 vs 0.59 us on the 13x13 layers, nothing on the others, kept as cfg 43; fragment read-ahead of 2 / 6 /
// 8 groups instead of 4 -- +-1 %; a direct-store epilogue (v_permlane32_swap pairs, 16-B stores straight from the accumulators, no
// LDS staging and no barrier) -- bit-identical and not one microsecond faster: what a layer pays outside its K loop is instruction
// issue in the set-up and the epilogue (stamped builds), which that variant did not shorten.)
// (Tried and dropped: starting half of the workgroups 3-6 us late -- by XCD parity, by CU parity inside an XCD, in four phases -- so
// that the prologue / epilogue bursts of the two halves do not coincide.  The late half costs its full delay on every layer (26x26:
// +2.2 us per 3 us of delay) and the early half gains < 1 us: the fixed phases are latency chains per CU, not an aggregate HBM limit.)
// (Tried and dropped, round 2: software-pipelining the K-steps inside each wave -- next step's filter and first pixel fragments read
// under the last MFMA groups, the DMA two steps ahead.  The wave's step then has no separate wait phase, but it got SLOWER, 1.10 vs
// 1.03 us per K-step on the 26x26 layers: with both waves of a SIMD reading fragments all the time the LDS round trip grows to ~300
// cycles and the four-deep read-ahead, not the start-up of a step, sets the pace (deeper read-ahead spills past 256 VGPRs).  A stamped
// build with four waves of 176 x 64 -- one per SIMD, 40 % fewer LDS bytes per FLOP, nothing else on the SIMD -- needs 1708 cycles for
// the 1408 cycles of MFMA of a K-step: the LDS array (213 KB of fragment reads plus 36 KB of DMA writes per K-step and CU) is the
// resource this tiling runs out of, at about the rate measured now.  A barrier-per-K-step halo form with 2 x 4 waves of 96 x 64 --
// 27 % fewer LDS bytes per FLOP -- was bit-identical and ran the 26x26 K-step in 0.99 us, the same as the 176 x 32 forms: every
// variant lands on ~1.45 PFLOP/s, the rate the chip sustains on random bf16 operands once its clock management has reacted (the
// CDNA4 guide's 'DVFS give-back': a cycle saved in an MFMA-dense main loop comes back partly as a lower clock).)
// SPLIT (H16 only; YOLO_FP16X2): 16-bit outputs are stored as split fp16 pairs -- hi = f16(v), lo = f16(v - hi), interleaved per 32-channel
// group (64 bytes of hi, 64 bytes of lo) -- by an epilogue of its own that stages the tile in LDS as fp32, half its channels at a time, and
// folds the shortcut.  The K loop is PAIRK's when the input is such a tensor, the plain fp16 one when it is a plain tensor (mixed plans)
// or the network input (hi | lo | hi blocks of its 8 padded channels against filter rows W_hi | W_hi | W_lo: rounds 4-5's form, kept there).
// PAIRK (H16 only; round 6): the INPUT is a split-fp16 pair tensor in the interleaved layout -- per 32-channel group 64 bytes of hi followed by
// 64 bytes of lo, i.e. one 128-byte K-step row [hi 32 | lo 32] of a pixel -- and the filter rows are packed the same way [W_hi 32 | W_lo 32].
// Staging, LDS layout, swizzle and the halo tile are those of an fp16 conv over 2 x Cin 'channels'; what changes is the MFMA phase: per
// K-step and accumulator tile THREE products, W_hi x_hi + W_lo x_hi + W_hi x_lo, from the same four fragment reads that feed two in the
// plain loop.  (Rounds 4-5 stored hi | lo | hi against W_hi | W_hi | W_lo and ran the plain loop over 3 x Cin: 1.5 x the activation bytes,
// LDS traffic, barriers and waits of this form for the same matrix work.)
template <int WP, int WC, int TP, int TC, int NS, int BK, bool UNI, int NL = 0, bool DIAG = false, int EB = 2, bool HALO = false, bool FREE = false, bool H16 = false, bool SPLIT = false, int BH = HALO_B, int BW = HALO_B, bool HEADT = false, bool PAIRK = false>
__global__ __launch_bounds__(64 * (WP * WC + NL)) void conv_igemm(const ConvArgs a)
{
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass only needs the launch stub (hipcc drops the stub of a
                                      // kernel whose body uses the buffer-resource builtins with array operands)
    constexpr int NC = WP * WC;                // consumer (MFMA) waves
    constexpr int NW = NL > 0 ? NL : NC;       // waves that issue the LDS-DMA
    constexpr int NTOT = NC + NL;              // waves in the workgroup
    constexpr int BP = WP * TP * 16;           // output pixels per workgroup
    constexpr int BC = WC * TC * 16;           // output channels per workgroup
    static_assert(BK == 64 || BK == 32, "K-step");
    static_assert(BC <= 256, "filters and bias are padded to multiples of 256 output channels (cout_pad): a wider tile would read past them");
    static_assert(EB == 2 || (EB == 1 && BK == 64), "fp8 operands need the 128-B-row form");
    static_assert(!H16 || EB == 2, "fp16 is a 16-bit storage type");
    static_assert(!SPLIT || (H16 && !DIAG), "split pairs are fp16 pairs");
    static_assert(!PAIRK || (H16 && EB == 2 && BK == 64 && UNI && !DIAG && NL == 0), "pair K loop: fp16, 128-byte rows = one 32-channel group hi | lo, uniform taps");
    constexpr int RB = BK * 2;                 // bytes of one LDS tile row (one K-step of one pixel / filter)
    constexpr int EPC = 16 / EB;               // elements per 16-B chunk
    constexpr int BKE = RB / EB;               // K elements per step
    constexpr int CPRW = RB / 16;              // 16-B chunks per row: 8 or 4
    constexpr int RG = 64 / CPRW;              // rows filled by one wave-level LDS-DMA instruction: 8 or 16
    constexpr int GP = (BP + RG - 1) / RG, GC = (BC + RG - 1) / RG;
    // every wave issues the same number of LDS-DMA instructions per K-step (the counted vmcnt relies on it): group
    // counts are padded up to a multiple of the wave count; padded rows get an out-of-range offset (zeros, no traffic)
    // (Tried and dropped: with WP == 1 no filter row is shared between waves, so the filter fragments could go
    // global -> VGPR directly and skip LDS.  Measured SLOWER, K-step 2114 -> 2900 cycles for p176c128: a
    // fragment-shaped load touches 16 rows x 64 B per instruction, which the texture addresser handles far worse than
    // the 8 x 128-B rows of an LDS-DMA piece.)
    constexpr int LA = (GP + NW - 1) / NW, LB = (GC + NW - 1) / NW;
    constexpr int L = LA + LB;
    constexpr int BPL = LA * NW * RG, BCL = LB * NW * RG;    // rows of the LDS images
    constexpr int STAGE_BYTES = (BPL + BCL) * RB;
    static_assert(!HALO || (WP == 1 && TP == (BH * BW + 15) / 16 && TP <= 16 && BK == 64 && UNI && (NS == 2 || (FREE && NS <= 4))), "halo form: one BH x BW block per workgroup; more than two filter stages in the free-running form only");
    static_assert((BH == 13 && BW == 13) || (BH == 10 && BW == 19) || (BH == 5 && BW == 19), "block shapes with a generated permutation table");
    static_assert(halo_div_ok(BW) && BH * BW <= 256, "row / BW by multiply-high");
    constexpr int HBH = BH, HBW = BW, HPW = BW + 2, HPH = BH + 2, APIX = HALO_APIX, APIECES = halo_apieces(BH, BW), ACT_BYTES = APIECES * 1024;
    constexpr int LAH = HALO ? (APIECES + NW - 1) / NW : 1;    // halo pieces per loading wave (one per tap, taps 0..LAH-1)
    constexpr int FSTAGE = BCL * RB;                            // halo form: bytes of one filter stage
    static_assert(!HALO || LAH <= 9, "a halo tile is fetched during the nine taps of the previous chunk");
    static_assert(!FREE || (HALO && NL == 0 && LB * RG == TC * 16), "free-running form: every wave fetches its own filter rows");

    if constexpr (H16) fp16_saturating_mode();
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [NS][ BP rows | BC rows ][128 B]

    unsigned long long t_top = 0;              // DIAG: first instruction of the wave
    if (DIAG) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_top)::"memory");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    // halo form: this lane's slice of the pixel permutation, requested before anything else (it is independent of the arguments)
    u32x4_t perm_rd = {0, 0, 0, 0}, perm_wr = {0, 0, 0, 0};
    if constexpr (HALO) {
        const u32x4_t *pk = (const u32x4_t *)(BH == 10 ? kHaloPermPk10x19.w[lane & 15] : BH == 5 ? kHaloPermPk5x19.w[lane & 15] : kHaloPermPk.w[lane & 15]);
        perm_rd = pk[0]; perm_wr = pk[1];
    }
    // the arguments the way to the first LDS-DMA needs, as ONE burst of scalar loads (left alone the compiler fetches each field
    // right before its first use: a dozen dependent scalar-cache round trips in front of the prologue)
    asm volatile("" ::"s"(a.in), "s"(a.wt), "s"(a.in_stride), "s"(a.N), "s"(a.H), "s"(a.W), "s"(a.Cout), "s"(a.Kpad), "s"(a.Ho), "s"(a.Wo),
                 "s"(a.tc_mul), "s"(a.tc_shift), "s"(a.bpi_mul), "s"(a.bpi_shift), "s"(a.bpr_mul), "s"(a.bpr_shift));
    const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_loader = NL == 0 || wave_id >= NC;          // wave-uniform role
    // deliberately a run-time value even when NL == 0 (always true then): with the branch folded away the scheduler
    // overlaps more of the K-step and the 176x128 shape goes from 240 to 272 VGPRs, i.e. from two resident workgroups
    // per CU to one
    const bool is_consumer = wave_id < NC;
    const int wid = NL > 0 ? (wave_id >= NC ? wave_id - NC : 0) : wave_id;   // index among the loading waves
    const int wpi = wave_id % WP, wci = (wave_id / WP) % WC;

    // XCD-aware tile assignment: workgroups b, b+8, b+16.. share an XCD (and its L2); give each XCD a
    // contiguous run of tiles ordered pixel-tile-major so that its resident workgroups re-use the same
    // activation rows (all channel tiles of a pixel tile) and walk the filter slices together.
    const int M = a.N * a.Ho * a.Wo;
    const int tilesC = (a.Cout + BC - 1) / BC;
    // (halo form: a size that is not a multiple of 13 gets ragged blocks on its bottom / right edge -- their columns past the image read
    //  zeros from the halo tile, as padding does, and are never stored)
    const int hbr = HALO ? (a.H + HBH - 1) / HBH : 0, hbc = HALO ? (a.W + HBW - 1) / HBW : 0;      // block rows / columns per image
    const int tilesP = HALO ? a.N * hbr * hbc : (M + BP - 1) / BP;
    const int per_xcd = gridDim.x >> 3;
    const int tile = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (tile >= tilesP * tilesC) return;
    const int pt = fast_div(tile, a.tc_mul, a.tc_shift);          // (scalar n / d by multiply-high: an s_ division is ~35 instructions)
    const int ct = tile - pt * tilesC;
    int bn = 0, by = 0, bx = 0;                // halo form: image, block row, block column of this workgroup's 13x13 block
    if constexpr (HALO) {
        const int bpr = hbc, bpi = hbc * hbr;
        bn = fast_div(pt, a.bpi_mul, a.bpi_shift); const int r = pt - bn * bpi; by = fast_div(r, a.bpr_mul, a.bpr_shift); bx = r - by * bpr;
    }

    // Buffer descriptors.  The activation base is moved back by (W+1) pixels so that the offset of tap (0,0) of a
    // border pixel (one row up, one column left) is still >= 0.
    const int shift = HALO ? 0 : (a.W + 1) * a.in_stride * EB;      // bytes (the halo form only forms in-image offsets)
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)a.in - shift), 0, BUF_RECORDS, 0x00020000);
    __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)a.wt, 0, BUF_RECORDS, 0x00020000);

    // ---- per-lane constants: wave w fills row groups w, w+NW, ...; inside a group lane l fills LDS slot
    //      (row l>>3, physical chunk l&7), i.e. logical K-chunk (l&7) ^ (row&7) of that row ----
    // LDS-DMA is lane-linear: lane l fills row (l / CPRW) of its row group, physical chunk (l % CPRW).  The chunk swizzle
    // that makes the ds_read_b128 fragment reads conflict-free is therefore applied to the SOURCE chunk:
    //   BK = 64 (128-B rows): phys = chunk ^ (row & 7);   BK = 32 (64-B rows): phys = chunk ^ (3 * ((row >> 2) & 1))
    const int rl = lane / CPRW;                // row within the row group (group bases are multiples of RG)
    const int chunk = BK == 64 ? ((lane & 7) ^ (rl & 7)) : ((lane & 3) ^ (3 * ((rl >> 2) & 1)));
    const int KK = a.ksize * a.ksize;
    const int HoWo = a.Ho * a.Wo;
    unsigned woff[LB > 0 ? LB : 1];
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        const int crow = (FREE ? wid * LB + i : wid + i * NW) * RG + rl;
        woff[i] = crow < BC ? (unsigned)((ct * BC + crow) * a.Kpad + chunk * EPC) * (unsigned)EB : OOB_OFFSET;
    }
    // K-step 0's filter rows are requested first -- their offsets are the cheapest to form, and the pixel addressing below (halo form:
    // ~130 instructions; tiled form: ~55 per row group) then runs while they are in flight.  Order in the vmcnt queue: filters of step 0,
    // activations of step 0, later stages: the counted waits of the K loop only rely on whole K-steps retiring in order.
    if (is_loader && 0 < a.Kpad / BKE) {
        char *const f0 = smem + (HALO ? 2 * ACT_BYTES : BPL * RB);
#pragma unroll
        for (int i = 0; i < LB; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void *)(f0 + (FREE ? wid * LB + i : wid + i * NW) * 1024), 16, woff[i], 0, 0, 0);
    }
    unsigned rowoff[LA];                       // byte offset of tap (0,0), channel chunk*8 (UNI) or 0 (per-lane tap)
    unsigned tapmask[LA];                      // bit t set: tap t of this pixel lies inside the image
    // halo form: per-lane source offsets of this wave's halo pieces.  LDS slot g = piece * 64 + lane holds 16-B piece g % 10 of
    // halo pixel g / 10 (pieces 8, 9 of a record are padding: out-of-range source, the DMA writes zeros)
    unsigned aoff[LAH];
    if constexpr (HALO) {
#pragma unroll
        for (int i = 0; i < LAH; ++i) {
            const int g = (wid + i * NW) * 64 + lane;
            const int pix = g / (APIX / 16), c16 = g - pix * (APIX / 16);
            const int hy = pix / HPW, hx = pix - hy * HPW;
            const int iy = by * HBH + hy - 1, ix = bx * HBW + hx - 1;
            const bool ok = pix < HPH * HPW && c16 < 8 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            aoff[i] = ok ? (unsigned)(((bn * a.H + iy) * a.W + ix) * a.in_stride * EB + c16 * 16) : OOB_OFFSET;
        }
    }
    // The CPRW lanes that fill one LDS row all need the same (offset, tap mask) pair for each of the wave's LA row groups: lane q of
    // such a lane group works out row group q (+ CPRW per pass) only, and the group exchanges the results with ds_bpermute -- one
    // evaluation per lane instead of LA (this setup is ~55 instructions per row; it was a third of a short layer's fixed cost).
    // 1x1 / stride 1 / pad 0 (a third of the launches): output pixel m IS input pixel m and its one tap is always inside the image --
    // no divisions, no exchange (wave-uniform branch)
    const bool pointwise = !HALO && a.ksize == 1 && a.stride == 1 && a.pad == 0;
    if (!HALO && pointwise) {
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            const int prow = (wid + i * NW) * RG + rl;
            const int m = pt * BP + prow;
            const bool ok = m < M && prow < BP;
            // (the descriptor base sits (W + 1) pixels before the tensor, see `shift`)
            rowoff[i] = (unsigned)((m + a.W + 1) * a.in_stride) * (unsigned)EB + (UNI ? (unsigned)(chunk * EPC * EB) : 0u);
            tapmask[i] = ok ? 1u : 0u;
        }
    }
    if (!HALO && !pointwise) {
        const int q = lane % CPRW;
#pragma unroll
        for (int p0 = 0; p0 < LA; p0 += CPRW) {
            const int mine = p0 + q;                    // the row group this lane evaluates in this pass
            const int prow = (wid + mine * NW) * RG + rl;
            const int m = pt * BP + prow;
            unsigned mask = 0, off = 0;
            if (mine < LA && m < M && prow < BP) {
                const int n = fast_div(m, a.howo_mul, a.howo_shift);
                const int rem = m - n * HoWo;
                const int oy = fast_div(rem, a.wo_mul, a.wo_shift);
                const int ox = rem - oy * a.Wo;
                const int iy0 = oy * a.stride - a.pad, ix0 = ox * a.stride - a.pad;
                off = (unsigned)(((n * a.H + iy0) * a.W + ix0 + a.W + 1) * a.in_stride) * (unsigned)EB;
                if (a.ksize == 3) {
                    unsigned ry = 0, cx = 0;
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        ry |= ((unsigned)(iy0 + d) < (unsigned)a.H ? 1u : 0u) << d;
                        cx |= ((unsigned)(ix0 + d) < (unsigned)a.W ? 1u : 0u) << d;
                    }
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh)
                        if (ry & (1u << kh)) mask |= cx << (3 * kh);
                } else if (a.ksize == 1) {
                    mask = ((unsigned)iy0 < (unsigned)a.H && (unsigned)ix0 < (unsigned)a.W) ? 1u : 0u;
                } else {
                    for (int t = 0; t < KK; ++t) {
                        const int kh = t / a.ksize, kw = t - kh * a.ksize;
                        if ((unsigned)(iy0 + kh) < (unsigned)a.H && (unsigned)(ix0 + kw) < (unsigned)a.W) mask |= 1u << t;
                    }
                }
            }
#pragma unroll
            for (int i = p0; i < LA && i < p0 + CPRW; ++i) {
                const int src = ((lane & ~(CPRW - 1)) | (i - p0)) << 2;          // byte address of the source lane
                rowoff[i] = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)off) + (UNI ? (unsigned)(chunk * EPC * EB) : 0u);
                tapmask[i] = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)mask);
            }
        }
    }

    // K cursor.  UNI: scalars (tap, kh, kw, channel base).  Otherwise per lane (chunk-dependent).
    int s_tap = 0, s_kh = 0, s_kw = 0, s_kb = 0, s_cb = 0;   // UNI: tap, its (kh, kw), channel offset in the chunk, chunk base
    int v_kc = chunk * EPC, v_tap = 0;                  // !UNI
    if (!UNI)
        while (v_kc >= a.Cin_pad) { v_kc -= a.Cin_pad; ++v_tap; }
    int s_wk = 0;                                       // byte offset of the K-step in a filter row
    auto stage = [&](char *sbase, const bool with_w = true) {
        char *dx = sbase + wid * 1024;
        char *dw = sbase + BPL * RB + wid * 1024;
        if (UNI) {
            // K order: channel chunks of a.kchunk outermost, the taps inside a chunk, the chunk's channels innermost
            // (k = (chunk * KK + tap) * kchunk + c).  Consecutive K-steps then read the SAME channels of neighbouring
            // pixels -- 175 of a tile's 176 rows are the rows of the previous tap shifted by one pixel -- so the re-reads
            // hit the CU's vector L1 instead of going back to L2 (with tap-outermost order the reuse distance was a whole
            // tap, 4-8 K-steps).
            const unsigned tapbit = s_cb < a.Cin_pad ? 1u << s_tap : 0u;
            const int soff = ((s_kh * a.W + s_kw) * a.in_stride + s_cb + s_kb) * EB;
#pragma unroll
            for (int i = 0; i < LA; ++i) {
                const unsigned vo = (tapmask[i] & tapbit) ? rowoff[i] : OOB_OFFSET;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void *)(dx + i * NW * 1024), 16, vo, soff, 0, 0);
            }
            // branch-free cursor update (a branch here would split the K-step into basic blocks and keep the scheduler
            // from placing these loads between the MFMAs)
            s_kb += BKE;
            const int w1 = s_kb >= a.kchunk ? 1 : 0;
            s_kb = w1 ? 0 : s_kb; s_tap += w1; s_kw += w1;
            const int w2 = s_kw == a.ksize ? 1 : 0;
            s_kw = w2 ? 0 : s_kw; s_kh += w2;
            const int w3 = s_tap == KK ? 1 : 0;
            s_tap = w3 ? 0 : s_tap; s_kh = w3 ? 0 : s_kh; s_cb += w3 ? a.kchunk : 0;
        } else {
            int kh = 0, kw = 0;
            if (a.ksize == 3) { kh = (v_tap * 11) >> 5; kw = v_tap - kh * 3; }
            else if (a.ksize != 1) { kh = v_tap / a.ksize; kw = v_tap - kh * a.ksize; }
            const unsigned tapbit = v_tap < KK ? 1u << v_tap : 0u;
            const unsigned delta = (unsigned)((kh * a.W + kw) * a.in_stride + v_kc) * (unsigned)EB;
#pragma unroll
            for (int i = 0; i < LA; ++i) {
                const unsigned vo = (tapmask[i] & tapbit) ? rowoff[i] + delta : OOB_OFFSET;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void *)(dx + i * NW * 1024), 16, vo, 0, 0, 0);
            }
            v_kc += BKE;
            while (v_kc >= a.Cin_pad) { v_kc -= a.Cin_pad; ++v_tap; }
        }
        if (with_w) {
#pragma unroll
            for (int i = 0; i < LB; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void *)(dw + i * NW * 1024), 16, woff[i], s_wk, 0, 0);
        }
        s_wk += RB;
    };

    f32x4 acc[TC][TP];
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int KT = a.Kpad / BKE;
    // fused 1x1 tail (see the epilogue): one channel tile covers the whole output (BC == Cout, checked by the host), the
    // consumer waves split the C2 = BC / 2 tail channels 16 apiece
    // (8-wave shapes only: in the 4-wave 176x128 shapes the extra live registers push the kernel past 256 VGPRs and cost the
    // second resident workgroup per CU -- measured slower overall even where the pair itself got faster)
    // (bf16: not in the role-split shapes -- the tail's addresses, hoisted above the K loop, push their 168-VGPR budget into spills)
    constexpr bool TAIL_OK = !SPLIT && WP == 1 && NC == 8 && (EB == 2 ? NL == 0 && (BC == 256 || BC == 128) : BC == 256);
    // (Tried and dropped: placing one LDS-DMA of the next stage behind every MFMA group with sched_group_barrier instead of
    // issuing the whole stage first.  A/B on one MI355X box, YOLOv3-416 batch 32: 2 % SLOWER in both bf16 (3.48 vs 3.40 ms)
    // and fp8 (2.39 vs 2.34 ms) -- a DMA blocks its wave's issue for ~60 cycles wherever it is placed, and the MFMA pipe
    // holds no queue to ride it out.)
    constexpr int D = NS - 1;                  // prefetch distance in K-steps
    if constexpr (HALO) {
        // LDS: [halo tile 0 | halo tile 1 | filter stage 0 | filter stage 1].  Chunk 0's halo tile and K-step 0's filters first.
        if (is_loader) {
#pragma unroll
            for (int i = 0; i < LAH; ++i)
                if (wid + i * NW < APIECES) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void *)(smem + (wid + i * NW) * 1024), 16, aoff[i], 0, 0, HALO_LOAD_AUX);
#pragma unroll
            for (int t = 1; t < (FREE ? NS - 1 : 1); ++t)        // (stage 0 was requested ahead of the halo addressing)
                if (t < KT) {
#pragma unroll
                    for (int i = 0; i < LB; ++i)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void *)(smem + 2 * ACT_BYTES + t * FSTAGE + (FREE ? wid * LB + i : wid + i * NW) * 1024), 16, woff[i], t * RB, 0, 0);
                }
            s_wk = (FREE ? NS - 1 : 1) * RB;
        }
    } else {
#pragma unroll
        for (int t = 0; t < D; ++t)
            if (t < KT && is_loader) stage(smem + t * STAGE_BYTES, t != 0);      // (step 0's filters were requested ahead of the addressing)
    }

    const int l15 = lane & 15, lq = lane >> 4;
    // fragment read offsets inside a stage (two K-halves), constant over the loop
    const int sw0 = BK == 64 ? ((0 + lq) ^ (l15 & 7)) << 4 : (lq ^ (3 * ((l15 >> 2) & 1))) << 4;
    const int sw1 = ((4 + lq) ^ (l15 & 7)) << 4;            // second K-half (BK = 64 only)
    const int offx = (wpi * TP * 16 + l15) * RB;
    const int offw = (HALO ? 0 : BPL * RB) + (wci * TC * 16 + l15) * RB;       // halo form: relative to the filter stage
    // halo form: byte offset of tap (0,0) of this lane's pixel of sub-tile j in a halo tile (+ its 16-B piece of the first K-half)
    int abase[HALO ? TP : 1];
    if constexpr (HALO) {
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            const int p = (int)((perm_rd[j / 4] >> (8 * (j % 4))) & 0xffu);
            const int y = p / HBW, x = p - y * HBW;
            abase[j] = (y * HPW + x) * APIX + lq * 16;
        }
    }
    int cur = 0, nxt = D % NS;                 // stage being multiplied / stage being filled
    // DIAG (separate diagnostic instantiation, never the shipped kernel): s_memtime stamps around the phases of a K-step
    unsigned long long t_wait = 0, t_issue = 0, t_mma = 0, t_all0 = 0, t_first = 0, te1 = 0, te2 = 0, te3 = 0, te4 = 0, te5 = 0, te6 = 0;
    auto stamp = [&]() -> unsigned long long {
        unsigned long long t = 0;
        if (DIAG && !a.dbg_light) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); __builtin_amdgcn_sched_barrier(0); }
        return t;
    };
    auto stamp_loop = [&]() -> unsigned long long {      // the two stamps around the K loop: always taken in a diagnostic build
        unsigned long long t = 0;
        if (DIAG) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); __builtin_amdgcn_sched_barrier(0); }
        return t;
    };
    unsigned long long rt0 = 0;
    if (DIAG) { t_all0 = stamp_loop(); rt0 = __builtin_amdgcn_s_memrealtime(); }
    // one K-step.  `fill` / `sb`: the stage being filled and the stage being multiplied.  They are distinct __restrict__ parameters
    // on purpose: hipcc orders every ds_read behind ALL outstanding LDS-DMA (s_waitcnt vmcnt(0) before the first
    // fragment read of each K-step, i.e. the loads just issued for the NEXT step were waited for before this step's
    // MFMAs) unless alias-scope metadata proves the DMA target and the read are different memory; restrict parameters
    // of an inlined function are what produces that metadata.
    // MFMA phase of one K-step.  xrow(j) / wrow(i): LDS address of this lane's row of pixel sub-tile j / channel sub-tile i;
    // xs0 / xs1: byte offsets of the lane's 16-B piece of a pixel row in the two K-halves (filter rows: sw0 / sw1).
    auto mma_phase = [&](auto xrow, auto wrow, const int xs0, const int xs1) {
        // Fragment reads are software-pipelined PD MFMA groups ahead and PINNED there with sched_group_barrier: left
        // alone the scheduler hoists every ds_read of the (half) step above the first MFMA and waits lgkmcnt(0), so
        // the LDS pipe and the MFMA pipe take turns instead of overlapping (all four waves are in the same phase).
        // A "group" is the TC MFMAs that share one pixel fragment.
        if (EB == 1) {
            // e4m3: one K = 128 MFMA per tile pair.  Lane (l15, lq) supplies 16-B chunks lq and lq + 4 of its row for
            // BOTH operands, so the pairing of K indices inside the instruction is consistent whatever its internal
            // order (tools/probe/mfma_fp8.hip).  Those are the chunks of the two bf16 half-steps, i.e. the one
            // assignment for which the XOR swizzle is conflict-free under ds_read_b128's lane groups; the "natural"
            // chunks 2*lq, 2*lq + 1 collide two-way in every group (measured: +35 % on the fragment-read phase).
            if (is_consumer) {
                constexpr int PD = TP < 3 ? TP : 3;
                i32x8 fw[TC], fx[TP];
                auto frag = [&](const char *row, const int s0, const int s1) -> i32x8 {
                    const uint4 lo = *(const uint4 *)(row + s0), hi = *(const uint4 *)(row + s1);
                    return i32x8{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
                };
#pragma unroll
                for (int i = 0; i < TC; ++i) fw[i] = frag(wrow(i), sw0, sw1);
#pragma unroll
                for (int j = 0; j < PD; ++j) fx[j] = frag(xrow(j), xs0, xs1);
#pragma unroll
                for (int j = 0; j < TP; ++j) {
                    if (j + PD < TP) fx[j + PD] = frag(xrow(j + PD), xs0, xs1);
#pragma unroll
                    for (int i = 0; i < TC; ++i)
                        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fw[i], fx[j], acc[i][j], 0, 0, 0, 0, 0, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 2 * (TC + PD), 0);
#pragma unroll
                for (int j = 0; j < TP; ++j) {
                    if (j + PD < TP) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, TC, 0);
                }
            }
        } else if constexpr (PAIRK) {
            // Pair K-step: the row's first K-half is hi, its second lo (activations and filters alike).  Per pixel sub-tile j three ops of TC
            // MFMAs each: A_j = W_hi x_hi[j], B_j = W_lo x_hi[j], C_j = W_hi x_lo[j], all into acc[.][j].  They are issued skewed by one
            // sub-tile -- step s: B_{s-1}, A_s, C_{s-1} -- so that two ops on the same accumulators are never adjacent (TC = 1 shapes would
            // otherwise issue three dependent MFMAs back to back).  Fragments in order of first use: hi_0, hi_1, lo_0, hi_2, lo_1, ...,
            // hi_{TP-1}, lo_{TP-2}, lo_{TP-1}; each is read PD first-uses ahead and pinned there, as in the plain loop.
            if (is_consumer) {
                constexpr int NF = 2 * TP;
                constexpr int PD = NF < PAIR_PD ? NF : PAIR_PD;
                bf16x8 fwh[TC], fwl[TC], fx[NF];
                auto rdf = [&](auto pc) {
                    constexpr int p = decltype(pc)::value;
                    constexpr bool lo = p == NF - 1 || (p >= 2 && p % 2 == 0);
                    constexpr int j = p == 0 ? 0 : p == NF - 1 ? TP - 1 : lo ? p / 2 - 1 : (p + 1) / 2;
                    fx[p] = *(const bf16x8 *)(xrow(j) + (lo ? xs1 : xs0));
                };
                auto first_use = [&](auto pc) {           // the op about to run uses fragment p for the first time: keep the read-ahead PD deep
                    constexpr int p = decltype(pc)::value;
                    if constexpr (p + PD < NF) rdf(std::integral_constant<int, p + PD>{});
                };
#pragma unroll
                for (int i = 0; i < TC; ++i) { fwh[i] = *(const bf16x8 *)(wrow(i) + sw0); fwl[i] = *(const bf16x8 *)(wrow(i) + sw1); }
                static_for<PD>([&](auto pc) { rdf(pc); });
                static_for<TP + 1>([&](auto sc) {
                    constexpr int s = decltype(sc)::value;
                    constexpr int pa = s == 0 ? 0 : 2 * s - 1;                        // hi_s
                    constexpr int pb = s <= 1 ? 0 : 2 * (s - 1) - 1;                  // hi_{s-1}
                    constexpr int pcx = s == TP ? NF - 1 : 2 * s;                      // lo_{s-1}
                    if constexpr (s >= 1) {
#pragma unroll
                        for (int i = 0; i < TC; ++i) acc[i][s - 1] = mma16<true>(fwl[i], fx[pb], acc[i][s - 1]);
                    }
                    if constexpr (s < TP) {
                        first_use(std::integral_constant<int, pa>{});
#pragma unroll
                        for (int i = 0; i < TC; ++i) acc[i][s] = mma16<true>(fwh[i], fx[pa], acc[i][s]);
                    }
                    if constexpr (s >= 1) {
                        first_use(std::integral_constant<int, pcx>{});
#pragma unroll
                        for (int i = 0; i < TC; ++i) acc[i][s - 1] = mma16<true>(fwh[i], fx[pcx], acc[i][s - 1]);
                    }
                });
                __builtin_amdgcn_sched_group_barrier(0x100, 2 * TC + PD, 0);
                static_for<TP + 1>([&](auto sc) {
                    constexpr int s = decltype(sc)::value;
                    constexpr int pa = s == 0 ? 0 : 2 * s - 1, pcx = s == TP ? NF - 1 : 2 * s;
                    if constexpr (s >= 1) __builtin_amdgcn_sched_group_barrier(0x008, TC, 0);
                    if constexpr (s < TP) {
                        if constexpr (pa + PD < NF) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, TC, 0);
                    }
                    if constexpr (s >= 1) {
                        if constexpr (pcx + PD < NF) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, TC, 0);
                    }
                });
            }
        } else if (is_consumer) {
            constexpr int KH = BK / 32;                 // K halves of 32 per step
            constexpr int NG = KH * TP;                 // MFMA groups per step
            constexpr int PD = NG < 4 ? NG : 4;
            bf16x8 fw[KH][TC], fx[NG];
            auto rdw = [&](int kk) {
#pragma unroll
                for (int i = 0; i < TC; ++i) fw[kk][i] = *(const bf16x8 *)(wrow(i) + (kk ? sw1 : sw0));
            };
            auto rdx = [&](int g) {
                const int kk = g / TP, j = g - kk * TP;
                fx[g] = *(const bf16x8 *)(xrow(j) + (kk ? xs1 : xs0));
            };
            // the second half's filter fragments are read when the pixel prefetch first reaches that half
            constexpr bool W1_UPFRONT = KH == 2 && PD >= TP;
            rdw(0);
            if (W1_UPFRONT) rdw(1);
#pragma unroll
            for (int g = 0; g < PD; ++g) rdx(g);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + PD < NG) {
                    if (KH == 2 && !W1_UPFRONT && g + PD == TP) rdw(1);
                    rdx(g + PD);
                }
                const int kk = g / TP, j = g - kk * TP;
#pragma unroll
                for (int i = 0; i < TC; ++i)
                    acc[i][j] = mma16<H16>(fw[kk][i], fx[g], acc[i][j]);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, TC + (W1_UPFRONT ? TC : 0) + PD, 0);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + PD < NG) {
                    if (KH == 2 && !W1_UPFRONT && g + PD == TP) __builtin_amdgcn_sched_group_barrier(0x100, TC, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, TC, 0);
            }
        }
    };
    auto kstep = [&](int kt, char *__restrict__ fill, const char *__restrict__ sb, const bool LOAD) {
        const unsigned long long s0 = stamp();
        // K-step kt has landed once at most (D-1) younger K-steps' loads remain outstanding (in-order counter)
        if (is_loader) { if (LOAD) wait_vmcnt<(D - 1) * L>(); else wait_vmcnt<0>(); }
        block_barrier();                       // everybody's part of K-step kt is in LDS; stage `nxt` is free again
        const unsigned long long s1 = stamp();
        if (DIAG && !t_first) t_first = s1;
        if (LOAD && is_loader) stage(fill);
        const unsigned long long s2 = stamp();
        __builtin_amdgcn_s_setprio(2);          // MFMA phase: win issue arbitration against the co-resident workgroup's DMA / epilogue phases
        mma_phase([&](int j) { return sb + offx + j * 16 * RB; }, [&](int i) { return sb + offw + i * 16 * RB; }, sw0, sw1);
        __builtin_amdgcn_s_setprio(0);
        if (DIAG) {
            if (!a.dbg_light) asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
            const unsigned long long s3 = stamp();
            t_wait += s1 - s0; t_issue += s2 - s1; t_mma += s3 - s2;
        }
    };
    // halo form: one K-step = tap TAP of the current channel chunk.  fill_f / sb_f: filter stage being filled / multiplied;
    // fill_a / sb_a: halo tile of the NEXT chunk (this wave fetches its piece number TAP of it now) / of this chunk.
    auto hstep = [&](auto tapc, char *__restrict__ fill_f, char *__restrict__ fill_a, const char *__restrict__ sb_f, const char *__restrict__ sb_a,
                     const bool load_f, const bool load_a, const int next_chunk) {
        constexpr int TAP = decltype(tapc)::value;
        if (is_loader) wait_vmcnt<0>();
        block_barrier();                       // this K-step's filters (and, at tap 0, the chunk's halo tile) are in LDS
        if (is_loader) {
            if (load_f) {
#pragma unroll
                for (int i = 0; i < LB; ++i)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void *)(fill_f + wid * 1024 + i * NW * 1024), 16, woff[i], s_wk, 0, 0);
                s_wk += RB;
            }
            if constexpr (TAP < LAH)
                if (load_a && wid + TAP * NW < APIECES)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void *)(fill_a + (wid + TAP * NW) * 1024), 16, aoff[TAP], next_chunk * RB, 0, HALO_LOAD_AUX);
        }
        __builtin_amdgcn_s_setprio(2);
        mma_phase([&](int j) { return sb_a + abase[j] + ((TAP / 3) * HPW + TAP % 3) * APIX; }, [&](int i) { return sb_f + offw + i * 16 * RB; }, 0, 64);
        __builtin_amdgcn_s_setprio(0);
    };
    // free-running halo form: one K-step of ONE wave (see the template comment).  Issue order inside a step: the wave's halo piece
    // of the next chunk (taps 0..LAH-1), the filters of the NEXT K-step into the stage this wave read in the previous step, then the
    // counted wait for THIS step's filters (issued one step ago: everything but what was just issued may still be in flight).
    int f_ops1 = 0, f_ops2 = 0;
    if (FREE && NS >= 3) { f_ops1 = NS == 3 ? LB : LB; f_ops2 = NS == 4 ? LB : 0; }     // the prologue's stages 1.. are in flight behind stage 0
    auto fstep = [&](auto tapc, char *__restrict__ fill_f, char *__restrict__ fill_a, const char *__restrict__ sb_f, const char *__restrict__ sb_a,
                     const bool load_f, const bool load_a, const int next_chunk) {
        constexpr int TAP = decltype(tapc)::value;
        constexpr bool ATAP = TAP < LAH;
        const bool do_a = ATAP && load_a && wid + TAP * NW < APIECES;          // wave-uniform
        // LDS-DMA instructions this wave issues in this step; f_ops1 / f_ops2: in the previous step / the one before (the filters of
        // K-step kt were issued NS - 1 steps ago: everything issued since may still be in flight when they are waited for)
        const int ops_now = (do_a ? 1 : 0) + (load_f ? LB : 0);
        auto issue = [&]() {
            if constexpr (ATAP)
                if (do_a) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void *)(fill_a + (wid + TAP * NW) * 1024), 16, aoff[TAP < LAH ? TAP : 0], next_chunk * RB, 0, HALO_LOAD_AUX);
            if (load_f) {
#pragma unroll
                for (int i = 0; i < LB; ++i)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void *)(fill_f + (wid * LB + i) * 1024), 16, woff[i], s_wk, 0, 0);
                s_wk += RB;
            }
        };
        const unsigned long long s0 = stamp();
        unsigned long long s1 = 0, s2 = 0;
        if constexpr (TAP == 0) {
            // chunk boundary: this wave's filters of the step and its pieces of this chunk's halo tile have landed; after the barrier
            // everybody's have, and nobody reads the previous chunk's halo tile any more (the next chunk's pieces overwrite it)
            if constexpr (NS == 2) wait_vmcnt<0>(); else wait_vmcnt_rt<(NS - 2) * (LB + 1)>(NS == 3 ? f_ops1 : f_ops1 + f_ops2);
            block_barrier();
            if (NC == 8 && wave_id >= NC / 2) __builtin_amdgcn_s_sleep(TP * TC * (PAIRK ? PAIR_SLEEP : 32) / 64);      // half a K-step behind the SIMD's other wave (eight-wave forms: two waves per SIMD)
            s1 = stamp();
            issue();
            s2 = stamp();
            if (DIAG) { t_wait += s1 - s0; t_issue += s2 - s1; if (!t_first) t_first = s1; }
        } else {
            issue();
            s1 = stamp();
            if constexpr (NS == 2) {
                if (load_f) { if (do_a) wait_vmcnt<LB + 1>(); else wait_vmcnt<LB>(); }
                else { if (do_a) wait_vmcnt<1>(); else wait_vmcnt<0>(); }
            } else wait_vmcnt_rt<(NS - 1) * (LB + 1)>(ops_now + f_ops1 + (NS == 4 ? f_ops2 : 0));
            s2 = stamp();
            if (DIAG) { t_issue += s1 - s0; t_wait += s2 - s1; }
        }
        f_ops2 = f_ops1; f_ops1 = ops_now;
        __builtin_amdgcn_s_setprio(2);
        mma_phase([&](int j) { return sb_a + abase[j] + ((TAP / 3) * HPW + TAP % 3) * APIX; }, [&](int i) { return sb_f + offw + i * 16 * RB; }, 0, 64);
        __builtin_amdgcn_s_setprio(0);
        if (DIAG) { if (!a.dbg_light) asm volatile("s_nop 7\n\ts_nop 7" ::: "memory"); t_mma += stamp() - s2; }
    };
    if constexpr (HALO) {
        char *const act0 = smem, *const act1 = smem + ACT_BYTES, *const flt0 = smem + 2 * ACT_BYTES, *const flt1 = flt0 + FSTAGE;
        const int NCH = KT / 9;                    // channel chunks; K-step kt = chunk * 9 + tap multiplies filter stage kt & 1
        int kt = 0, cur_i = 0, fill_i = NS - 1;
        for (int c = 0; c < NCH; ++c) {
            char *const sa = (c & 1) ? act1 : act0, *const fa = (c & 1) ? act0 : act1;
            const bool la = c + 1 < NCH;
            static_for<9>([&](auto tapc) {
                if constexpr (FREE && NS > 2) {
                    // ring of NS private filter stages: K-step kt multiplies stage kt % NS and refills the stage it read one step ago
                    fstep(tapc, flt0 + fill_i * FSTAGE, fa, flt0 + cur_i * FSTAGE, sa, kt + NS - 1 < KT, la, c + 1);
                    cur_i = cur_i + 1 == NS ? 0 : cur_i + 1; fill_i = fill_i + 1 == NS ? 0 : fill_i + 1;
                } else {
                    const bool odd = kt & 1;
                    if constexpr (FREE) fstep(tapc, odd ? flt0 : flt1, fa, odd ? flt1 : flt0, sa, kt + 1 < KT, la, c + 1);
                    else hstep(tapc, odd ? flt0 : flt1, fa, odd ? flt1 : flt0, sa, kt + 1 < KT, la, c + 1);
                }
                ++kt;
            });
        }
    } else {
        int kt = 0;
        if (NS == 2) {
            char *const s0 = smem, *const s1 = smem + STAGE_BYTES;
            const int KM = KT - D;              // K-steps that still have a later step to load for
            for (; kt + 1 < KM; kt += 2) { kstep(kt, s1, s0, true); kstep(kt + 1, s0, s1, true); }
            if (kt < KM) { kstep(kt, s1, s0, true); ++kt; }
            for (; kt < KT; ++kt) kstep(kt, (kt & 1) ? s0 : s1, (kt & 1) ? s1 : s0, false);      // stage = kt & 1
        } else {
            for (; kt < KT; ++kt) {
                kstep(kt, smem + nxt * STAGE_BYTES, smem + cur * STAGE_BYTES, kt + D < KT);
                cur = cur + 1 == NS ? 0 : cur + 1;
                nxt = nxt + 1 == NS ? 0 : nxt + 1;
            }
        }
    }
    unsigned long long t_loop_end = 0, rt_loop = 0;
    if (DIAG) { t_loop_end = stamp_loop(); rt_loop = __builtin_amdgcn_s_memrealtime(); }

    // ---- epilogue ----
    // bf16 / fp8 stores and shortcut loads go through buffer descriptors based at this workgroup's first pixel and channel: the lane
    // offset is 32-bit (pixel index relative to the tile origin x pixel stride, one 24-bit multiply) and a row or channel that is not
    // stored gets an out-of-range offset (loads return zero, stores are dropped) -- no 64-bit address arithmetic, no exec-mask
    // branches.  The epilogue is VALU-issue-bound (two waves per SIMD, ~4 cycles per instruction each): instruction count is its cost.
    const size_t m0 = HALO ? ((size_t)(bn * a.H + by * HBH) * a.W + bx * HBW) : (size_t)pt * BP;      // tile origin (flat pixel index)
    const int rows_left = HALO ? BP : (int)((size_t)M - m0 < (size_t)BP ? (size_t)M - m0 : (size_t)BP);    // tiled form: rows of the tile inside the tensor
    // descriptor based at `p` (a wave-uniform address; the read-first-lane pins it to SGPRs, otherwise every access is wrapped in a
    // waterfall loop)
    auto tile_rsrc = [&](const void *p) -> __amdgpu_buffer_rsrc_t {
        const unsigned long long v = (unsigned long long)p;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return __builtin_amdgcn_make_buffer_rsrc((void *)(((unsigned long long)hi << 32) | lo), 0, BUF_RECORDS, 0x00020000);
    };
    // byte offset (from the tile origin, channel tile ct) of 16-byte piece c of the tile -- row-major, `cpr` pieces of `cpp` channels
    // per row, pixel stride `sb` bytes -- or OOB_OFFSET when the piece is not stored; also returns the piece's row and position
    // halo form: rows / columns of this block inside the image (13 unless the block is a ragged one on the bottom / right edge)
    const unsigned ylim = HALO ? (unsigned)(a.H - by * HBH < HBH ? a.H - by * HBH : HBH) : 0u, xlim = HALO ? (unsigned)(a.W - bx * HBW < HBW ? a.W - bx * HBW : HBW) : 0u;
    auto piece_off = [&](int c, int cpr, int cpp, unsigned sb, int &row, int &cc) -> unsigned {
        row = c / cpr; cc = c - row * cpr;
        unsigned rel; bool ok;
        if constexpr (HALO) {
            const unsigned y = __umul24((unsigned)row, halo_div_mul(HBW)) >> 16, x = (unsigned)row - __umul24(y, (unsigned)HBW);       // row / BW (row < 256)
            rel = __umul24(y, (unsigned)a.W) + x; ok = y < ylim && x < xlim;       // (y < BH is row < BH * BW)
        }
        else { rel = (unsigned)row; ok = row < rows_left; }
        ok = ok && ct * BC + cc * cpp < a.Cout;
        const unsigned off = __umul24(rel, sb) + (unsigned)cc * 16u;
        return ok ? off : OOB_OFFSET;
    };
    if (SPLIT && a.out_dt != DT_F32) {
        // ---- split fp16 pairs (YOLO_FP16X2): the tile goes through LDS as FP32, one half of its channels at a time (the same LDS footprint
        //      as the 16-bit tile), and leaves as 16-byte pieces of eight channels: v -> hi = f16(v), lo = f16(v - hi).  Round 6: the
        //      INTERLEAVED pair layout -- channel c of a pixel has its hi at element (c / 32) * 64 + c % 32 and its lo 32 elements
        //      (64 bytes) further, so that a 128-byte run of the pixel is one K-step row [hi 32 | lo 32] of the next conv (PAIRK) and a
        //      concatenation of pair tensors is a channel window again.  Channels up to the next multiple of 32 are written (zeros: their
        //      filter rows and biases are zero), the consumer reads whole groups.  A fused shortcut is exactly the separate launch's
        //      arithmetic (k_add_split): the conv's own pair is formed first, then (f_hi + f_lo) + (x_hi + x_lo) in fp32, split again. ----
        if constexpr (SPLIT) {
        constexpr int HC = BC / 2, RS4 = HC * 4 + 16, NT = 64 * NTOT;
        static_assert(!SPLIT || HC % 16 == 0, "a 16-channel sub-tile belongs to one half");
        static_assert(!SPLIT || BC % 32 == 0, "a channel tile is whole 32-channel groups");
        constexpr int CPRH = HC / 8, NITH = (BP * CPRH + NT - 1) / NT;       // 8-channel pieces per row of a half / per thread
        const char *__restrict__ res = (const char *)a.res;
        const __amdgpu_buffer_rsrc_t rs_out = tile_rsrc((char *)a.out + (m0 * a.out_stride + (size_t)ct * BC * 2) * 2);
        const __amdgpu_buffer_rsrc_t rs_res = tile_rsrc(res ? res + (m0 * a.res_stride + (size_t)ct * BC * 2) * 2 : nullptr);
        const unsigned out_sb = a.out_stride * 2u, res_sb = a.res_stride * 2u;
        const int cout32 = (a.Cout + 31) & ~31;
        int prow4[HALO ? TP : 1];
        if constexpr (HALO) {
#pragma unroll
            for (int j = 0; j < TP; ++j) prow4[j] = (int)((perm_wr[j / 4] >> (8 * (j % 4))) & 0xffu) * RS4;
        }
        f32x4 bvs[TC];
#pragma unroll
        for (int i = 0; i < TC; ++i) bvs[i] = is_consumer ? *(const f32x4 *)(a.bias + ct * BC + (wci * TC + i) * 16 + lq * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        const float slope = a.act == ACT_LEAKY ? 0.1f : 1.0f;
        auto split8 = [](const float *v, u32x4_t &H, u32x4_t &L) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { uint32_t h, l; pair_split2(v[2 * q], v[2 * q + 1], h, l); H[q] = h; L[q] = l; }
        };
        static_for<2>([&](auto halfc) {
            constexpr int HALF = decltype(halfc)::value;
            // this thread's shortcut pieces of the half (hi and lo), requested ahead of the barrier and the LDS pass
            u32x4_t rh[NITH], rl[NITH];
            unsigned offs[NITH];
#pragma unroll
            for (int it = 0; it < NITH; ++it) {
                int row, cc;
                unsigned off = piece_off(tid + it * NT, CPRH, 0, out_sb, row, cc);
                const int cw = HALF * HC + cc * 8;                               // first channel of the piece within the tile
                const unsigned ilv = ((unsigned)(cw >> 5) << 7) | ((unsigned)(cw & 31) << 1);      // byte offset of its hi piece from the tile's first group
                if (ct * BC + cw >= cout32 || ((BP * CPRH) % NT != 0 && tid + it * NT >= BP * CPRH)) off = OOB_OFFSET;
                offs[it] = off == OOB_OFFSET ? OOB_OFFSET : off - (unsigned)cc * 16u + ilv;
                if (res) {
                    int r2, c2;
                    unsigned ro = piece_off(tid + it * NT, CPRH, 0, res_sb, r2, c2);
                    ro = off == OOB_OFFSET || ro == OOB_OFFSET ? OOB_OFFSET : ro - (unsigned)c2 * 16u + ilv;
                    rh[it] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, ro, 0, RES_LOAD_AUX);
                    rl[it] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, ro == OOB_OFFSET ? OOB_OFFSET : ro + 64u, 0, RES_LOAD_AUX);
                }
            }
            block_barrier();                                  // the stages (half 0) / the previous half's tile (half 1) are no longer read
            if (is_consumer)
#pragma unroll
            for (int i = 0; i < TC; ++i) {
                const int chl = (wci * TC + i) * 16 + lq * 4;     // channel within the tile
                if (chl / HC != HALF) continue;                   // (wave- and sub-tile-uniform)
#pragma unroll
                for (int j = 0; j < TP; ++j) {
                    f32x4 v = acc[i][j] + bvs[i];
                    const f32x4 t = v * slope;
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = vmax_f32(v[q], t[q]);
                    *(f32x4 *)(smem + (HALO ? prow4[j] : ((wpi * TP + j) * 16 + l15) * RS4) + (chl - HALF * HC) * 4) = v;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            block_barrier();
#pragma unroll
            for (int it = 0; it < NITH; ++it) {
                if ((BP * CPRH) % NT != 0 && tid + it * NT >= BP * CPRH) continue;
                const int c = tid + it * NT, row = c / CPRH, cc = c - row * CPRH;
                const f32x4 v0 = *(const f32x4 *)(smem + row * RS4 + cc * 32), v1 = *(const f32x4 *)(smem + row * RS4 + cc * 32 + 16);
                float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                u32x4_t H, L;
                split8(v, H, L);
                if (res) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float f0, f1, x0, x1;
                        pair_join2(H[q], L[q], f0, f1); pair_join2(rh[it][q], rl[it][q], x0, x1);
                        v[2 * q] = f0 + x0; v[2 * q + 1] = f1 + x1;
                    }
                    split8(v, H, L);
                }
                const unsigned off = offs[it];
                __builtin_amdgcn_raw_buffer_store_b128(H, rs_out, off, 0, OUT_STORE_AUX);
                __builtin_amdgcn_raw_buffer_store_b128(L, rs_out, off == OOB_OFFSET ? OOB_OFFSET : off + 64u, 0, OUT_STORE_AUX);
            }
        });
        }
    } else if (a.out_dt != DT_F32) {
        // bf16 / fp8 output: scale + bias + activation in registers, then the tile goes through LDS (as bf16) so that
        // global stores (and the residual loads) are 16 B per lane along the channel axis -- whole 128-B lines per
        // pixel instead of 16 scattered 32-B pieces per store instruction (row-per-lane dwordx2 stores are
        // issue-bound: ~600 cycles each).  A 16-B piece is 8 bf16 or 16 fp8 channels.
        constexpr int RS = BC * 2 + 16;                       // padded LDS row (bytes)
        constexpr int NT = 64 * NTOT;
        constexpr int CPR = BC / 8, NIT = (BP * CPR + NT - 1) / NT;       // bf16 pieces per row / per thread
        constexpr int CPR8 = BC / 16, NIT8 = (BP * CPR8 + NT - 1) / NT;   // fp8 pieces
        const bool out8 = !H16 && a.out_dt == DT_FP8;            // (an fp16 network has no e4m3 tensors)
        const unsigned osz = out8 ? 1u : 2u;                  // bytes per stored element
        const char *__restrict__ res = (const char *)a.res;
        const __amdgpu_buffer_rsrc_t rs_out = tile_rsrc((char *)a.out + (m0 * a.out_stride + (size_t)ct * BC) * osz);
        const __amdgpu_buffer_rsrc_t rs_res = tile_rsrc(res ? res + (m0 * a.res_stride + (size_t)ct * BC) * osz : nullptr);
        const unsigned out_sb = a.out_stride * osz, res_sb = a.res_stride * osz;       // pixel strides in bytes (< 2^24)
        // residual (shortcut source) pieces are fetched now, all at once, so that their latency is covered by the
        // accumulator -> LDS pass below instead of being paid once per piece in the store loop
        u32x4_t rpre[NIT];
        if (res) {
            if (out8) {
#pragma unroll
                for (int it = 0; it < NIT8; ++it) {
                    int row, cc;
                    rpre[it] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, piece_off(tid + it * NT, CPR8, 16, res_sb, row, cc), 0, RES_LOAD_AUX);
                }
            } else {
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    int row, cc;
                    rpre[it] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, piece_off(tid + it * NT, CPR, 8, res_sb, row, cc), 0, RES_LOAD_AUX);
                }
            }
        }
        // halo form: MFMA column (j, l15) holds pixel kHaloPerm13[..] >> 8 of the block; the LDS tile is in raster order
        int prow[HALO ? TP : 1];
        if constexpr (HALO) {
#pragma unroll
            for (int j = 0; j < TP; ++j) prow[j] = (int)((perm_wr[j / 4] >> (8 * (j % 4))) & 0xffu) * RS;      // byte offset of the row in the LDS tile
        }
        // bias (and the fp8 dequantisation scale) of this lane's channels: requested before the barrier, which covers their latency
        f32x4 bvs[TC], svs[TC];
#pragma unroll
        for (int i = 0; i < TC; ++i) {
            const int chl = (wci * TC + i) * 16 + lq * 4;
            bvs[i] = is_consumer ? *(const f32x4 *)(a.bias + ct * BC + chl) : f32x4{0.f, 0.f, 0.f, 0.f};
            svs[i] = f32x4{1.f, 1.f, 1.f, 1.f};
            if (EB == 1 && a.oscale && is_consumer) svs[i] = *(const f32x4 *)(a.oscale + ct * BC + chl);
        }
        block_barrier();                                      // every wave is done reading the last stage
        if (DIAG) te1 = stamp();
        // (Tried and dropped, round 3: converting the accumulators to packed bf16 in registers BEFORE this barrier, so that only the LDS
        // writes remain behind it.  The stamped build moved 1 300 cycles in front of the barrier and took 100 off the phase behind it: that
        // phase is the LDS store path -- 176 ds_write_b64 per wave pair at ~12 cycles each -- not the arithmetic; the step got 1.7 % slower.)
        const float slope = a.act == ACT_LEAKY ? 0.1f : 1.0f;
        {
            if (is_consumer)
#pragma unroll
            for (int i = 0; i < TC; ++i) {
                const int chl = (wci * TC + i) * 16 + lq * 4;     // channel within the tile
                const f32x4 bv = bvs[i], sv = svs[i];
#pragma unroll
                for (int j = 0; j < TP; ++j) {
                    f32x4 v = acc[i][j];
                    if (EB == 1) v = v * sv;
                    v = v + bv;
                    const f32x4 t = v * slope;                    // leaky: max(v, 0.1 v) == v > 0 ? v : 0.1 v; linear: max(v, v)
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = vmax_f32(v[q], t[q]);
                    uint2 pk;
                    pk.x = pack16x2<H16>(v[0], v[1]);
                    pk.y = pack16x2<H16>(v[2], v[3]);
                    *(uint2 *)(smem + (HALO ? prow[j] : ((wpi * TP + j) * 16 + l15) * RS) + chl * 2) = pk;
                    if (j & 1) __builtin_amdgcn_sched_barrier(0);   // bounds the scheduler's look-ahead (one straight-line block of TC * TP sub-tiles otherwise)
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            block_barrier();
        }
        if (DIAG) te2 = stamp();
        // fused 1x1 tail (bf16): eight consumer waves = T2G groups of T2W 16-channel tiles x T2P parts of the tile's pixel sub-tiles
        // (T2W = 2 in the halo forms, 1 in the tiled 8-wave shape, which has no registers for 64 of filter fragments).  A wave's filter
        // fragments (16 T2W rows x BC) are fetched here -- the accumulators are dead, the store loop below covers the latency
        constexpr int C2 = BC / 2, K2S = BC / 32, T2W = HALO ? 2 : 1, T2G = (C2 / (16 * T2W)) > 0 ? C2 / (16 * T2W) : 1, T2P = 8 / T2G > 0 ? 8 / T2G : 1;
        // (round 5) the tail may be a detection HEAD (a.tail_f32; halo forms with BC = 256 only): up to 256 filters, so every wave owns its own
        // 32 of them for ALL pixel sub-tiles, and the result leaves as fp32 rows straight from the accumulators, as the stand-alone head does
        // HEADT: its own instantiation -- as a run-time mode of the ordinary tail kernels the extra code cost THEIR layers 1 % (same-box A/B)
        constexpr bool HEAD_TAIL_OK = HEADT && TAIL_OK && EB == 2 && HALO && BC == 256;
        const bool head_tail = HEAD_TAIL_OK && a.tail_f32;
        const int t2g = head_tail ? wave_id : wave_id % T2G, t2p = head_tail ? 0 : wave_id / T2G;
        bf16x8 fw2[TAIL_OK && EB == 2 ? T2W : 1][TAIL_OK && EB == 2 ? K2S : 1];
        auto load_fw2 = [&]() {
            if constexpr (TAIL_OK && EB == 2)
                if (a.w2 && is_consumer)
#pragma unroll
                    for (int t = 0; t < T2W; ++t)
#pragma unroll
                        for (int kk = 0; kk < K2S; ++kk)
                            fw2[t][kk] = *(const bf16x8 *)((const bf16_t *)a.w2f + ((size_t)((t2g * T2W + t) * K2S + kk) * 64 + lane) * 8);
        };
        load_fw2();
        if (out8) {
            // e4m3 output: the bf16-rounded value times 1/scale, RNE, saturating (shortcut: see below)
#pragma unroll
            for (int it = 0; it < NIT8; ++it) {
                int row, cc;
                const unsigned off = piece_off(tid + it * NT, CPR8, 16, out_sb, row, cc);
                if (off == OOB_OFFSET) continue;
                const uint4 o0 = *(const uint4 *)(smem + row * RS + cc * 32), o1 = *(const uint4 *)(smem + row * RS + cc * 32 + 16);
                const uint32_t ow[8] = {o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w};
                float v[16];
#pragma unroll
                for (int q = 0; q < 8; ++q) { v[2 * q] = bf16_bits_to_f32(ow[q] & 0xffff); v[2 * q + 1] = bf16_bits_to_f32(ow[q] >> 16); }
                if (res) {
                    // as if the shortcut ran as its own kernel (ew_ops k_add): this conv's output is first quantised
                    // with its own scale, then (x * s_x + r * s_r) is formed with separately rounded operations
                    const u32x4_t r = rpre[it];
                    const int rw[4] = {(int)r[0], (int)r[1], (int)r[2], (int)r[3]};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int xw = (int)f32x2_to_fp8<true>(v[4 * q + 2] * a.mid_inv_scale, v[4 * q + 3] * a.mid_inv_scale,
                                                               f32x2_to_fp8<false>(v[4 * q] * a.mid_inv_scale, v[4 * q + 1] * a.mid_inv_scale, 0));
                        v[4 * q + 0] = __fadd_rn(__fmul_rn(__builtin_amdgcn_cvt_f32_fp8(xw, 0), a.mid_scale), __fmul_rn(__builtin_amdgcn_cvt_f32_fp8(rw[q], 0), a.res_scale));
                        v[4 * q + 1] = __fadd_rn(__fmul_rn(__builtin_amdgcn_cvt_f32_fp8(xw, 1), a.mid_scale), __fmul_rn(__builtin_amdgcn_cvt_f32_fp8(rw[q], 1), a.res_scale));
                        v[4 * q + 2] = __fadd_rn(__fmul_rn(__builtin_amdgcn_cvt_f32_fp8(xw, 2), a.mid_scale), __fmul_rn(__builtin_amdgcn_cvt_f32_fp8(rw[q], 2), a.res_scale));
                        v[4 * q + 3] = __fadd_rn(__fmul_rn(__builtin_amdgcn_cvt_f32_fp8(xw, 3), a.mid_scale), __fmul_rn(__builtin_amdgcn_cvt_f32_fp8(rw[q], 3), a.res_scale));
                    }
                }
                uint32_t pw[4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    pw[q] = f32x2_to_fp8<true>(v[4 * q + 2] * a.out_inv_scale, v[4 * q + 3] * a.out_inv_scale,
                                               f32x2_to_fp8<false>(v[4 * q] * a.out_inv_scale, v[4 * q + 1] * a.out_inv_scale, 0));
                __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{pw[0], pw[1], pw[2], pw[3]}, rs_out, off, 0, OUT_STORE_AUX);
                if (TAIL_OK && EB == 1) if (a.w2) *(uint4 *)(smem + BP * RS + row * (BC + 16) + cc * 16) = uint4{pw[0], pw[1], pw[2], pw[3]};
            }
            if constexpr (TAIL_OK && EB == 1) if (a.w2) {
                // ---- fused 1x1 tail, e4m3 form: the codes just stored are also kept in LDS ([BP][BC] bytes, pitch BC + 16);
                //      wave w owns tail channels 16w..16w+15; K = BC codes in 128-wide steps on the fp8 MFMA with the same
                //      chunk assignment (lq, lq + 4) as the main loop; epilogue = the stand-alone fp8 kernel's: acc * osc + b,
                //      leaky, bf16 rounding, e4m3(v / scale).  Bit-identical to the separate launch. ----
                constexpr int C2 = BC / 2, K2S = BC / 128, RSC = BC + 16, RS2 = C2 + 16;
                const char *const codes = smem + BP * RS;
                char *const st2 = smem;                            // the bf16 tile is dead once the store loop has run
                i32x8 fw2q[K2S];
                if (is_consumer)
#pragma unroll
                    for (int kk = 0; kk < K2S; ++kk) {
                        const char *wr = (const char *)a.w2 + (size_t)(wave_id * 16 + l15) * a.K2pad + kk * 128;
                        const uint4 lo = *(const uint4 *)(wr + lq * 16), hi = *(const uint4 *)(wr + (lq + 4) * 16);
                        fw2q[kk] = i32x8{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
                    }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                block_barrier();
                if (is_consumer) {
                    const float4 b2v = *(const float4 *)(a.b2 + wave_id * 16 + lq * 4);
                    const float4 s2v = *(const float4 *)(a.oscale2 + wave_id * 16 + lq * 4);
#pragma unroll 1
                    for (int j = 0; j < TP; ++j) {
                        f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int kk = 0; kk < K2S; ++kk) {
                            const char *xr = codes + (j * 16 + l15) * RSC + kk * 128;
                            const uint4 lo = *(const uint4 *)(xr + lq * 16), hi = *(const uint4 *)(xr + (lq + 4) * 16);
                            const i32x8 x = i32x8{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
                            acc2 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fw2q[kk], x, acc2, 0, 0, 0, 0, 0, 0);
                        }
                        float v[4] = {acc2[0] * s2v.x + b2v.x, acc2[1] * s2v.y + b2v.y, acc2[2] * s2v.z + b2v.z, acc2[3] * s2v.w + b2v.w};
                        if (a.act2 == ACT_LEAKY) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.1f * v[q]);
                        }
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] = bf16_bits_to_f32(f32_to_bf16_rn(v[q])) * a.out2_inv_scale;
                        const uint32_t w8 = f32x2_to_fp8<true>(v[2], v[3], f32x2_to_fp8<false>(v[0], v[1], 0));
                        *(uint32_t *)(st2 + (j * 16 + l15) * RS2 + wave_id * 16 + lq * 4) = w8;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                block_barrier();
                constexpr int CPR2 = C2 / 16;
                const __amdgpu_buffer_rsrc_t rs_out2 = tile_rsrc((char *)a.out2 + m0 * a.out2_stride);
#pragma unroll
                for (int it = 0; it < (BP * CPR2 + NT - 1) / NT; ++it) {
                    int row, cc;
                    const unsigned off = piece_off(tid + it * NT, CPR2, 0, (unsigned)a.out2_stride, row, cc);
                    __builtin_amdgcn_raw_buffer_store_b128(*(const u32x4_t *)(st2 + row * RS2 + cc * 16), rs_out2, off, 0, OUT_STORE_AUX);
                }
            }
        } else {
            // The loop is written out with and without the shortcut: with that run-time condition inside it, the compiler's s_waitcnt
            // vmcnt for a shortcut piece is the minimum over the merged paths (vmcnt(5)), which also waits for the stores of the previous
            // pieces to COMPLETE -- five memory operations in flight per wave instead of all of them.  (The tail's write-back stays a
            // run-time branch: an LDS store does not touch vmcnt, and a third copy of the loop costs registers.)
            const bool tail_wb = TAIL_OK && EB == 2 && a.w2;
            // a head riding as the tail is this conv's ONLY reader (planner): its own tensor is never stored (wave-uniform)
            const bool skip_out = HEAD_TAIL_OK && a.tail_f32 && a.w2;
            auto store_tile = [&](auto resc) {
                constexpr bool RES = decltype(resc)::value;
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    int row, cc;
                    const unsigned off = piece_off(tid + it * NT, CPR, 8, out_sb, row, cc);
                    if ((BP * CPR) % NT != 0 && tid + it * NT >= BP * CPR) continue;       // (only shapes whose piece count is ragged)
                    u32x4_t o = *(const u32x4_t *)(smem + row * RS + cc * 16);
                    if constexpr (RES) {
                        // the layer's own output was rounded to bf16 above, exactly as if it had been stored and re-read
                        // by a separate shortcut kernel; the sum is rounded once more
                        const u32x4_t r = rpre[it];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float lo = unpack16_lo<H16>(o[q]) + unpack16_lo<H16>(r[q]);
                            const float hi = unpack16_hi<H16>(o[q]) + unpack16_hi<H16>(r[q]);
                            o[q] = pack16x2<H16>(lo, hi);
                        }
                        if (tail_wb) *(u32x4_t *)(smem + row * RS + cc * 16) = o;     // the tail consumes the summed tile
                    }
                    if (!skip_out) __builtin_amdgcn_raw_buffer_store_b128(o, rs_out, off, 0, OUT_STORE_AUX);
                }
            };
            if (res) store_tile(std::true_type{}); else store_tile(std::false_type{});
            if (DIAG) te3 = stamp();
            if constexpr (TAIL_OK && EB == 2) if (a.w2) {
                // ---- fused 1x1 tail: out2[pixel][C2] = act2(W2 . tile[pixel][0..BC) + b2) on the finished tile in LDS.  Wave (t2g, t2p)
                //      owns output channels 32 t2g .. 32 t2g + 31 for the pixel sub-tiles of part t2p: every pixel fragment it reads
                //      feeds two MFMAs (the tail is LDS-read-bound: one fragment per MFMA in the one-channel-tile-per-wave form cost
                //      twice the LDS traffic).  K is walked in ascending 32-wide steps, the order of the stand-alone 1x1 kernel, so
                //      the result is bit-identical to the unfused layer. ----
                constexpr int RS2 = C2 * 2 + 16;
                char *const st2 = smem + BP * RS;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                block_barrier();                                   // tile (with the shortcut added) complete in LDS
                if (DIAG) te4 = stamp();
                if (is_consumer) {
                    const float slope2 = a.act2 == ACT_LEAKY ? 0.1f : 1.0f;
                    f32x4 b2v[T2W];
#pragma unroll
                    for (int t = 0; t < T2W; ++t) b2v[t] = *(const f32x4 *)(a.b2 + (t2g * T2W + t) * 16 + lq * 4);
                    // head mode: raster row r = j * 16 + l15 of the block -> its pixel; this lane's four channels and whether one of them is a box's objectness logit
                    const int hch = (t2g * T2W) * 16 + lq * 4;
                    auto finish = [&](const f32x4 &acc2, int j, int t) {
                        f32x4 v = acc2 + b2v[t];
                        if constexpr (HEAD_TAIL_OK) if (head_tail) {
                            // (exactly the stand-alone head's arithmetic: fp32 accumulator + bias, linear; same K order -> the same bits)
                            const unsigned r = (unsigned)(j * 16 + l15), y = __umul24(r, halo_div_mul(BW)) >> 16, x = r - __umul24(y, (unsigned)BW);
                            if (y < ylim && x < xlim) {
                                const size_t m = m0 + (size_t)y * a.W + x;
                                const int ch = hch + t * 16;
                                float *o = (float *)a.out2 + m * a.out2_stride + ch;
                                if (a.obj_out) {
#pragma unroll
                                    for (int q = 0; q < 4; ++q) {
                                        const int an = fast_div(ch + q, a.obj_mul, a.obj_shift);
                                        if (ch + q - an * a.obj_attrs == 4 && ch + q < a.C2out) a.obj_out[m * a.obj_na + an] = v[q];
                                    }
                                }
                                if (ch + 3 < a.C2out) *(float4 *)o = float4{v[0], v[1], v[2], v[3]};
                                else
                                    for (int q = 0; q < 4; ++q) if (ch + q < a.C2out) o[q] = v[q];
                            }
                            return;
                        }
                        const f32x4 u = v * slope2;
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] = vmax_f32(v[q], u[q]);
                        *(uint2 *)(st2 + (j * 16 + l15) * RS2 + ((t2g * T2W + t) * 16 + lq * 4) * 2) = uint2{pack16x2<H16>(v[0], v[1]), pack16x2<H16>(v[2], v[3])};
                    };
                    const int j0 = head_tail ? 0 : (TP * t2p) / T2P, j1 = head_tail ? TP : (TP * (t2p + 1)) / T2P;       // this wave's pixel sub-tiles
                    // Two sub-tiles at a time: 2 T2W independent accumulation chains, each K-ordered; the loop stays rolled so the fragment
                    // reads are not all hoisted.  (Reading the next pair's fragments under the MFMAs of the current one -- a second
                    // 64-register buffer -- measured slower in the stamped build, 5500 vs 4650 cycles.)
                    {
                        int j = j0;
#pragma unroll 1
                        for (; j + 1 < j1; j += 2) {
                            f32x4 ca[T2W], cb[T2W];
#pragma unroll
                            for (int t = 0; t < T2W; ++t) ca[t] = cb[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                            for (int kk = 0; kk < K2S; ++kk) {
                                const bf16x8 xa = *(const bf16x8 *)(smem + (j * 16 + l15) * RS + (kk * 4 + lq) * 16);
                                const bf16x8 xb = *(const bf16x8 *)(smem + ((j + 1) * 16 + l15) * RS + (kk * 4 + lq) * 16);
#pragma unroll
                                for (int t = 0; t < T2W; ++t) {
                                    ca[t] = mma16<H16>(fw2[t][kk], xa, ca[t]);
                                    cb[t] = mma16<H16>(fw2[t][kk], xb, cb[t]);
                                }
                            }
#pragma unroll
                            for (int t = 0; t < T2W; ++t) { finish(ca[t], j, t); finish(cb[t], j + 1, t); }
                        }
                        if (j < j1) {
                            f32x4 cc_[T2W];
#pragma unroll
                            for (int t = 0; t < T2W; ++t) cc_[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                            for (int kk = 0; kk < K2S; ++kk) {
                                const bf16x8 x = *(const bf16x8 *)(smem + (j * 16 + l15) * RS + (kk * 4 + lq) * 16);
#pragma unroll
                                for (int t = 0; t < T2W; ++t) cc_[t] = mma16<H16>(fw2[t][kk], x, cc_[t]);
                            }
#pragma unroll
                            for (int t = 0; t < T2W; ++t) finish(cc_[t], j, t);
                        }
                    }
                }
                if (DIAG) te5 = stamp();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                block_barrier();
                if (DIAG) te6 = stamp();
                constexpr int CPR2 = C2 / 8;
                const __amdgpu_buffer_rsrc_t rs_out2 = tile_rsrc((char *)a.out2 + m0 * a.out2_stride * 2);
                if (!head_tail)          // (a head left from the registers above)
#pragma unroll
                for (int it = 0; it < (BP * CPR2 + NT - 1) / NT; ++it) {
                    int row, cc;
                    const unsigned off = piece_off(tid + it * NT, CPR2, 0, (unsigned)a.out2_stride * 2u, row, cc);
                    __builtin_amdgcn_raw_buffer_store_b128(*(const u32x4_t *)(st2 + row * RS2 + cc * 16), rs_out2, off, 0, OUT_STORE_AUX);
                }
            }
        }
    } else if (is_consumer && !HALO) {
        // fp32 output (detection heads, Cout = 255): 4 consecutive channels per lane, 16-B stores.  PLAIN stores, unlike every other output
        // (OUT_STORE_AUX): a lane writes 16 bytes of its own pixel's 1 KB row, 64 different rows per instruction -- write-through sends each
        // piece to memory alone, while a plain store lets L2 assemble whole lines first (round 5, same tile: 33 -> 69 us on the 52 x 52 head,
        // 17 -> 26 us on the 26 x 26 one with sc1)
#pragma unroll
        for (int i = 0; i < TC; ++i) {
            const int ch = ct * BC + (wci * TC + i) * 16 + lq * 4;
            if (ch >= a.Cout) continue;
            const float4 bv = *(const float4 *)(a.bias + ch);
            float4 sv = float4{1.f, 1.f, 1.f, 1.f};
            if (EB == 1 && a.oscale) sv = *(const float4 *)(a.oscale + ch);
            // [yolo] head: is one of this lane's four channels a box's objectness logit (channel an * (5 + classes) + 4)?  Those are also
            // written to a compact plane [pixel][anchor] -- the decode's objectness pre-filter then reads 12 bytes per cell in one
            // coalesced stream instead of one 128-byte line per box out of this tensor
            int oq = -1, oan = 0;
            if (a.obj_out) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int an = fast_div(ch + q, a.obj_mul, a.obj_shift);
                    if (ch + q - an * a.obj_attrs == 4 && ch + q < a.Cout) { oq = q; oan = an; }
                }
            }
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                const int m = pt * BP + (wpi * TP + j) * 16 + l15;
                if (m >= M) continue;
                float v[4];
                if (EB == 1) { v[0] = acc[i][j][0] * sv.x + bv.x; v[1] = acc[i][j][1] * sv.y + bv.y; v[2] = acc[i][j][2] * sv.z + bv.z; v[3] = acc[i][j][3] * sv.w + bv.w; }
                else { v[0] = acc[i][j][0] + bv.x; v[1] = acc[i][j][1] + bv.y; v[2] = acc[i][j][2] + bv.z; v[3] = acc[i][j][3] + bv.w; }
                if (a.act == ACT_LEAKY) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.1f * v[q]);     // == v > 0 ? v : 0.1 v
                }
                float *o = (float *)a.out + (size_t)m * a.out_stride + ch;
                if (oq >= 0) a.obj_out[(size_t)m * a.obj_na + oan] = oq == 0 ? v[0] : oq == 1 ? v[1] : oq == 2 ? v[2] : v[3];
                if (ch + 3 < a.Cout) *(float4 *)o = float4{v[0], v[1], v[2], v[3]};
                else
                    for (int q = 0; q < 4; ++q) if (ch + q < a.Cout) o[q] = v[q];
            }
        }
    }
    if (DIAG && a.dbg && lane == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t_end = stamp_loop();
        unsigned long long *d = a.dbg + ((size_t)tile * NTOT + wave_id) * 16;
        const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
        d[0] = t_wait; d[1] = t_issue; d[2] = t_mma; d[3] = t_loop_end - t_all0; d[4] = t_end - t_loop_end;
        d[6] = t_all0 - t_top; d[7] = t_first - t_all0; d[8] = te1 - t_loop_end; d[9] = te2 - te1; d[10] = te3 - te2; d[11] = t_end - (te6 ? te6 : te3);
        d[12] = te4 ? te4 - te3 : 0; d[13] = te4 ? te5 - te4 : 0; d[14] = te4 ? te6 - te5 : 0;
        d[15] = (t_loop_end - t_all0) * 100ull / (rt_loop - rt0 ? rt_loop - rt0 : 1);      // shader MHz over the K loop alone (s_memtime / s_memrealtime, 100 MHz)
        d[5] = ((unsigned long long)KT << 40) | ((t_end - t_all0) * 100ull / (rt1 - rt0 ? rt1 - rt0 : 1));   // KT | shader MHz (realtime = 100 MHz)
    }
#endif
}

// dynamic LDS of one instantiation: NS staging buffers, re-used by the epilogue's padded output tile
template <int WP, int WC, int TP, int TC, int NS, int BK, int NL = 0, bool HALO = false, int BH = HALO_B, int BW = HALO_B>
constexpr size_t conv_lds_bytes()
{
    constexpr int NW = NL > 0 ? NL : WP * WC, BP = WP * TP * 16, BC = WC * TC * 16;
    constexpr int RG = 64 / (BK * 2 / 16);
    constexpr int LA = ((BP + RG - 1) / RG + NW - 1) / NW, LB = ((BC + RG - 1) / RG + NW - 1) / NW;
    constexpr size_t stage = (size_t)(LA + LB) * NW * RG * (BK * 2);
    constexpr bool tail = WP == 1 && WC == 8 && (BC == 256 || BC == 128); // TAIL_OK shapes also stage the tail's [BP][BC/2] tile
    constexpr size_t lds0 = HALO ? (size_t)2 * halo_apieces(BH, BW) * 1024 + (size_t)NS * LB * NW * RG * (BK * 2) : (size_t)NS * stage, ldso = (size_t)BP * (BC * 2 + 16) + (tail ? (size_t)BP * (BC + 16) : 0);
    return lds0 > ldso ? lds0 : ldso;
}
