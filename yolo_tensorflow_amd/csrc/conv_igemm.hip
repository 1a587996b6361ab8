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

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
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
template <int WP, int WC, int TP, int TC, int NS, int BK, bool UNI, int NL = 0, bool DIAG = false, int EB = 2>
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
    static_assert(EB == 2 || (EB == 1 && BK == 64), "fp8 operands need the 128-B-row form");
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

    extern __shared__ __attribute__((aligned(16))) char smem[];   // [NS][ BP rows | BC rows ][128 B]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
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
    const int tilesP = (M + BP - 1) / BP;
    const int per_xcd = gridDim.x >> 3;
    const int tile = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (tile >= tilesP * tilesC) return;
    const int ct = tile % tilesC;
    const int pt = tile / tilesC;

    // Buffer descriptors.  The activation base is moved back by (W+1) pixels so that the offset of tap (0,0) of a
    // border pixel (one row up, one column left) is still >= 0.
    const int shift = (a.W + 1) * a.in_stride * EB;                  // bytes
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
    unsigned rowoff[LA];                       // byte offset of tap (0,0), channel chunk*8 (UNI) or 0 (per-lane tap)
    unsigned tapmask[LA];                      // bit t set: tap t of this pixel lies inside the image
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const int prow = (wid + i * NW) * RG + rl;
        const int m = pt * BP + prow;
        unsigned mask = 0, off = 0;
        if (m < M && prow < BP) {
            const int n = fast_div(m, a.howo_mul, a.howo_shift);
            const int rem = m - n * HoWo;
            const int oy = fast_div(rem, a.wo_mul, a.wo_shift);
            const int ox = rem - oy * a.Wo;
            const int iy0 = oy * a.stride - a.pad, ix0 = ox * a.stride - a.pad;
            off = (unsigned)(((n * a.H + iy0) * a.W + ix0 + a.W + 1) * a.in_stride + (UNI ? chunk * EPC : 0)) * (unsigned)EB;
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
        rowoff[i] = off;
        tapmask[i] = mask;
    }
    unsigned woff[LB > 0 ? LB : 1];
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        const int crow = (wid + i * NW) * RG + rl;
        woff[i] = crow < BC ? (unsigned)((ct * BC + crow) * a.Kpad + chunk * EPC) * (unsigned)EB : OOB_OFFSET;
    }

    // K cursor.  UNI: scalars (tap, kh, kw, channel base).  Otherwise per lane (chunk-dependent).
    int s_tap = 0, s_kh = 0, s_kw = 0, s_kb = 0, s_cb = 0;   // UNI: tap, its (kh, kw), channel offset in the chunk, chunk base
    int v_kc = chunk * EPC, v_tap = 0;                  // !UNI
    if (!UNI)
        while (v_kc >= a.Cin_pad) { v_kc -= a.Cin_pad; ++v_tap; }
    int s_wk = 0;                                       // byte offset of the K-step in a filter row
    auto stage = [&](char *sbase) {
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
#pragma unroll
        for (int i = 0; i < LB; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void *)(dw + i * NW * 1024), 16, woff[i], s_wk, 0, 0);
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
    constexpr bool TAIL_OK = !DIAG && WP == 1 && NC == 8 && NC * 16 == BC / 2;
    // (Tried and dropped: placing one LDS-DMA of the next stage behind every MFMA group with sched_group_barrier instead of
    // issuing the whole stage first.  A/B on one MI355X box, YOLOv3-416 batch 32: 2 % SLOWER in both bf16 (3.48 vs 3.40 ms)
    // and fp8 (2.39 vs 2.34 ms) -- a DMA blocks its wave's issue for ~60 cycles wherever it is placed, and the MFMA pipe
    // holds no queue to ride it out.)
    constexpr int D = NS - 1;                  // prefetch distance in K-steps
#pragma unroll
    for (int t = 0; t < D; ++t)
        if (t < KT && is_loader) stage(smem + t * STAGE_BYTES);

    const int l15 = lane & 15, lq = lane >> 4;
    // fragment read offsets inside a stage (two K-halves), constant over the loop
    const int sw0 = BK == 64 ? ((0 + lq) ^ (l15 & 7)) << 4 : (lq ^ (3 * ((l15 >> 2) & 1))) << 4;
    const int sw1 = ((4 + lq) ^ (l15 & 7)) << 4;            // second K-half (BK = 64 only)
    const int offx = (wpi * TP * 16 + l15) * RB;
    const int offw = BPL * RB + (wci * TC * 16 + l15) * RB;
    int cur = 0, nxt = D % NS;                 // stage being multiplied / stage being filled
    // DIAG (separate diagnostic instantiation, never the shipped kernel): s_memtime stamps around the phases of a K-step
    unsigned long long t_wait = 0, t_issue = 0, t_mma = 0, t_all0 = 0;
    auto stamp = [&]() -> unsigned long long {
        unsigned long long t = 0;
        if (DIAG) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); __builtin_amdgcn_sched_barrier(0); }
        return t;
    };
    unsigned long long rt0 = 0;
    if (DIAG) { t_all0 = stamp(); rt0 = __builtin_amdgcn_s_memrealtime(); }
    // one K-step.  `fill` / `sb`: the stage being filled and the stage being multiplied.  They are distinct __restrict__ parameters
    // on purpose: hipcc orders every ds_read behind ALL outstanding LDS-DMA (s_waitcnt vmcnt(0) before the first
    // fragment read of each K-step, i.e. the loads just issued for the NEXT step were waited for before this step's
    // MFMAs) unless alias-scope metadata proves the DMA target and the read are different memory; restrict parameters
    // of an inlined function are what produces that metadata.
    auto kstep = [&](int kt, char *__restrict__ fill, const char *__restrict__ sb, const bool LOAD) {
        const unsigned long long s0 = stamp();
        // K-step kt has landed once at most (D-1) younger K-steps' loads remain outstanding (in-order counter)
        if (is_loader) { if (LOAD) wait_vmcnt<(D - 1) * L>(); else wait_vmcnt<0>(); }
        block_barrier();                       // everybody's part of K-step kt is in LDS; stage `nxt` is free again
        const unsigned long long s1 = stamp();
        if (LOAD && is_loader) stage(fill);
        const unsigned long long s2 = stamp();
        __builtin_amdgcn_s_setprio(2);          // MFMA phase: win issue arbitration against the co-resident workgroup's DMA / epilogue phases
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
                auto frag = [&](const char *row) -> i32x8 {
                    const uint4 lo = *(const uint4 *)(row + sw0), hi = *(const uint4 *)(row + sw1);
                    return i32x8{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
                };
#pragma unroll
                for (int i = 0; i < TC; ++i) fw[i] = frag(sb + offw + i * 16 * RB);
#pragma unroll
                for (int j = 0; j < PD; ++j) fx[j] = frag(sb + offx + j * 16 * RB);
#pragma unroll
                for (int j = 0; j < TP; ++j) {
                    if (j + PD < TP) fx[j + PD] = frag(sb + offx + (j + PD) * 16 * RB);
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
        } else if (is_consumer) {
            constexpr int KH = BK / 32;                 // K halves of 32 per step
            constexpr int NG = KH * TP;                 // MFMA groups per step
            constexpr int PD = NG < 4 ? NG : 4;
            bf16x8 fw[KH][TC], fx[NG];
            auto rdw = [&](int kk) {
#pragma unroll
                for (int i = 0; i < TC; ++i) fw[kk][i] = *(const bf16x8 *)(sb + offw + i * 16 * RB + (kk ? sw1 : sw0));
            };
            auto rdx = [&](int g) {
                const int kk = g / TP, j = g - kk * TP;
                fx[g] = *(const bf16x8 *)(sb + offx + j * 16 * RB + (kk ? sw1 : sw0));
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
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[kk][i], fx[g], acc[i][j], 0, 0, 0);
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
        __builtin_amdgcn_s_setprio(0);
        if (DIAG) {
            asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
            const unsigned long long s3 = stamp();
            t_wait += s1 - s0; t_issue += s2 - s1; t_mma += s3 - s2;
        }
    };
    {
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
    unsigned long long t_loop_end = 0;
    if (DIAG) t_loop_end = stamp();

    // ---- epilogue ----
    const bool full = (pt * BP + BP <= M) && (ct * BC + BC <= a.Cout);     // no ragged edge in this tile
    if (a.out_dt != DT_F32) {
        // bf16 / fp8 output: scale + bias + activation in registers, then the tile goes through LDS (as bf16) so that
        // global stores (and the residual loads) are 16 B per lane along the channel axis -- whole 128-B lines per
        // pixel instead of 16 scattered 32-B pieces per store instruction (row-per-lane dwordx2 stores are
        // issue-bound: ~600 cycles each).  A 16-B piece is 8 bf16 or 16 fp8 channels.
        constexpr int RS = BC * 2 + 16;                       // padded LDS row (bytes)
        constexpr int NT = 64 * NTOT;
        constexpr int CPR = BC / 8, NIT = (BP * CPR + NT - 1) / NT;       // bf16 pieces per row / per thread
        constexpr int CPR8 = BC / 16, NIT8 = (BP * CPR8 + NT - 1) / NT;   // fp8 pieces
        const bool out8 = a.out_dt == DT_FP8;
        // residual (shortcut source) pieces are fetched now, all at once, so that their latency is covered by the
        // accumulator -> LDS pass below instead of being paid once per piece in the store loop
        const char *__restrict__ res = (const char *)a.res;
        uint4 rpre[NIT];
        if (res) {
            if (out8) {
#pragma unroll
                for (int it = 0; it < NIT8; ++it) {
                    const int c = tid + it * NT;
                    const int row = c / CPR8, cc = c - row * CPR8;
                    const int m = pt * BP + row, ch = ct * BC + cc * 16;
                    rpre[it] = (c < BP * CPR8 && m < M && ch < a.Cout) ? *(const uint4 *)(res + (size_t)m * a.res_stride + ch)
                                                                       : uint4{0, 0, 0, 0};
                }
            } else {
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int c = tid + it * NT;
                    const int row = c / CPR, cc = c - row * CPR;
                    const int m = pt * BP + row, ch = ct * BC + cc * 8;
                    rpre[it] = (c < BP * CPR && m < M && ch < a.Cout) ? *(const uint4 *)(res + ((size_t)m * a.res_stride + ch) * 2)
                                                                      : uint4{0, 0, 0, 0};
                }
            }
        }
        // fused 1x1 tail: its filter fragments (16 rows x BC per consumer wave) are fetched here when the registers allow
        // (8-wave shapes), so their latency hides behind the epilogue; the 4-wave shapes fetch them just before use
        constexpr bool TAIL_EARLY = TAIL_OK && NL == 0 && EB == 2;   // the 12-wave role-split shape has no registers to spare
        bf16x8 fw2[TAIL_OK && EB == 2 ? BC / 32 : 1];
        if (TAIL_EARLY) if (a.w2 && is_consumer)
#pragma unroll
            for (int kk = 0; kk < BC / 32; ++kk)
                fw2[kk] = *(const bf16x8 *)((const bf16_t *)a.w2 + (size_t)(wave_id * 16 + l15) * a.K2pad + (kk * 4 + lq) * 8);
        block_barrier();                                      // every wave is done reading the last stage
        if (is_consumer)
#pragma unroll
        for (int i = 0; i < TC; ++i) {
            const int chl = (wci * TC + i) * 16 + lq * 4;     // channel within the tile
            const float4 bv = *(const float4 *)(a.bias + ct * BC + chl);
            float4 sv = float4{1.f, 1.f, 1.f, 1.f};
            if (EB == 1 && a.oscale) sv = *(const float4 *)(a.oscale + ct * BC + chl);
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                float v[4];
                if (EB == 1) { v[0] = acc[i][j][0] * sv.x + bv.x; v[1] = acc[i][j][1] * sv.y + bv.y; v[2] = acc[i][j][2] * sv.z + bv.z; v[3] = acc[i][j][3] * sv.w + bv.w; }
                else { v[0] = acc[i][j][0] + bv.x; v[1] = acc[i][j][1] + bv.y; v[2] = acc[i][j][2] + bv.z; v[3] = acc[i][j][3] + bv.w; }
                if (a.act == ACT_LEAKY) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.1f * v[q]);     // == v > 0 ? v : 0.1 v
                }
                uint2 pk;
                pk.x = f32_to_bf16_rn(v[0]) | (f32_to_bf16_rn(v[1]) << 16);
                pk.y = f32_to_bf16_rn(v[2]) | (f32_to_bf16_rn(v[3]) << 16);
                *(uint2 *)(smem + ((wpi * TP + j) * 16 + l15) * RS + chl * 2) = pk;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        block_barrier();
        if (out8) {
            // e4m3 output: the bf16-rounded value times 1/scale, RNE, saturating (shortcut: see below)
#pragma unroll
            for (int it = 0; it < NIT8; ++it) {
                const int c = tid + it * NT;
                const int row = c / CPR8, cc = c - row * CPR8;
                const int m = pt * BP + row, ch = ct * BC + cc * 16;
                if (c >= BP * CPR8 || (!full && (m >= M || ch >= a.Cout))) continue;
                const uint4 o0 = *(const uint4 *)(smem + row * RS + cc * 32), o1 = *(const uint4 *)(smem + row * RS + cc * 32 + 16);
                const uint32_t ow[8] = {o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w};
                float v[16];
#pragma unroll
                for (int q = 0; q < 8; ++q) { v[2 * q] = bf16_bits_to_f32(ow[q] & 0xffff); v[2 * q + 1] = bf16_bits_to_f32(ow[q] >> 16); }
                if (res) {
                    // as if the shortcut ran as its own kernel (ew_ops k_add): this conv's output is first quantised
                    // with its own scale, then (x * s_x + r * s_r) is formed with separately rounded operations
                    const uint4 r = rpre[it];
                    const int rw[4] = {(int)r.x, (int)r.y, (int)r.z, (int)r.w};
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
                *(uint4 *)((char *)a.out + (size_t)m * a.out_stride + ch) = uint4{pw[0], pw[1], pw[2], pw[3]};
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
                for (int c = tid; c < BP * CPR2; c += NT) {
                    const int row = c / CPR2, cc = c - row * CPR2;
                    const int m = pt * BP + row;
                    if (m < M) *(uint4 *)((char *)a.out2 + (size_t)m * a.out2_stride + cc * 16) = *(const uint4 *)(st2 + row * RS2 + cc * 16);
                }
            }
        } else {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int c = tid + it * NT;
                const int row = c / CPR, cc = c - row * CPR;
                const int m = pt * BP + row, ch = ct * BC + cc * 8;
                if (c >= BP * CPR || (!full && (m >= M || ch >= a.Cout))) continue;
                uint4 o = *(const uint4 *)(smem + row * RS + cc * 16);
                if (res) {
                    // the layer's own output was rounded to bf16 above, exactly as if it had been stored and re-read
                    // by a separate shortcut kernel; the sum is rounded once more
                    const uint4 r = rpre[it];
                    uint32_t ov[4] = {o.x, o.y, o.z, o.w}, rv[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float lo = bf16_bits_to_f32(ov[q] & 0xffff) + bf16_bits_to_f32(rv[q] & 0xffff);
                        const float hi = bf16_bits_to_f32(ov[q] >> 16) + bf16_bits_to_f32(rv[q] >> 16);
                        ov[q] = f32_to_bf16_rn(lo) | (f32_to_bf16_rn(hi) << 16);
                    }
                    o = uint4{ov[0], ov[1], ov[2], ov[3]};
                    if (TAIL_OK && EB == 2 && a.w2) *(uint4 *)(smem + row * RS + cc * 16) = o;     // the tail consumes the summed tile
                }
                *(uint4 *)((bf16_t *)a.out + (size_t)m * a.out_stride + ch) = o;
            }
            if constexpr (TAIL_OK && EB == 2) if (a.w2) {
                // ---- fused 1x1 tail: out2[pixel][C2] = act2(W2 . tile[pixel][0..BC) + b2) on the finished tile in LDS.
                //      Consumer wave w owns output channels 16w..16w+15 for every pixel of the tile; its filter fragments
                //      (16 rows x BC, 8 KB) come straight from global; K is walked in ascending 32-wide steps, the order of
                //      the stand-alone 1x1 kernel, so the result is bit-identical to the unfused layer. ----
                constexpr int C2 = BC / 2, K2S = BC / 32, RS2 = C2 * 2 + 16;
                char *const st2 = smem + BP * RS;
                if (!TAIL_EARLY && is_consumer)
#pragma unroll
                    for (int kk = 0; kk < K2S; ++kk)
                        fw2[kk] = *(const bf16x8 *)((const bf16_t *)a.w2 + (size_t)(wave_id * 16 + l15) * a.K2pad + (kk * 4 + lq) * 8);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                block_barrier();                                   // tile (with the shortcut added) complete in LDS
                if (is_consumer) {
                    const float4 b2v = *(const float4 *)(a.b2 + wave_id * 16 + lq * 4);
                    auto finish = [&](const f32x4 &acc2, int j) {
                        float v[4] = {acc2[0] + b2v.x, acc2[1] + b2v.y, acc2[2] + b2v.z, acc2[3] + b2v.w};
                        if (a.act2 == ACT_LEAKY) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.1f * v[q]);
                        }
                        uint2 pk;
                        pk.x = f32_to_bf16_rn(v[0]) | (f32_to_bf16_rn(v[1]) << 16);
                        pk.y = f32_to_bf16_rn(v[2]) | (f32_to_bf16_rn(v[3]) << 16);
                        *(uint2 *)(st2 + (j * 16 + l15) * RS2 + (wave_id * 16 + lq * 4) * 2) = pk;
                    };
                    // two pixel tiles at a time: two independent accumulation chains keep the matrix pipe fed (each chain
                    // is K-ordered); the loop stays rolled so the fragment reads are not all hoisted (176 VGPRs otherwise)
#pragma unroll 1
                    for (int j = 0; j + 1 < TP; j += 2) {
                        f32x4 acc2a = {0.f, 0.f, 0.f, 0.f}, acc2b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int kk = 0; kk < K2S; ++kk) {
                            const bf16x8 xa = *(const bf16x8 *)(smem + (j * 16 + l15) * RS + (kk * 4 + lq) * 16);
                            const bf16x8 xb = *(const bf16x8 *)(smem + ((j + 1) * 16 + l15) * RS + (kk * 4 + lq) * 16);
                            acc2a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw2[kk], xa, acc2a, 0, 0, 0);
                            acc2b = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw2[kk], xb, acc2b, 0, 0, 0);
                        }
                        finish(acc2a, j); finish(acc2b, j + 1);
                    }
                    if (TP & 1) {
                        f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int kk = 0; kk < K2S; ++kk) {
                            const bf16x8 x = *(const bf16x8 *)(smem + ((TP - 1) * 16 + l15) * RS + (kk * 4 + lq) * 16);
                            acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw2[kk], x, acc2, 0, 0, 0);
                        }
                        finish(acc2, TP - 1);
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                block_barrier();
                constexpr int CPR2 = C2 / 8;
                for (int c = tid; c < BP * CPR2; c += NT) {
                    const int row = c / CPR2, cc = c - row * CPR2;
                    const int m = pt * BP + row;
                    if (m < M) *(uint4 *)((bf16_t *)a.out2 + (size_t)m * a.out2_stride + cc * 8) = *(const uint4 *)(st2 + row * RS2 + cc * 16);
                }
            }
        }
    } else if (is_consumer) {
        // fp32 output (detection heads, Cout = 255): 4 consecutive channels per lane, 16-B stores
#pragma unroll
        for (int i = 0; i < TC; ++i) {
            const int ch = ct * BC + (wci * TC + i) * 16 + lq * 4;
            if (ch >= a.Cout) continue;
            const float4 bv = *(const float4 *)(a.bias + ch);
            float4 sv = float4{1.f, 1.f, 1.f, 1.f};
            if (EB == 1 && a.oscale) sv = *(const float4 *)(a.oscale + ch);
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
                if (ch + 3 < a.Cout) *(float4 *)o = float4{v[0], v[1], v[2], v[3]};
                else
                    for (int q = 0; q < 4; ++q) if (ch + q < a.Cout) o[q] = v[q];
            }
        }
    }
    if (DIAG && a.dbg && lane == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t_end = stamp();
        unsigned long long *d = a.dbg + ((size_t)tile * NTOT + wave_id) * 6;
        const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
        d[0] = t_wait; d[1] = t_issue; d[2] = t_mma; d[3] = t_loop_end - t_all0; d[4] = t_end - t_loop_end;
        d[5] = ((unsigned long long)KT << 40) | ((t_end - t_all0) * 100ull / (rt1 - rt0 ? rt1 - rt0 : 1));   // KT | shader MHz (realtime = 100 MHz)
    }
#endif
}

// dynamic LDS of one instantiation: NS staging buffers, re-used by the epilogue's padded output tile
template <int WP, int WC, int TP, int TC, int NS, int BK, int NL = 0>
constexpr size_t conv_lds_bytes()
{
    constexpr int NW = NL > 0 ? NL : WP * WC, BP = WP * TP * 16, BC = WC * TC * 16;
    constexpr int RG = 64 / (BK * 2 / 16);
    constexpr int LA = ((BP + RG - 1) / RG + NW - 1) / NW, LB = ((BC + RG - 1) / RG + NW - 1) / NW;
    constexpr size_t stage = (size_t)(LA + LB) * NW * RG * (BK * 2);
    constexpr bool tail = WP == 1 && WC == 8 && WC * 16 == BC / 2; // TAIL_OK shapes also stage the tail's [BP][BC/2] tile
    constexpr size_t lds0 = (size_t)NS * stage, ldso = (size_t)BP * (BC * 2 + 16) + (tail ? (size_t)BP * (BC + 16) : 0);
    return lds0 > ldso ? lds0 : ldso;
}

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
    hipLaunchKernelGGL((conv_igemm<WP, WC, TP, TC, NS, BK, true, 0, true, EB>), dim3((unsigned)((tiles + 7) / 8 * 8)), dim3(64 * NW), lds, s, a);
    return hipGetLastError();
}
hipError_t launch_conv_diag(const ConvArgs &a, hipStream_t s) { return a.in_dt == DT_FP8 ? launch_conv_diag_t<1>(a, s) : launch_conv_diag_t<2>(a, s); }

// ---------------------------------------------------------------------------------------------
// First layer (3x3, stride 1, 3 real input channels padded to 8): HBM-bound (it writes N*H*W*Cout bf16), K is only
// 9 taps x 8 channels = 72 (padded to 96).  No LDS: a lane's MFMA B fragment for K-group (kk, lq) is exactly the
// 16-byte channel vector of ONE input pixel (tap kk*4+lq), so it is a single global_load_dwordx4; the filters
// (Cout x 96 bf16) live in registers for the whole wave.  One wave computes 16 pixels x Cout per step.
template <int TC>
__global__ __launch_bounds__(256) void conv_c8_3x3_direct(const ConvArgs a)
{
#if defined(__HIP_DEVICE_COMPILE__)
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
                for (int kk = 0; kk < 3; ++kk) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[i][kk], fx[u][kk], acc[i], 0, 0, 0);
            }
            if (mrow[u] >= 0) {
#pragma unroll
                for (int i = 0; i < TC; ++i) {
                    float v[4] = {acc[i][0] + bv[i].x, acc[i][1] + bv[i].y, acc[i][2] + bv[i].z, acc[i][3] + bv[i].w};
                    if (a.act == ACT_LEAKY)
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.1f * v[q]);     // == v > 0 ? v : 0.1 v
                    uint2 pk;
                    pk.x = f32_to_bf16_rn(v[0]) | (f32_to_bf16_rn(v[1]) << 16);
                    pk.y = f32_to_bf16_rn(v[2]) | (f32_to_bf16_rn(v[3]) << 16);
                    if (a.out_dt == DT_FP8) {
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
    return a.in_dt == DT_BF16 && a.ksize == 3 && a.stride == 1 && a.pad == 1 && a.Cin_pad == 8 && a.out_dt != DT_F32 && !a.res && a.Kpad >= 96 &&
           (a.Cout == 16 || a.Cout == 32 || a.Cout == 64);
}

hipError_t launch_conv_c8_direct(const ConvArgs &a, hipStream_t s)
{
    const long M = (long)a.N * a.Ho * a.Wo;
    long waves = (M + 63) / 64;
    long blocks = (waves + 3) / 4; if (blocks > 256 * 8) blocks = 256 * 8;
    dim3 grid((unsigned)blocks), block(256);
    if (a.Cout == 16) hipLaunchKernelGGL((conv_c8_3x3_direct<1>), grid, block, 0, s, a);
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
    X(33, 1, 4, 11, 2, 3, 64, 0) X(34, 1, 4, 6, 2, 3, 64, 0)

struct CfgDesc { int id, wp, wc, tp, tc, ns, bk, nl; };
#define X(id, wp, wc, tp, tc, ns, bk, nl) {id, wp, wc, tp, tc, ns, bk, nl},
static const CfgDesc kCfgs[] = {CONV_CFGS(X)};
#undef X
int conv_num_cfgs() { return (int)(sizeof(kCfgs) / sizeof(kCfgs[0])); }
bool conv_cfg_tail_ok(int cfg, int cout)
{
    if (cfg < 0 || cfg >= conv_num_cfgs()) return false;
    const CfgDesc &c = kCfgs[cfg];
    const int bc = c.wc * c.tc * 16;
    return c.wp == 1 && c.wc == 8 && c.wc * 16 == bc / 2 && bc == cout;
}
const char *conv_cfg_name(int cfg)
{
    static char names[64][32];
    if (cfg < 0 || cfg >= conv_num_cfgs()) return cfg == CONV_CFG_DIRECT ? "direct_c8" : "?";
    const CfgDesc &c = kCfgs[cfg];
    snprintf(names[cfg], sizeof names[cfg], "p%dc%d_s%d_k%d%s%d", c.wp * c.tp * 16, c.wc * c.tc * 16, c.ns, c.bk, c.nl ? "_L" : "_w", c.nl ? c.nl : c.wp * c.wc);
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

template <int WP, int WC, int TP, int TC, int NS, int BK, int NL, bool UNI, int EB>
static hipError_t launch_u(const ConvArgs &a, hipStream_t s)
{
    constexpr int BP = WP * TP * 16, BC = WC * TC * 16;
    const long M = (long)a.N * a.Ho * a.Wo;
    const long tiles = ((M + BP - 1) / BP) * ((a.Cout + BC - 1) / BC);
    constexpr size_t lds = conv_lds_bytes<WP, WC, TP, TC, NS, BK, NL>();
    dim3 grid((unsigned)((tiles + 7) / 8 * 8)), block(64 * (WP * WC + NL));   // multiple of 8: see the XCD mapping
    if (lds > 65536) {
        hipError_t e = conv_opt_in_lds((const void *)conv_igemm<WP, WC, TP, TC, NS, BK, UNI, NL, false, EB>, lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((conv_igemm<WP, WC, TP, TC, NS, BK, UNI, NL, false, EB>), grid, block, lds, s, a);
    return hipGetLastError();
}

template <int WP, int WC, int TP, int TC, int NS, int BK, int NL, int EB>
static hipError_t launch_t(const ConvArgs &a, hipStream_t s)
{
    // 32-bit buffer offsets: the activation window must stay below 2 GiB
    if (((double)a.N * a.H * a.W * a.in_stride + 2.0 * (a.W + 1) * a.in_stride) * EB >= 2147483648.0) return hipErrorInvalidValue;
    constexpr int BKE = BK * 2 / EB;
    if (a.Kpad % BKE) return hipErrorInvalidValue;
    return (a.Cin_pad % BKE) == 0 ? launch_u<WP, WC, TP, TC, NS, BK, NL, true, EB>(a, s) : launch_u<WP, WC, TP, TC, NS, BK, NL, false, EB>(a, s);
}

hipError_t launch_conv_bf16(const ConvArgs &a, int cfg, hipStream_t s)
{
    if (a.in_dt != DT_BF16) return hipErrorInvalidValue;
    if (cfg == CONV_CFG_DIRECT) return conv_c8_direct_ok(a) ? launch_conv_c8_direct(a, s) : hipErrorInvalidValue;
    switch (cfg) {
#define X(id, wp, wc, tp, tc, ns, bk, nl) case id: return launch_t<wp, wc, tp, tc, ns, bk, nl, 2>(a, s);
        CONV_CFGS(X)
#undef X
    default: return hipErrorInvalidValue;
    }
}

// fp8 operands: the same tile ids, restricted to the two-stage 128-B-row symmetric shapes that the bf16 tuning kept
#define CONV_CFGS_FP8(X)                                                                                     \
    X(0, 2, 2, 4, 4, 2, 64, 0)  X(2, 2, 2, 2, 4, 2, 64, 0)  X(4, 4, 1, 4, 2, 2, 64, 0)  X(6, 2, 2, 4, 2, 2, 64, 0)    \
    X(8, 4, 1, 4, 4, 2, 64, 0)  X(12, 2, 4, 4, 4, 2, 64, 0) X(14, 2, 2, 2, 2, 2, 64, 0) X(16, 1, 4, 11, 2, 2, 64, 0)  \
    X(17, 1, 4, 11, 4, 2, 64, 0) X(19, 1, 4, 10, 2, 2, 64, 0) X(20, 1, 4, 12, 2, 2, 64, 0) X(23, 1, 4, 6, 2, 2, 64, 0) \
    X(32, 1, 8, 11, 2, 2, 64, 0) X(31, 1, 8, 11, 2, 2, 64, 4) X(33, 1, 4, 11, 2, 3, 64, 0) X(34, 1, 4, 6, 2, 3, 64, 0)
bool conv_cfg_fp8_ok(int cfg)
{
    switch (cfg) {
#define X(id, wp, wc, tp, tc, ns, bk, nl) case id: return true;
        CONV_CFGS_FP8(X)
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
    default: return hipErrorInvalidValue;
    }
}
