// Fused network stem for gfx950: conv 3x3/s1 (3 -> C0 = 32 channels) immediately consumed by conv 3x3/s2 (C0 -> C1 = 64),
// both with BN folded, bias and leaky/linear epilogues -- darknet-53's first two layers (V3/yolo_v3.py:15-24 `darknet53`,
// DN cfg layers 0-1).  Run separately they are pure HBM traffic: layer 0 writes N*S*S*32 bf16 (354 MB at 416^2, batch 32)
// that layer 1 reads straight back.  Here a workgroup produces an 8 x 16 tile of layer-1 pixels: it first computes the
// 17 x 33 layer-0 pixels that tile needs into LDS (bf16, exactly the values the unfused layer would have stored, zeros
// where layer 1 pads), then contracts them with the layer-1 filters held in registers.  HBM sees the 8-channel image
// once and the layer-1 output once.
//
//   input    the 19 x 35 input pixels (16 B each) a tile needs are gathered into LDS by LDS-DMA two tiles AHEAD (double
//            buffered), so no wave ever waits on global-memory latency inside a tile; image borders are zero-filled
//            by the buffer range check.
//   phase A  (producer waves) one 16-pixel group per step and wave: a lane's B fragment for K-group (kk, lq) is the 16-byte channel
//            vector of ONE input pixel (tap kk*4+lq), one ds_read_b128; 6 MFMAs; the 4 channels a lane ends up with go
//            to LDS as one ds_write_b64.  LDS pixel pitch is 80 B: with layer 1's stride-2 access
//            the 16 lanes of every ds_read_b128 lane group then fall on 16 distinct 16-B bank slots.
//   phase B  (consumer waves) wave w (of 4) owns output rows 2w, 2w+1 of the tile (16 pixels each) x 64 channels: per tap and row one ds_read_b128
//            (K-step = the tap's 32 channels) feeding 4 MFMAs against register-resident filters.
//   epilogue through LDS so that the global stores are 16 B per lane, whole 128-B pixel rows.
// Optional tail: the 1x1 conv 64 -> 32 that follows (darknet-53 layer 2) runs on the staged layer-1 tile before it
// leaves LDS, so that layer never re-reads its 177 MB input.
// Workgroups are persistent (one 8-wave workgroup per CU) so the filter fragments (144 VGPRs of layer 1 in a consumer wave) are loaded once per wave.
#include "kernels.h"
#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Both kernels are VALU-issue-bound (two waves per SIMD, ~4 cycles per instruction): the epilogues are written for instruction count.
typedef __bf16 st_bf16x2 __attribute__((ext_vector_type(2)));
typedef float st_f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 st_f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 st_f16x8 __attribute__((ext_vector_type(8)));
// two floats -> packed 16-bit pair (lo | hi << 16), RNE: bf16 (one v_cvt_pk_bf16_f32) or, H16, fp16 saturating at +-65504 (MODE.FP16_OVFL, set at the top of the H16 kernels)
template <bool H16> __device__ __forceinline__ uint32_t stem_pk(float lo, float hi)
{
    if constexpr (H16) return __builtin_bit_cast(uint32_t, __builtin_convertvector(st_f32x2{lo, hi}, st_f16x2));      // (the H16 kernels run with MODE.FP16_OVFL: an overflowing conversion saturates)
    else return __builtin_bit_cast(uint32_t, __builtin_convertvector(st_f32x2{lo, hi}, st_bf16x2));
}
template <bool H16> __device__ __forceinline__ float stem_lo(uint32_t w) { if constexpr (H16) return (float)__builtin_bit_cast(st_f16x2, w)[0]; else return __builtin_bit_cast(float, w << 16); }
template <bool H16> __device__ __forceinline__ float stem_hi(uint32_t w) { if constexpr (H16) return (float)__builtin_bit_cast(st_f16x2, w)[1]; else return __builtin_bit_cast(float, w & 0xffff0000u); }
// max for finite operands as ONE instruction (fmaxf is preceded by an sNaN-quieting v_max)
__device__ __forceinline__ float stem_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// acc + bias, activation (slope 0.1: leaky as max(v, 0.1 v); slope 1: linear), rounded to bf16: four channels as two packed words
template <bool H16> __device__ __forceinline__ uint2 stem_epi(const f32x4 acc, const f32x4 bias, const float slope)
{
    f32x4 v = acc + bias;
    const f32x4 t = v * slope;
    return uint2{stem_pk<H16>(stem_max(v[0], t[0]), stem_max(v[1], t[1])), stem_pk<H16>(stem_max(v[2], t[2]), stem_max(v[3], t[3]))};
}
template <bool H16> __device__ __forceinline__ f32x4 stem_mma(const bf16x8 a, const bf16x8 b, const f32x4 c)
{
    if constexpr (H16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(st_f16x8, a), __builtin_bit_cast(st_f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

constexpr int ST_TH = 8, ST_TW = 16;                 // layer-1 output tile
constexpr int ST_LH = 2 * ST_TH + 1, ST_LW = 2 * ST_TW + 1;   // layer-0 pixels it needs: 17 x 33
constexpr int ST_NPIX = ST_LH * ST_LW;               // 561
constexpr int ST_GROUPS = (ST_NPIX + 15) / 16;       // 36
constexpr int ST_PITCH = 80;                         // LDS bytes per layer-0 pixel (32 bf16 + 16 B pad)
constexpr int ST_L0_BYTES = ST_GROUPS * 16 * ST_PITCH;        // 46080
constexpr int ST_OPITCH = 64 * 2 + 16;               // staged layer-1 tile: 64 bf16 + pad
constexpr int ST_OUT_BYTES = ST_TH * ST_TW * ST_OPITCH;       // 18432


constexpr int ST_NW = 8;                             // waves per workgroup == ST_TH
typedef __attribute__((address_space(3))) void st_lds_void;
typedef __attribute__((address_space(3))) char lds_char;
constexpr int ST_IH = ST_LH + 2, ST_IW = ST_LW + 2;          // input pixels a tile needs: 19 x 35
constexpr int ST_INPIX = ST_IH * ST_IW;                      // 665
constexpr int ST_INCHUNKS = (ST_INPIX + 63) / 64;            // 64-pixel LDS-DMA pieces: 11
constexpr int ST_IN_BYTES = ST_INCHUNKS * 1024 + 16;         // + one 16-B slot of zeros (taps 9..11 of the K padding)
constexpr int ST_RAW_ROW = 112;                              // U8 form: bytes of one raw input row in LDS (35 px x 3 B = 105, + up to 3 of alignment slack, as 28 dwords)
constexpr int ST_RAW_DW = ST_IH * (ST_RAW_ROW / 4);          // 532 dwords per tile
constexpr int ST_RAW_PIECES = (ST_RAW_DW + 63) / 64;         // 4-byte LDS-DMA pieces (256 B each): 9
constexpr int ST_RAW_BYTES = ST_RAW_PIECES * 256;
constexpr int ST_O2PITCH = 32 * 2 + 16;                      // staged tail tile: 32 bf16 + pad
constexpr int ST_OUT2_BYTES = ST_TH * ST_TW * ST_O2PITCH;    // 10240
constexpr int ST_W2PITCH = 64 * 2 + 16;                      // tail filter row pitch: 128 B rows would put all 16 lanes of a
                                                             // fragment read on two banks (8-way conflict)
constexpr int ST_W2_BYTES = 32 * ST_W2PITCH;                 // tail filters [32][64] bf16, kept in LDS (no registers left)
constexpr int ST_B1_BYTES = 64 * 4;                          // layer-1 bias: read from LDS in the epilogue (16 VGPRs the filters need)
constexpr int ST_B2_BYTES = 32 * 4;                          // tail bias, in LDS too: a global load inside the tile loop
                                                             // would make hipcc wait vmcnt(0) and drain the input prefetch

// U8: the image is read as the caller's uint8 [N, H, W, 3] itself: the raw rows of a tile are gathered by 4-byte LDS-DMA and converted to
// the 16-byte pixel records (x * scale [* mul + add], rounded to the storage type: exactly k_preprocess's arithmetic) one tile ahead of
// phase A -- the separate conversion launch and its [N, H, W, 8] tensor (88 MB written, 88 MB read at 416 x 416 x 32) disappear.
//
// Wave roles.  Phase A is VALU-bound (bias / leaky / rounding of 32 channels per layer-0 pixel behind three short MFMAs), phase B is
// MFMA-bound; run one after the other by all eight waves, each left the other pipe idle (ablation, 416 x 416 x 32: A 76, B 41, tail 25,
// stores 17, conversion 10 of 187 us).  So the workgroup is split: waves 0-3 PRODUCE (phase A of tile i+1 into one half of a
// double-buffered layer-0 tile), waves 4-7 CONSUME (phase B, epilogue, stores and 1x1 tail of tile i -- two tile rows per wave, staged
// through wave-private LDS rows, no workgroup barrier -- and the uint8 conversion of tile i+2).  Every SIMD holds one wave of each kind,
// so its matrix pipe and its VALU are fed from different instruction streams, and ONE barrier per tile moves the pipeline on:
//   raw rows (LDS-DMA, producers)  tile i+3  ->  records (consumers)  tile i+2  ->  layer-0 tile (producers)  tile i+1  ->  output (consumers)  tile i
// Only producers issue LDS-DMA (their vmcnt(0) before the barrier is exact); consumers never wait for their global stores.
template <bool H16, bool U8 = false>
__global__ __launch_bounds__(64 * ST_NW) void conv_stem_c32_c64(const StemArgs a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if constexpr (H16) __builtin_amdgcn_s_setreg((0 << 11) | (23 << 6) | 1, 1);      // hwreg(HW_REG_MODE, 23, 1) = FP16_OVFL: fp16 conversions saturate
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const int sub = wave & 3;                                    // index among the four waves of this wave's role
    const float slope0 = a.act0 == ACT_LEAKY ? 0.1f : 1.f, slope1 = a.act1 == ACT_LEAKY ? 0.1f : 1.f, slope2 = a.act2 == ACT_LEAKY ? 0.1f : 1.f;

    const int tiles_x = (a.Wo + ST_TW - 1) / ST_TW, tiles_y = (a.Ho + ST_TH - 1) / ST_TH;
    const int per_img = tiles_x * tiles_y, ntiles = a.N * per_img;
    const int nt = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;     // tiles of this workgroup: blockIdx.x + j * gridDim.x
    // A workgroup walks its tiles in steps of gridDim.x; their (image, tile row, tile column) are carried, not divided out of the tile
    // index at every use (three integer divisions per decode, ~190 scalar instructions per tile in a wave that is issue-bound).
    struct TilePos { int n, ty, tx; };
    auto tile_pos = [&](int tile) { TilePos q; q.n = tile / per_img; const int tr = tile - q.n * per_img; q.ty = tr / tiles_x; q.tx = tr - q.ty * tiles_x; return q; };
    const TilePos gstep = tile_pos((int)gridDim.x);
    auto advance = [&](TilePos &q) {
        q.tx += gstep.tx; q.ty += gstep.ty; q.n += gstep.n;
        if (q.tx >= tiles_x) { q.tx -= tiles_x; ++q.ty; }
        if (q.ty >= tiles_y) { q.ty -= tiles_y; ++q.n; }
    };
    // input pixel (iy0, ix0) = record (0, 0) of `tile` (may be negative at the left / top border)
    auto tile_origin = [&](const TilePos &q, int &n, int &iy0, int &ix0) {
        n = q.n; const int ty = q.ty, tx = q.tx;
        iy0 = 2 * ty * ST_TH - 2; ix0 = 2 * tx * ST_TW - 2;
    };

    // LDS: [raw rows x 2 (U8) | pixel records x 2 | layer-0 tile x 2 | staged layer-1 tile | staged tail tile | tail filters, biases]
    constexpr int RAWB = U8 ? ST_RAW_BYTES : 0;
    char *const raw0 = smem, *const raw1 = smem + RAWB;
    char *const pix0 = smem + 2 * RAWB, *const pix1 = pix0 + ST_IN_BYTES;
    char *const l00 = pix1 + ST_IN_BYTES, *const l01 = l00 + ST_L0_BYTES;
    char *const lo = l01 + ST_L0_BYTES, *const lo2 = lo + ST_OUT_BYTES, *const lw2 = lo2 + ST_OUT2_BYTES;
    if (a.w2)                                                    // tail filters -> LDS, 16 B per thread (256 pieces)
        if (tid < 32 * 8) {
            const int row = tid >> 3, piece = tid & 7;
            *(uint4 *)(lw2 + row * ST_W2PITCH + piece * 16) = *(const uint4 *)((const bf16_t *)a.w2 + (size_t)row * a.Kpad2 + piece * 8);
            if (tid < 32) *(float *)(lw2 + ST_W2_BYTES + tid * 4) = a.b2[tid];
        }
    if (tid < 64) *(float *)(lw2 + ST_W2_BYTES + ST_B2_BYTES + tid * 4) = a.b1[tid];
    if (tid < 4) { ((uint32_t *)(pix0 + ST_INCHUNKS * 1024))[tid] = 0; ((uint32_t *)(pix1 + ST_INCHUNKS * 1024))[tid] = 0; }   // the zero slot of the K padding

    if (wave < 4) {
        // =========================================== producers ===========================================
        bf16x8 fw0[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int kk = 0; kk < 3; ++kk)
                fw0[i][kk] = *(const bf16x8 *)((const bf16_t *)a.w0 + (size_t)(i * 16 + l15) * a.Kpad0 + (kk * 4 + lq) * 8);
        f32x4 b0v[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) b0v[i] = *(const f32x4 *)(a.b0 + i * 16 + lq * 4);
        // byte offsets of this lane's taps lq and 4 + lq relative to its pixel in the record tile; tap 8 + lq exists for lq == 0 only
        // (the others are K padding: they read the zero slot)
        const int tap0 = lq, tap1 = 4 + lq;
        const int tapoff0 = (((tap0 * 11) >> 5) * ST_IW + tap0 - 3 * ((tap0 * 11) >> 5)) * 16, tapoff1 = (((tap1 * 11) >> 5) * ST_IW + tap1 - 3 * ((tap1 * 11) >> 5)) * 16;
        // the image is read through a buffer descriptor: an out-of-range offset makes the LDS-DMA write zeros (the padding); U8: the
        // descriptor ends with the image tensor -- the 4-byte loads of a row's slack must not touch what lies behind it
        __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)(U8 ? (const void *)a.in_u8 : a.in), 0, U8 ? (unsigned)(((size_t)a.N * a.H * a.W * 3) & ~(size_t)3) : 0x80000000u, 0x00020000);

        // gather the 19 x 35 input pixels of `tile` into `dst`: pieces of 64 lanes, piece c by producer wave c % 4
        auto fetch = [&](const TilePos &tile, lds_char *dst) {
            int n, iy0, ix0; tile_origin(tile, n, iy0, ix0);
            if constexpr (U8) {
#pragma unroll
                for (int k = 0; k < (ST_RAW_PIECES + 3) / 4; ++k) {
                    const int c = sub + 4 * k;
                    if (c < ST_RAW_PIECES) {
                        const int g = c * 64 + lane;                         // dword of the tile's raw image: row g / 28, dword g % 28
                        const int r = (g * 2341) >> 16;                      // g / 28 for g < 1170
                        const int d = g - r * (ST_RAW_ROW / 4);
                        const int iy = iy0 + r;
                        const int s_r = ((n * a.H + iy) * a.W + ix0) * 3;    // first byte the row needs
                        const bool ok = g < ST_RAW_DW && (unsigned)iy < (unsigned)a.H;
                        const unsigned off = ok ? (unsigned)((s_r & ~3) + 4 * d) : 0x80000000u;      // (a negative offset is out of range too: zeros)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (st_lds_void *)(dst + c * 256), 4, off, 0, 0, 0);
                    }
                }
            } else {
#pragma unroll
                for (int k = 0; k < (ST_INCHUNKS + 3) / 4; ++k) {
                    const int c = sub + 4 * k;
                    if (c < ST_INCHUNKS) {
                        const int q = c * 64 + lane;
                        const int ry = (q * 1873) >> 16;                 // q / 35 for q < 704
                        const int rxx = q - ry * ST_IW;
                        const int iy = iy0 + ry, ix = ix0 + rxx;
                        const bool ok = q < ST_INPIX && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                        const unsigned off = ok ? (unsigned)(((n * a.H + iy) * a.W + ix) * a.in_stride) * 2u : 0x80000000u;
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (st_lds_void *)(dst + c * 1024), 16, off, 0, 0, 0);
                    }
                }
            }
        };
        // phase A: the 17 x 33 layer-0 pixels of `tile` (its records in `in_cur`) -> `l0`; one 16-pixel group per step, a lane's B
        // fragment (kk, lq) = the record of tap kk*4+lq.  36 groups = 9 per producer wave, and everything about a group except the
        // image border is the same in every tile: the three record offsets of each group live in registers (a producer wave has them
        // to spare), the layer-0 address is linear in the group.  Three groups are in flight at a time (reads, 18 MFMAs, epilogues).
        constexpr int NG = ST_GROUPS / 4;
        typedef const __attribute__((address_space(3))) bf16x8 *lds_frag_p;
        typedef __attribute__((address_space(3))) uint2 *lds_u2_p;
        // (the addresses below carry the LDS address of buffer 0; phase A's `in_cur` / `l0` parameters are then the byte distance of the
        // buffer in use from buffer 0 -- a constant that folds into the instruction's offset field, no per-read address add)
        const uint32_t pixa = (uint32_t)(uintptr_t)(lds_char *)pix0, l0a = (uint32_t)(uintptr_t)(lds_char *)l00;
        uint32_t rd0[NG], rd1[NG], rd2[NG]; int lyx[NG];
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const int idx = (sub + 4 * j) * 16 + l15;
            const int ly = (idx * 1986) >> 16;                   // idx / 33 for idx < 576
            const int lx = idx - ly * ST_LW;
            // layer-0 pixel (ly, lx) sits at record (ly + 1, lx + 1); tap (kh, kw) reads (ly + kh, lx + kw).  (The rows past the
            // tile, idx >= ST_NPIX, read defined bytes of the fetched pieces and are zeroed below.)
            const int pbase = (ly * ST_IW + lx) * 16;
            rd0[j] = pixa + pbase + tapoff0; rd1[j] = pixa + pbase + tapoff1; rd2[j] = pixa + (lq == 0 ? pbase + (2 * ST_IW + 2) * 16 : ST_INCHUNKS * 1024);
            lyx[j] = idx < ST_NPIX ? (ly | lx << 8) : 0x7f7f;    // (past the tile: never inside)
            asm volatile("" : "+v"(rd0[j]), "+v"(rd1[j]), "+v"(rd2[j]));     // (held, not recomputed from their parts at every use)
        }
        const uint32_t wr = l0a + (sub * 16 + l15) * ST_PITCH + lq * 8;      // + j * 64 * ST_PITCH + i * 32
        const bool past = sub == 3 && l15 >= 1;                  // the last group holds ONE pixel of the tile (561 = 35 * 16 + 1)
        auto phase_a = [&](const TilePos &tile, const lds_char *__restrict__ in_cur, lds_char *__restrict__ l0) {
            int n, iy0, ix0; tile_origin(tile, n, iy0, ix0);
            const int gy0 = iy0 + 1, gx0 = ix0 + 1;              // layer-0 coordinates of LDS pixel (0, 0)
            // a tile whose 17 x 33 layer-0 pixels all lie inside the image (every tile but the top row and the left column) needs no
            // per-pixel test
            const bool interior = gy0 >= 0 && gx0 >= 0 && gy0 + ST_LH <= a.H && gx0 + ST_LW <= a.W;
            auto run = [&](auto interior_c) {
                constexpr bool INT = decltype(interior_c)::value;
#pragma unroll
                for (int jj = 0; jj < NG; jj += 3) {
                    bf16x8 fx[3][3];
#pragma unroll
                    for (int u = 0; u < 3; ++u) {
                        fx[u][0] = *(lds_frag_p)(in_cur + rd0[jj + u]);
                        fx[u][1] = *(lds_frag_p)(in_cur + rd1[jj + u]);
                        fx[u][2] = *(lds_frag_p)(in_cur + rd2[jj + u]);
                    }
                    f32x4 acc[3][2];
#pragma unroll
                    for (int u = 0; u < 3; ++u)
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            acc[u][i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                            for (int kk = 0; kk < 3; ++kk) acc[u][i] = stem_mma<H16>(fw0[i][kk], fx[u][kk], acc[u][i]);
                        }
#pragma unroll
                    for (int u = 0; u < 3; ++u) {
                        const int j = jj + u;
                        bool zero;                               // layer 1's zero padding, and the unused tail of the last group
                        if constexpr (INT) zero = j == NG - 1 && past;
                        else zero = !(((unsigned)(gy0 + (lyx[j] & 0xff)) < (unsigned)a.H) & ((unsigned)(gx0 + (lyx[j] >> 8)) < (unsigned)a.W));
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            uint2 pk = stem_epi<H16>(acc[u][i], b0v[i], slope0);
                            if (!INT || j == NG - 1) { pk.x = zero ? 0u : pk.x; pk.y = zero ? 0u : pk.y; }
                            *(lds_u2_p)(l0 + wr + j * 64 * ST_PITCH + i * 32) = pk;
                        }
                    }
                }
            };
            if (interior) run(std::true_type{}); else run(std::false_type{});
        };
        // Every LDS region is its own __restrict__ parameter: that is what lets hipcc see that the LDS-DMA filling `dma_dst`
        // cannot alias phase A's reads, instead of waiting vmcnt(0) before the first ds_read after it.
        TilePos pa = tile_pos((int)blockIdx.x), pf = pa;         // tiles of phase A (i + 1) and of the fetch (i + 3, or i + 2) at step i
        auto produce = [&](int i, lds_char *__restrict__ dma_dst, const lds_char *__restrict__ in_cur, lds_char *__restrict__ l0) {
            const int jf = i + (U8 ? 3 : 2);                     // tile whose input is fetched during this step
            if (jf < nt) fetch(pf, dma_dst);
            advance(pf);
            if (i + 1 >= 0 && i + 1 < nt) { phase_a(pa, in_cur, l0); advance(pa); }
        };
        static_assert(ST_GROUPS % 12 == 0, "phase A: whole groups per producer wave, three at a time");
        if constexpr (U8) { fetch(pf, (lds_char *)raw0); advance(pf); }
        __builtin_amdgcn_s_waitcnt(0x0070);
        __builtin_amdgcn_s_barrier();
        for (int i = -2; i <= nt; ++i) {                         // (one step past the last tile: the consumers drain it there)
            if (i & 1) produce(i, (lds_char *)(U8 ? raw0 : pix1), (const lds_char *)(uintptr_t)0, (lds_char *)(uintptr_t)0);
            else produce(i, (lds_char *)(U8 ? raw1 : pix0), (const lds_char *)(uintptr_t)ST_IN_BYTES, (lds_char *)(uintptr_t)ST_L0_BYTES);
            __builtin_amdgcn_s_waitcnt(0x0070);                  // vmcnt(0) lgkmcnt(0): the fetched input has landed, the layer-0 tile is written
            __builtin_amdgcn_s_barrier();
        }
    } else {
        // =========================================== consumers ===========================================
        // The consumers are the second-dispatched half of the workgroup -- the loser of every VALU arbitration against the producer on
        // its SIMD -- and the longer of the two streams: static priority for them (same box: 149.7 -> 131.8 us; moving the uint8
        // conversion to the producers instead: 136.3, both: 137.7).
        __builtin_amdgcn_s_setprio(1);
        bf16x8 fw1[4][9];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int t = 0; t < 9; ++t)
                fw1[ct][t] = *(const bf16x8 *)((const bf16_t *)a.w1 + (size_t)(ct * 16 + l15) * a.Kpad1 + t * 32 + lq * 8);
        const int ctid = tid - 256;
        // raw rows of `tile` (in `raw`) -> 16-byte pixel records (3 converted channels + 5 zeros) in `pix`; pixels outside the image are zeros
        auto convert_raw = [&](const TilePos &tile, const char *__restrict__ raw, char *__restrict__ pix) {
            int n, iy0, ix0; tile_origin(tile, n, iy0, ix0);
#pragma unroll
            for (int j = 0; j < (ST_INPIX + 255) / 256; ++j) {
                const int p = ctid + j * 256;
                if (p < ST_INPIX) {
                    const int ry = (p * 1873) >> 16, rxx = p - ry * ST_IW;       // p / 35
                    const int iy = iy0 + ry, ix = ix0 + rxx;
                    const int s_r = ((n * a.H + iy) * a.W + ix0) * 3;
                    const int b = (s_r & 3) + 3 * rxx;                           // byte of the pixel in its row's LDS slot
                    const unsigned *q = (const unsigned *)(raw + ry * ST_RAW_ROW + (b & ~3));
                    const unsigned x = __builtin_amdgcn_alignbyte(q[1], q[0], (unsigned)(b & 3));
                    float v0 = __fmul_rn((float)(x & 0xffu), a.in_scale), v1 = __fmul_rn((float)((x >> 8) & 0xffu), a.in_scale), v2 = __fmul_rn((float)((x >> 16) & 0xffu), a.in_scale);
                    if (a.in_mul != 1.0f || a.in_add != 0.0f) { v0 = __fadd_rn(__fmul_rn(v0, a.in_mul), a.in_add); v1 = __fadd_rn(__fmul_rn(v1, a.in_mul), a.in_add); v2 = __fadd_rn(__fmul_rn(v2, a.in_mul), a.in_add); }
                    const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                    *(uint4 *)(pix + p * 16) = ok ? uint4{stem_pk<H16>(v0, v1), stem_pk<H16>(v2, 0.f), 0u, 0u} : uint4{0u, 0u, 0u, 0u};
                }
            }
        };
        // Phase B of a tile: this wave owns tile rows 2*sub and 2*sub+1 (16 pixels each) x 64 channels; per tap and row one ds_read_b128
        // (K-step = the tap's 32 channels) feeds 4 MFMAs against register-resident filters.  Epilogue, stores and the 1x1 tail go through
        // this wave's own rows of the staged tiles: LDS is in order within a wave, so no workgroup barrier separates them.
        // One consumer step is software-pipelined: the 18 tap steps of tile i carry, in their shadow, what is left of tile i-1 -- the
        // stores of its staged rows, its 1x1 tail and the tail's stores -- and the uint8 conversion of tile i+2; the epilogue of tile i
        // then overwrites the staged rows.  One straight-line block: stores that must not happen (no previous tile, pixels past the
        // image) go to an out-of-range buffer offset instead of around a branch.  (Same box: 131.9 -> 129.2 us against the plain
        // sequence -- the step is bound by what the two waves of a SIMD can issue and by LDS traffic, ~340 KB per tile, not by stalls.)
        __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)a.out, 0, 0x80000000u, 0x00020000);
        __amdgpu_buffer_rsrc_t ro2 = __builtin_amdgcn_make_buffer_rsrc((void *)(a.w2 ? a.out2 : a.out), 0, 0x80000000u, 0x00020000);
        typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
        char *const lrow = lo + (2 * sub) * ST_TW * ST_OPITCH;           // this wave's 32 staged pixels
        char *const lrow2 = lo2 + (2 * sub) * ST_TW * ST_O2PITCH;
        int pn = 0, poy0 = 0, pox0 = 0;                                  // image and origin (this wave's rows) of the previous tile
        TilePos pi = tile_pos((int)blockIdx.x), pc = pi;         // tile i (from step 0 on) and tile i + 2, the one converted at step i
        auto consume_p = [&](int i, const char *__restrict__ l0, const char *__restrict__ raw, char *__restrict__ pix, char *__restrict__ lr, char *__restrict__ lr2,
                             const char *__restrict__ lw2_) {
            const bool pv = i >= 1;                                      // there is a previous tile whose rows are staged
            const int n = pi.n, ty = pi.ty, tx = pi.tx;                  // (the step after the last tile only drains: its phase B result is never stored)
            // ---- previous tile: buffer offsets of its 4 + 2 store pieces (formed where they are used: six registers less across the taps) ----
            auto piece_off = [&](int it) {
                const int c = lane + it * 64, px = c >> 3, chunk = c & 7;
                const int oy = poy0 + (px >> 4), ox = pox0 + (px & 15);
                return pv && oy < a.Ho && ox < a.Wo ? (unsigned)((((pn * a.Ho + oy) * a.Wo + ox) * a.out_stride + chunk * 8) * 2) : 0x80000000u;
            };
            auto piece2_off = [&](int it) {
                const int c = lane + it * 64, px = c >> 2, chunk = c & 3;
                const int oy = poy0 + (px >> 4), ox = pox0 + (px & 15);
                return pv && a.w2 && oy < a.Ho && ox < a.Wo ? (unsigned)((((pn * a.Ho + oy) * a.Wo + ox) * a.out2_stride + chunk * 8) * 2) : 0x80000000u;
            };
            // uint8 conversion of tile i+2: geometry
            int cn = 0, ciy0 = 0, cix0 = 0; const bool cv = U8 && i + 2 < nt;
            if (cv) tile_origin(pc, cn, ciy0, cix0);
            auto convert_round = [&](int j) {
                const int p = ctid + j * 256;
                const int ry = (p * 1873) >> 16, rxx = p - ry * ST_IW;           // p / 35
                const int iy = ciy0 + ry, ix = cix0 + rxx;
                const int s_r = ((cn * a.H + iy) * a.W + cix0) * 3;
                const int b = (s_r & 3) + 3 * rxx;                               // byte of the pixel in its row's LDS slot
                const unsigned *q = (const unsigned *)(raw + (p < ST_INPIX ? ry * ST_RAW_ROW + (b & ~3) : 0));
                const unsigned x = __builtin_amdgcn_alignbyte(q[1], q[0], (unsigned)(b & 3));
                float v0 = __fmul_rn((float)(x & 0xffu), a.in_scale), v1 = __fmul_rn((float)((x >> 8) & 0xffu), a.in_scale), v2 = __fmul_rn((float)((x >> 16) & 0xffu), a.in_scale);
                if (a.in_mul != 1.0f || a.in_add != 0.0f) { v0 = __fadd_rn(__fmul_rn(v0, a.in_mul), a.in_add); v1 = __fadd_rn(__fmul_rn(v1, a.in_mul), a.in_add); v2 = __fadd_rn(__fmul_rn(v2, a.in_mul), a.in_add); }
                const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                if (cv && p < ST_INPIX) *(uint4 *)(pix + p * 16) = ok ? uint4{stem_pk<H16>(v0, v1), stem_pk<H16>(v2, 0.f), 0u, 0u} : uint4{0u, 0u, 0u, 0u};
            };
            const char *xb = l0 + ((4 * sub) * ST_LW + 2 * l15) * ST_PITCH + lq * 16;
            u32x4_t sd[2];
            f32x4 acc2[2][2];
            bf16x8 tx_[2], tw_[2];
            auto tail_frags = [&](int kk) {
                tw_[0] = *(const bf16x8 *)(lw2_ + l15 * ST_W2PITCH + (kk * 4 + lq) * 16); tw_[1] = *(const bf16x8 *)(lw2_ + (16 + l15) * ST_W2PITCH + (kk * 4 + lq) * 16);
#pragma unroll
                for (int r = 0; r < 2; ++r) tx_[r] = *(const bf16x8 *)(lr + (r * ST_TW + l15) * ST_OPITCH + (kk * 4 + lq) * 16);
            };
            auto tail_mma = [&]() {
#pragma unroll
                for (int r = 0; r < 2; ++r) { acc2[r][0] = stem_mma<H16>(tw_[0], tx_[r], acc2[r][0]); acc2[r][1] = stem_mma<H16>(tw_[1], tx_[r], acc2[r][1]); }
            };
            // The wave's two tile rows one after the other (16 accumulator registers at a time, not 32): 18 steps of four MFMAs; the first
            // row's epilogue rides with the second row's taps.
            f32x4 accr[2][4];
            auto epilogue = [&](int r, int c0) {
#pragma unroll
                for (int ct = c0; ct < c0 + 2; ++ct)
                    *(uint2 *)(lr + (r * ST_TW + l15) * ST_OPITCH + (ct * 16 + lq * 4) * 2) = stem_epi<H16>(accr[r][ct], *(const f32x4 *)(lw2_ + ST_W2_BYTES + ST_B2_BYTES + (ct * 16 + lq * 4) * 4), slope1);
            };
            bf16x8 xn = *(const bf16x8 *)xb;                             // fragment of the next step
#pragma unroll
            for (int h = 0; h < 18; ++h) {
                const int r = h / 9, t = h - r * 9;
                // ---- the piece of the older / newer tiles that rides with this step ----
                if (h == 0) { sd[0] = *(const u32x4_t *)(lr + (lane >> 3) * ST_OPITCH + (lane & 7) * 16); sd[1] = *(const u32x4_t *)(lr + ((lane + 64) >> 3) * ST_OPITCH + (lane & 7) * 16); }
                if (h == 1) {
                    __builtin_amdgcn_raw_buffer_store_b128(sd[0], ro, piece_off(0), 0, OUT_STORE_AUX); __builtin_amdgcn_raw_buffer_store_b128(sd[1], ro, piece_off(1), 0, OUT_STORE_AUX);
                    sd[0] = *(const u32x4_t *)(lr + ((lane + 128) >> 3) * ST_OPITCH + (lane & 7) * 16); sd[1] = *(const u32x4_t *)(lr + ((lane + 192) >> 3) * ST_OPITCH + (lane & 7) * 16);
                }
                if (h == 2) { __builtin_amdgcn_raw_buffer_store_b128(sd[0], ro, piece_off(2), 0, OUT_STORE_AUX); __builtin_amdgcn_raw_buffer_store_b128(sd[1], ro, piece_off(3), 0, OUT_STORE_AUX); }
                if (h == 3) tail_frags(0);
                if (h == 4) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) { acc2[q][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[q][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
                    tail_mma();
                }
                if (h == 5) tail_frags(1);
                if (h == 6) tail_mma();
                if (h == 7) {
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int k = 0; k < 2; ++k)
                            *(uint2 *)(lr2 + (q * ST_TW + l15) * ST_O2PITCH + (k * 16 + lq * 4) * 2) = stem_epi<H16>(acc2[q][k], *(const f32x4 *)(lw2_ + ST_W2_BYTES + (k * 16 + lq * 4) * 4), slope2);
                }
                if (h == 8) { sd[0] = *(const u32x4_t *)(lr2 + (lane >> 2) * ST_O2PITCH + (lane & 3) * 16); sd[1] = *(const u32x4_t *)(lr2 + ((lane + 64) >> 2) * ST_O2PITCH + (lane & 3) * 16); }
                if (h == 9) { __builtin_amdgcn_raw_buffer_store_b128(sd[0], ro2, piece2_off(0), 0, OUT_STORE_AUX); __builtin_amdgcn_raw_buffer_store_b128(sd[1], ro2, piece2_off(1), 0, OUT_STORE_AUX); }
                if (h == 10) epilogue(0, 0);
                if (h == 11) epilogue(0, 2);
                if (h == 12) { if constexpr (U8) convert_round(0); }
                if (h == 14) { if constexpr (U8) convert_round(1); }
                if (h == 16) { if constexpr (U8) convert_round(2); }
                // ---- tap t of row r of tile i (its fragment was requested one step earlier: nothing crosses the fence below) ----
                const bf16x8 x = xn;
                if (h + 1 < 18) { const int r1 = (h + 1) / 9, t1 = h + 1 - r1 * 9, kh1 = t1 / 3, kw1 = t1 - kh1 * 3; xn = *(const bf16x8 *)(xb + ((kh1 + 2 * r1) * ST_LW + kw1) * ST_PITCH); }
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) accr[r][ct] = stem_mma<H16>(fw1[ct][t], x, t == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : accr[r][ct]);
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- epilogue of the second row: with the first row's, it replaces the previous tile's staged rows ----
            epilogue(1, 0); epilogue(1, 2);
            pn = n; poy0 = ty * ST_TH + 2 * sub; pox0 = tx * ST_TW;
        };
        __builtin_amdgcn_s_waitcnt(0x0070);
        __builtin_amdgcn_s_barrier();
        for (int i = -2; i <= nt; ++i) {
            if (i >= 0) { if (i & 1) consume_p(i, l01, raw1, pix1, lrow, lrow2, lw2); else consume_p(i, l00, raw0, pix0, lrow, lrow2, lw2); advance(pi); }
            else if constexpr (U8) { if (i + 2 < nt) { if (i & 1) convert_raw(pc, raw1, pix1); else convert_raw(pc, raw0, pix0); } }
            advance(pc);
            __builtin_amdgcn_s_waitcnt(0xc07f);                  // lgkmcnt(0): the records are written, the layer-0 tile is read
            __builtin_amdgcn_s_barrier();
        }
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// Halo-staged 3x3/s1 conv for a SMALL channel count (Cin = 32 -> Cout = 64, + bias, leaky/linear, optional shortcut):
// darknet-53 layer 3.  In the tiled kernel every input pixel crosses the L2 -> LDS path nine times (once per tap) and
// a K of 288 is nine short K-steps per tile; here the 10 x 18 input pixels of an 8 x 16 output tile (64 B each) are
// gathered into LDS ONCE by LDS-DMA, one tile ahead, and the nine taps are nine ds_read_b128 at shifted addresses against
// register-resident filters -- the phase-B structure of the stem kernel above with stride 1.  The shortcut source tile is
// DMA'd into LDS at the start of its tile and added in the store pass (conv output rounded to bf16 first, as everywhere).
// LDS pixel records are 64 B with the 16-B chunk index XOR 2*((pixel>>2)&1): conflict-free for ds_read_b128's lane groups
// at pixel stride 1.  HBM-bound by design: input once, shortcut once, output once.
// A tile is ~0.5 us of work against ~2 us of memory latency, so input AND shortcut tiles are requested TWO tiles ahead into rings of
// three LDS slots each (27.5 KB per tile: 55 KB in flight per CU, 14 MB on the chip); the one counted vmcnt of a tile waits for the
// requests of the tile before it only.  (With the shortcut requested at the start of its own tile and a vmcnt(0) behind nine taps of
// MFMAs, every tile paid one whole round trip: 2.5 us per tile, 4.2 TB/s.)
constexpr int HL_IH = ST_TH + 2, HL_IW = ST_TW + 2;          // 10 x 18 input pixels
constexpr int HL_INPIX = HL_IH * HL_IW;                      // 180
constexpr int HL_INCHUNKS = (HL_INPIX * 4 + 63) / 64;        // 16-B pieces / 64 lanes: 12 LDS-DMA instructions
constexpr int HL_IN_BYTES = HL_INCHUNKS * 1024;
constexpr int HL_RES_BYTES = ST_TH * ST_TW * 128;            // shortcut tile [128 px][64 bf16], lane-linear

template <bool H16>
__global__ __launch_bounds__(64 * ST_NW) void conv_halo_c32_c64(const HaloArgs a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if constexpr (H16) __builtin_amdgcn_s_setreg((0 << 11) | (23 << 6) | 1, 1);      // hwreg(HW_REG_MODE, 23, 1) = FP16_OVFL: fp16 conversions saturate
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    bf16x8 fw[4][9];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int t = 0; t < 9; ++t)
            fw[ct][t] = *(const bf16x8 *)((const bf16_t *)a.w + (size_t)(ct * 16 + l15) * a.Kpad + t * 32 + lq * 8);
    f32x4 bv[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) bv[ct] = *(const f32x4 *)(a.b + ct * 16 + lq * 4);
    const float slope = a.act == ACT_LEAKY ? 0.1f : 1.f;

    const int tiles_x = (a.W + ST_TW - 1) / ST_TW, tiles_y = (a.H + ST_TH - 1) / ST_TH;
    const int per_img = tiles_x * tiles_y, ntiles = a.N * per_img;
    __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)a.in, 0, 0x80000000u, 0x00020000);
    __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc((void *)(a.res ? a.res : a.in), 0, 0x80000000u, 0x00020000);

    auto fetch_in = [&](int tile, char *dst) {
        const int n = tile / per_img, tr = tile - n * per_img;
        const int ty = tr / tiles_x, tx = tr - ty * tiles_x;
        const int iy0 = ty * ST_TH - 1, ix0 = tx * ST_TW - 1;
#pragma unroll
        for (int k = 0; k < (HL_INCHUNKS + ST_NW - 1) / ST_NW; ++k) {
            const int c = wave + ST_NW * k;
            if (c < HL_INCHUNKS) {
                const int q = c * 64 + lane;                     // 16-B piece: pixel q >> 2, physical chunk q & 3
                const int px = q >> 2, pc = q & 3;
                const int ry = (px * 3641) >> 16;                // px / 18 for px < 192
                const int rxx = px - ry * HL_IW;
                const int iy = iy0 + ry, ix = ix0 + rxx;
                const int sc = pc ^ (2 * ((px >> 2) & 1));       // source chunk that belongs in this physical slot
                const bool ok = px < HL_INPIX && tile < ntiles && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                const unsigned off = ok ? (unsigned)(((n * a.H + iy) * a.W + ix) * a.in_stride + sc * 8) * 2u : 0x80000000u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (st_lds_void *)(dst + c * 1024), 16, off, 0, 0, 0);
            }
        }
    };
    auto fetch_res = [&](int tile, char *dst) {                  // [128 px][8 pieces], linear
        const int n = tile / per_img, tr = tile - n * per_img;
        const int ty = tr / tiles_x, tx = tr - ty * tiles_x;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int q = (wave + ST_NW * k) * 64 + lane;
            const int px = q >> 3, pc = q & 7;
            const int oy = ty * ST_TH + (px >> 4), ox = tx * ST_TW + (px & 15);
            const bool ok = tile < ntiles && oy < a.H && ox < a.W;
            const unsigned off = ok ? (unsigned)(((n * a.H + oy) * a.W + ox) * a.res_stride + pc * 8) * 2u : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rres, (st_lds_void *)(dst + (wave + ST_NW * k) * 1024), 16, off, 0, 0, 0);
        }
    };

    // every wave's vector-memory operations per tile, in issue order: 2 (waves 0-3) or 1 input pieces and 2 shortcut pieces for the tile
    // two ahead, then (store pass) 2 stores
    const bool has_res = a.res != nullptr;
    auto do_tile = [&](int tile, int ahead_tile, char *__restrict__ in_fill, const char *__restrict__ in_cur,
                       char *__restrict__ res_fill, const char *__restrict__ lres, char *__restrict__ lo) {
        fetch_in(ahead_tile, in_fill);
        if (has_res) fetch_res(ahead_tile, res_fill);
        const int n = tile / per_img, tr = tile - n * per_img;
        const int ty = tr / tiles_x, tx = tr - ty * tiles_x;
        const int oy0 = ty * ST_TH, ox0 = tx * ST_TW;
        f32x4 acc[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int kh = t / 3, kw = t - kh * 3;
            const int p = (wave + kh) * HL_IW + l15 + kw;        // input-tile pixel of this lane for this tap
            const bf16x8 x = *(const bf16x8 *)(in_cur + p * 64 + ((lq ^ (2 * ((p >> 2) & 1))) << 4));
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[ct] = stem_mma<H16>(fw[ct][t], x, acc[ct]);
        }
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            *(uint2 *)(lo + (wave * ST_TW + l15) * ST_OPITCH + (ct * 16 + lq * 4) * 2) = stem_epi<H16>(acc[ct], bv[ct], slope);
        }
        // staged tile complete, every wave done with this tile's input; the next tile's input and this tile's shortcut (requested one
        // and two tiles ago) must have landed.  Younger than the next tile's input pieces in this wave's queue: the next tile's shortcut
        // pieces, the previous tile's two stores, the pieces requested at the top of this tile -- they may stay in flight (the first
        // tile has no stores behind it; the start-up wait below has landed its tiles anyway)
        if (has_res) { if (wave < 4) __builtin_amdgcn_s_waitcnt(0x0078); else __builtin_amdgcn_s_waitcnt(0x0077); }
        else { if (wave < 4) __builtin_amdgcn_s_waitcnt(0x0074); else __builtin_amdgcn_s_waitcnt(0x0073); }
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int it = 0; it < ST_TH * ST_TW * 8 / (64 * ST_NW); ++it) {
            const int c = tid + it * 64 * ST_NW;
            const int px = c >> 3, chunk = c & 7;
            const int oy = oy0 + (px >> 4), ox = ox0 + (px & 15);
            uint4 o = *(const uint4 *)(lo + px * ST_OPITCH + chunk * 16);
            if (a.res) {
                const uint4 r = *(const uint4 *)(lres + c * 16);
                uint32_t ov[4] = {o.x, o.y, o.z, o.w}, rv[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float lo_ = stem_lo<H16>(ov[q]) + stem_lo<H16>(rv[q]);
                    const float hi_ = stem_hi<H16>(ov[q]) + stem_hi<H16>(rv[q]);
                    ov[q] = stem_pk<H16>(lo_, hi_);
                }
                o = uint4{ov[0], ov[1], ov[2], ov[3]};
            }
            // (conv_halo_ok: the tensor is below 2 GiB.  A pixel outside the image gets the out-of-range offset: the store is ISSUED and
            // dropped by the bounds check, so that every wave issues exactly two stores per tile -- the counted wait above relies on it)
            const unsigned so = (oy < a.H && ox < a.W) ? (unsigned)((((size_t)(n * a.H + oy) * a.W + ox) * a.out_stride + chunk * 8) * 2) : 0x80000000u;
            out_store16_at(a.out, so, o.x, o.y, o.z, o.w);
        }
        // the next tile writes `lo` only after its own pre-store barrier, and re-fills this tile's input / shortcut slots at its top:
        // every thread must be past the reads above first
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
    };

    char *const inb = smem, *const resb = smem + 3 * HL_IN_BYTES, *const lo = resb + 3 * HL_RES_BYTES;
    int tile = blockIdx.x;
    const int G = gridDim.x;
    fetch_in(tile, inb); if (has_res) fetch_res(tile, resb);
    fetch_in(tile + G, inb + HL_IN_BYTES); if (has_res) fetch_res(tile + G, resb + HL_RES_BYTES);
    __builtin_amdgcn_s_waitcnt(0x0070);
    __builtin_amdgcn_s_barrier();
    for (int slot = 0; tile < ntiles; tile += G, slot = slot == 2 ? 0 : slot + 1) {
        const int fill = slot == 0 ? 2 : slot - 1;               // (slot + 2) % 3
        do_tile(tile, tile + 2 * G, inb + fill * HL_IN_BYTES, inb + slot * HL_IN_BYTES, resb + fill * HL_RES_BYTES, resb + slot * HL_RES_BYTES, lo);
    }
#endif
}

bool conv_halo_ok(const HaloArgs &a)
{
    // 32-bit buffer offsets below 0x80000000 (the out-of-range sentinel): input and shortcut windows must stay under 2 GiB
    const double px = (double)a.N * a.H * a.W;
    if (px * a.in_stride * 2.0 >= 2147483648.0 || (a.res && px * a.res_stride * 2.0 >= 2147483648.0) || px * a.out_stride * 2.0 >= 2147483648.0) return false;
    return (a.dt == DT_BF16 || a.dt == DT_F16) && a.Cin == 32 && a.Cout == 64 && a.Kpad >= 288 && (a.in_stride % 8) == 0 && a.in_stride >= 32 && (a.out_stride % 8) == 0 &&
           a.out_stride >= 64 && (!a.res || ((a.res_stride % 8) == 0 && a.res_stride >= 64));
}
hipError_t launch_conv_halo(const HaloArgs &a, hipStream_t s)
{
    if (!conv_halo_ok(a)) return hipErrorInvalidValue;
    const size_t lds = (size_t)3 * HL_IN_BYTES + 3 * HL_RES_BYTES + ST_OUT_BYTES;
    const bool h16 = a.dt == DT_F16;
    { hipError_t e = conv_opt_in_lds(h16 ? (const void *)conv_halo_c32_c64<true> : (const void *)conv_halo_c32_c64<false>, lds); if (e != hipSuccess) return e; }
    const long tiles = (long)a.N * ((a.W + ST_TW - 1) / ST_TW) * ((a.H + ST_TH - 1) / ST_TH);
    long blocks = 256; if (blocks > tiles) blocks = tiles;
    if (h16) hipLaunchKernelGGL(conv_halo_c32_c64<true>, dim3((unsigned)blocks), dim3(64 * ST_NW), lds, s, a);
    else hipLaunchKernelGGL(conv_halo_c32_c64<false>, dim3((unsigned)blocks), dim3(64 * ST_NW), lds, s, a);
    return hipGetLastError();
}

bool conv_stem_ok(const StemArgs &a)
{
    if ((double)a.N * a.H * a.W * a.in_stride * 2.0 >= 2147483648.0) return false;       // 32-bit buffer offsets, see conv_halo_ok
    return (a.dt == DT_BF16 || a.dt == DT_F16) && a.C0 == 32 && a.C1 == 64 && a.in_stride == 8 && a.Kpad0 >= 96 && a.Kpad1 >= 288 && (a.out_stride % 8) == 0 && a.out_stride >= 64 &&
           a.Ho == (a.H + 2 - 3) / 2 + 1 && a.Wo == (a.W + 2 - 3) / 2 + 1 &&
           (!a.w2 || (a.C2 == 32 && a.Kpad2 >= 64 && a.out2 && (a.out2_stride % 8) == 0 && a.out2_stride >= 32));
}

hipError_t launch_conv_stem(const StemArgs &a, hipStream_t s)
{
    if (!conv_stem_ok(a)) return hipErrorInvalidValue;
    const bool h16 = a.dt == DT_F16, u8 = a.in_u8 != nullptr;
    if (u8 && (double)a.N * a.H * a.W * 3 >= 2147483648.0) return hipErrorInvalidValue;
    const size_t ldsb = (size_t)(u8 ? 2 * ST_RAW_BYTES : 0) + 2 * ST_IN_BYTES + 2 * ST_L0_BYTES + ST_OUT_BYTES + ST_OUT2_BYTES + ST_W2_BYTES + ST_B2_BYTES + ST_B1_BYTES;
    const void *k = h16 ? (u8 ? (const void *)conv_stem_c32_c64<true, true> : (const void *)conv_stem_c32_c64<true, false>)
                        : (u8 ? (const void *)conv_stem_c32_c64<false, true> : (const void *)conv_stem_c32_c64<false, false>);
    { hipError_t e = conv_opt_in_lds(k, ldsb); if (e != hipSuccess) return e; }
    const long tiles = (long)a.N * ((a.Wo + ST_TW - 1) / ST_TW) * ((a.Ho + ST_TH - 1) / ST_TH);
    long blocks = 256; if (blocks > tiles) blocks = tiles;          // persistent: one workgroup per CU
    if (h16 && u8) hipLaunchKernelGGL((conv_stem_c32_c64<true, true>), dim3((unsigned)blocks), dim3(64 * ST_NW), ldsb, s, a);
    else if (h16) hipLaunchKernelGGL((conv_stem_c32_c64<true, false>), dim3((unsigned)blocks), dim3(64 * ST_NW), ldsb, s, a);
    else if (u8) hipLaunchKernelGGL((conv_stem_c32_c64<false, true>), dim3((unsigned)blocks), dim3(64 * ST_NW), ldsb, s, a);
    else hipLaunchKernelGGL((conv_stem_c32_c64<false, false>), dim3((unsigned)blocks), dim3(64 * ST_NW), ldsb, s, a);
    return hipGetLastError();
}
