// 3x3 / stride 2 / pad 1 conv, 64 -> 128 channels (+ bias, leaky / linear), for gfx950: darknet-53's second downsampling layer (DN cfg
// layer 5, V3/yolo_v3.py:24: 208 x 208 x 64 -> 104 x 104 x 128 at 416 x 416).  K = 576 is nine K-steps per tile in the tiled kernel: a
// tile's fixed cost weighs as much as its K loop, one K-step of requests is all a workgroup keeps in flight, and the layer streams 177 MB
// in and 88 MB out (74 us at 416 x 416 x 32 against an HBM floor of 48; 46 us at 608 x 608 x 8, where the round trip is the bound).
// Here -- the structure of conv_halo_c32_c64 (conv_stem.hip) at stride 2 -- a persistent workgroup owns 8 x 8 output pixels at a time:
// the 17 x 17 x 64 input window comes into LDS ONCE by LDS-DMA, TWO tiles ahead (a ring of three 41-KB slots: 82 KB in flight per CU),
// the nine taps are ds_read_b128 at shifted addresses against REGISTER-resident filters (a wave keeps 32 output channels x 576 K = 144
// registers for the life of the workgroup), and one counted vmcnt per tile waits for the requests of the tile before only.
//
//   waves      8 = 4 channel groups (32 output channels) x 2 pixel halves (sub-tiles 0-1 / 2-3; a sub-tile = two output rows of 8 pixels)
//   K order    tap-major, the two 32-channel halves of a tap in turn: the order of the tiled kernel's K-steps -> bit-identical results
//   LDS        window pixel records of 144 B (64 channels + 16 B pad), the 16-B piece index XOR ((window row >> 1) & 1): the 16 lanes of a
//              fragment read (two output rows, pixel stride 2) then fall on 16 distinct bank slots for every tap.  The LDS-DMA writes
//              lane-linear 1-KiB pieces; the layout comes from permuting which global piece each lane fetches (the pad piece and
//              everything outside the image are out-of-range requests: zeros).
//   epilogue   bias, activation, rounding -> LDS (272-B rows) -> 16-byte pieces -> global (write-through, kernels.h OUT_STORE_AUX)
#include "kernels.h"

typedef __bf16 s2_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 s2_f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 s2_bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 s2_f16x2 __attribute__((ext_vector_type(2)));
typedef float s2_f32x2 __attribute__((ext_vector_type(2)));
typedef float s2_f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void s2_lds_void;

template <bool H16> __device__ __forceinline__ uint32_t s2_pk(float lo, float hi)
{
    if constexpr (H16) return __builtin_bit_cast(uint32_t, __builtin_convertvector(s2_f32x2{lo, hi}, s2_f16x2));      // (MODE.FP16_OVFL: saturating)
    else return __builtin_bit_cast(uint32_t, __builtin_convertvector(s2_f32x2{lo, hi}, s2_bf16x2));
}
__device__ __forceinline__ float s2_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
template <bool H16> __device__ __forceinline__ uint2 s2_epi(const s2_f32x4 acc, const s2_f32x4 bias, const float slope)
{
    s2_f32x4 v = acc + bias;
    const s2_f32x4 t = v * slope;
    return uint2{s2_pk<H16>(s2_max(v[0], t[0]), s2_max(v[1], t[1])), s2_pk<H16>(s2_max(v[2], t[2]), s2_max(v[3], t[3]))};
}
template <bool H16> __device__ __forceinline__ s2_f32x4 s2_mma(const s2_bf16x8 a, const s2_bf16x8 b, const s2_f32x4 c)
{
    if constexpr (H16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(s2_f16x8, a), __builtin_bit_cast(s2_f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

constexpr int S2_T = 8;                                  // output tile edge
constexpr int S2_WIN = 2 * S2_T + 1;                     // input window edge: 17
constexpr int S2_WPIX = S2_WIN * S2_WIN;                 // 289
constexpr int S2_PITCH = 144;                            // bytes of one window pixel record: 8 data pieces + 1 pad piece
constexpr int S2_PIECES = S2_WPIX * (S2_PITCH / 16);     // 2601 16-byte pieces
constexpr int S2_CHUNKS = (S2_PIECES + 63) / 64;         // 41 LDS-DMA instructions per tile (wave 0: 6, waves 1-7: 5)
constexpr int S2_IN_BYTES = S2_CHUNKS * 1024;            // 41984
constexpr int S2_OPITCH = 128 * 2 + 16;                  // staged output rows
constexpr int S2_OUT_BYTES = S2_T * S2_T * S2_OPITCH;    // 17408
constexpr int S2_NW = 8;
constexpr int S2_KMAX = (S2_CHUNKS + S2_NW - 1) / S2_NW; // 6 (device code only)
constexpr size_t S2_LDS = (size_t)3 * S2_IN_BYTES + S2_OUT_BYTES;      // 143360

template <bool H16>
__global__ __launch_bounds__(64 * S2_NW) void conv_s2_c64_c128(const HaloArgs a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if constexpr (H16) __builtin_amdgcn_s_setreg((0 << 11) | (23 << 6) | 1, 1);      // hwreg(HW_REG_MODE, 23, 1) = FP16_OVFL: fp16 conversions saturate
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const int grp = wave & 3, half = wave >> 2;                  // channel group (32 output channels), pixel half (sub-tiles 2 half, 2 half + 1)
    // this wave's filters: 2 channel tiles x 18 K-slices (tap * 2 + channel half), rows K-contiguous with k = tap * 64 + c
    s2_bf16x8 fw[2][18];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int ks = 0; ks < 18; ++ks)
            fw[ct][ks] = *(const s2_bf16x8 *)((const bf16_t *)a.w + (size_t)(grp * 32 + ct * 16 + l15) * a.Kpad + ks * 32 + lq * 8);
    s2_f32x4 bv[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) bv[ct] = *(const s2_f32x4 *)(a.b + grp * 32 + ct * 16 + lq * 4);
    const float slope = a.act == ACT_LEAKY ? 0.1f : 1.f;

    const int Ho = (a.H - 1) / 2 + 1, Wo = (a.W - 1) / 2 + 1;
    const int tiles_x = (Wo + S2_T - 1) / S2_T, tiles_y = (Ho + S2_T - 1) / S2_T;
    const int per_img = tiles_x * tiles_y, ntiles = a.N * per_img;
    __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)a.in, 0, 0x80000000u, 0x00020000);

    // the pieces this lane requests for every tile: LDS slot g = (wave + 8 k) * 64 + lane is piece g % 9 of window pixel g / 9
    unsigned rel[S2_KMAX]; int wyx[S2_KMAX];                     // byte offset from the window's first pixel; window row | column << 8 (-1: no request)
#pragma unroll
    for (int k = 0; k < S2_KMAX; ++k) {
        const int g = (wave + S2_NW * k) * 64 + lane;
        const int px = g / 9, j = g - px * 9;
        const int wy = px / S2_WIN, wx = px - wy * S2_WIN;
        const bool valid = px < S2_WPIX && j < 8;
        rel[k] = (unsigned)(((wy * a.W + wx) * a.in_stride + ((j ^ ((wy >> 1) & 1)) * 8)) * 2);
        wyx[k] = valid ? (wy | (wx << 8)) : -1;
    }
    auto fetch_in = [&](int tile, char *dst) {
        const int n = tile / per_img, tr = tile - n * per_img;
        const int ty = tr / tiles_x, tx = tr - ty * tiles_x;
        const int iy0 = 2 * ty * S2_T - 1, ix0 = 2 * tx * S2_T - 1;
        const unsigned base = (unsigned)(((n * a.H + iy0) * a.W + ix0) * a.in_stride * 2);      // (mod 2^32: the sum with rel is a valid offset wherever the pixel is inside the image)
#pragma unroll
        for (int k = 0; k < S2_KMAX; ++k) {
            const int c = wave + S2_NW * k;
            if (c < S2_CHUNKS) {
                const int wy = wyx[k] & 0xff, wx = (wyx[k] >> 8) & 0xff;
                const bool ok = wyx[k] >= 0 && tile < ntiles && (unsigned)(iy0 + wy) < (unsigned)a.H && (unsigned)(ix0 + wx) < (unsigned)a.W;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (s2_lds_void *)(dst + c * 1024), 16, ok ? base + rel[k] : 0x80000000u, 0, 0, 0);
            }
        }
    };

    // LDS byte offset of this lane's window pixel for tap (0, 0), per sub-tile of this wave; the tap adds (kh * 17 + kw) * 144
    int pb[2];
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) {
        const int r = 2 * (2 * half + sb) + (l15 >> 3), c = l15 & 7;
        pb[sb] = ((2 * r) * S2_WIN + 2 * c) * S2_PITCH;
    }
    const int rpar = (l15 >> 3) & 1;                             // parity of this lane's output row within the tile (sub-tiles start on even rows)

    // every wave's vector-memory operations per tile, in issue order: 6 (wave 0) or 5 window pieces for the tile two ahead, then (store
    // pass) 2 stores -- issued for pixels outside the image too, with the out-of-range offset, so that the counted wait holds
    auto do_tile = [&](int tile, int ahead_tile, char *__restrict__ in_fill, const char *__restrict__ in_cur, char *__restrict__ lo) {
        fetch_in(ahead_tile, in_fill);
        const int n = tile / per_img, tr = tile - n * per_img;
        const int ty = tr / tiles_x, tx = tr - ty * tiles_x;
        const int oy0 = ty * S2_T, ox0 = tx * S2_T;
        s2_f32x4 acc[2][2];
#pragma unroll
        for (int sb = 0; sb < 2; ++sb)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) acc[sb][ct] = s2_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int kh = t / 3, kw = t - kh * 3;
            const int swz = rpar ^ (kh >> 1);                    // ((window row) >> 1) & 1 of this lane's pixel for this tap
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int coff = ((hh * 4 + lq) ^ swz) << 4;
#pragma unroll
                for (int sb = 0; sb < 2; ++sb) {
                    const s2_bf16x8 x = *(const s2_bf16x8 *)(in_cur + pb[sb] + (kh * S2_WIN + kw) * S2_PITCH + coff);
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) acc[sb][ct] = s2_mma<H16>(fw[ct][t * 2 + hh], x, acc[sb][ct]);
                }
            }
        }
#pragma unroll
        for (int sb = 0; sb < 2; ++sb)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
                *(uint2 *)(lo + ((2 * half + sb) * 16 + l15) * S2_OPITCH + (grp * 32 + ct * 16 + lq * 4) * 2) = s2_epi<H16>(acc[sb][ct], bv[ct], slope);
        // staged tile complete, every wave done with this tile's window; the next tile's window (requested a tile ago) must have landed.
        // Younger than its pieces in this wave's queue: the previous tile's two stores and the pieces requested at the top of this tile
        if (wave == 0) __builtin_amdgcn_s_waitcnt(0x0078); else __builtin_amdgcn_s_waitcnt(0x0077);
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int it = 0; it < S2_T * S2_T * 16 / (64 * S2_NW); ++it) {
            const int c = tid + it * 64 * S2_NW;
            const int px = c >> 4, chunk = c & 15;
            const int oy = oy0 + (px >> 3), ox = ox0 + (px & 7);
            const uint4 o = *(const uint4 *)(lo + px * S2_OPITCH + chunk * 16);
            const unsigned so = (oy < Ho && ox < Wo) ? (unsigned)((((size_t)(n * Ho + oy) * Wo + ox) * a.out_stride + chunk * 8) * 2) : 0x80000000u;
            out_store16_at(a.out, so, o.x, o.y, o.z, o.w);
        }
        // the next tile writes `lo` only after its own pre-store barrier and re-fills this tile's window slot at its top
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
    };

    char *const inb = smem, *const lo = smem + 3 * S2_IN_BYTES;
    int tile = blockIdx.x;
    const int G = gridDim.x;
    fetch_in(tile, inb);
    fetch_in(tile + G, inb + S2_IN_BYTES);
    __builtin_amdgcn_s_waitcnt(0x0070);
    __builtin_amdgcn_s_barrier();
    for (int slot = 0; tile < ntiles; tile += G, slot = slot == 2 ? 0 : slot + 1) {
        const int fill = slot == 0 ? 2 : slot - 1;               // (slot + 2) % 3
        do_tile(tile, tile + 2 * G, inb + fill * S2_IN_BYTES, inb + slot * S2_IN_BYTES, lo);
    }
#endif
}

bool conv_s2_ok(const HaloArgs &a)
{
    // 32-bit buffer offsets below 0x80000000 (the out-of-range sentinel): input and output windows must stay under 2 GiB
    const double pin = (double)a.N * a.H * a.W, pout = (double)a.N * ((a.H - 1) / 2 + 1) * ((a.W - 1) / 2 + 1);
    if (pin * a.in_stride * 2.0 >= 2147483648.0 || pout * a.out_stride * 2.0 >= 2147483648.0) return false;
    return (a.dt == DT_BF16 || a.dt == DT_F16) && a.Cin == 64 && a.Cout == 128 && a.Kpad >= 576 && !a.res && a.H >= 2 && a.W >= 2 &&
           (a.in_stride % 8) == 0 && a.in_stride >= 64 && (a.out_stride % 8) == 0 && a.out_stride >= 128;
}
hipError_t launch_conv_s2(const HaloArgs &a, hipStream_t s)
{
    if (!conv_s2_ok(a)) return hipErrorInvalidValue;
    const bool h16 = a.dt == DT_F16;
    { hipError_t e = conv_opt_in_lds(h16 ? (const void *)conv_s2_c64_c128<true> : (const void *)conv_s2_c64_c128<false>, S2_LDS); if (e != hipSuccess) return e; }
    const int Ho = (a.H - 1) / 2 + 1, Wo = (a.W - 1) / 2 + 1;
    const long tiles = (long)a.N * ((Wo + S2_T - 1) / S2_T) * ((Ho + S2_T - 1) / S2_T);
    long blocks = 256; if (blocks > tiles) blocks = tiles;
    if (h16) hipLaunchKernelGGL(conv_s2_c64_c128<true>, dim3((unsigned)blocks), dim3(64 * S2_NW), S2_LDS, s, a);
    else hipLaunchKernelGGL(conv_s2_c64_c128<false>, dim3((unsigned)blocks), dim3(64 * S2_NW), S2_LDS, s, a);
    return hipGetLastError();
}
