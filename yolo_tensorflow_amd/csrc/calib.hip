// yolo_calibrate: what THIS chip sustains on a register-resident MFMA loop, and at which clock.
//
// The same binary reads 12.3 - 13.5 k img/s by box (profiles/README.md): a driver line from one box cannot be compared with a line from
// another.  This loop is the yardstick that travels with every line: 8 waves per CU (two per SIMD), each with eight independent
// 16x16 fp32 accumulators fed by `v_mfma_f32_16x16x32_bf16` (or _f16) from operands held in registers -- no LDS, no memory, nothing but the
// matrix pipe -- on random operands (the data toggles as the conv's do: DVFS answers to that), launched back to back for the requested
// time.  Reported: the TFLOP/s of the launches after the ramp (HIP events around the second half), and the shader clock the waves held
// (s_memtime / s_memrealtime, the latter a constant 100 MHz; median over waves of the last launch).
// `roofline.frac / (calib_tflops / peak)` is then comparable across boxes.  Nothing of the reference corresponds to it.
#include "kernels.h"
#include "../../include/yolo_hip.h"
#include <algorithm>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned cal_u32x4;      // eight 16-bit operands of one MFMA fragment
typedef __attribute__((ext_vector_type(4))) float cal_f32x4;

namespace {

constexpr int kAcc = 8;            // independent accumulator tiles per wave
constexpr int kWavesPerCU = 8;     // two per SIMD

template <bool F16>
__global__ __launch_bounds__(256) void k_mfma_calib(const cal_u32x4 *__restrict__ operands, int iters, float *__restrict__ sink, unsigned long long *__restrict__ stamps)
{
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    // four A and four B fragments per lane, random bits reinterpreted as 16-bit floats with the exponent confined to [2^-4, 2^4) by the host
    cal_u32x4 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = operands[(i * 64 + lane)]; b[i] = operands[((4 + i) * 64 + lane)]; }
    cal_f32x4 acc[kAcc];
#pragma unroll
    for (int i = 0; i < kAcc; ++i) acc[i] = cal_f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    // (inline asm with the accumulator tied in place: left to the intrinsic, hipcc rotates the eight accumulators through AGPR copies every
    // iteration -- 30 v_accvgpr moves per 8 MFMAs -- and the loop measures the copy traffic)
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < kAcc; ++i) {
            if constexpr (F16) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i & 3]), "v"(b[(i >> 1) & 3]));
            else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i & 3]), "v"(b[(i >> 1) & 3]));
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // (the compiler does not know the asm statements are MFMAs: cover the result hazard by hand)
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kAcc; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) sink[0] = s;                    // (keeps the loop alive; never true on these operands)
    if (lane == 0) { stamps[2 * wave] = t1 - t0; stamps[2 * wave + 1] = r1 - r0; }
}

}  // namespace

// The memory side of the same yardstick: a streaming copy (16 bytes per lane in, 16 out, grid-stride over the chip) between two buffers
// far larger than the 256 MiB Infinity Cache, launched back to back for the requested time -> GB/s (read + written bytes).  A conv stack is
// part matrix-pipe-bound and part traffic-bound, and the boxes of one pool differ in both (round 6: the box with the LOWER MFMA-loop clock
// read the HIGHER img/s): the pair (calib_tflops, calib_copy_gbs) is what identifies a box class.
namespace {
__global__ __launch_bounds__(256) void k_copy_calib(const uint4 *__restrict__ in, uint4 *__restrict__ out, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = in[i];
}
}  // namespace

extern "C" int yolo_calibrate_copy(int device, void *stream_, double seconds, float *gbs)
{
    if (seconds <= 0 || seconds > 10 || !gbs) return YOLO_ERR_INVALID;
    if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return YOLO_ERR_HIP; }
    hipStream_t stream = (hipStream_t)stream_;
    const size_t bytes = (size_t)1 << 30, n = bytes / 16;            // 1 GiB in, 1 GiB out
    uint4 *a = nullptr, *b = nullptr; hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
    int rc = YOLO_ERR_HIP;
    do {
        if (hipMalloc((void **)&a, bytes) != hipSuccess || hipMalloc((void **)&b, bytes) != hipSuccess) { rc = YOLO_ERR_NOMEM; break; }
        if (hipMemsetAsync(a, 1, bytes, stream) != hipSuccess) break;
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess || hipEventCreate(&e2) != hipSuccess) break;
        auto launch = [&]() { hipLaunchKernelGGL(k_copy_calib, dim3(256 * 8), dim3(256), 0, stream, a, b, n); };
        (void)hipEventRecord(e0, stream); launch(); (void)hipEventRecord(e1, stream);
        if (hipEventSynchronize(e1) != hipSuccess) break;
        float one = 0; (void)hipEventElapsedTime(&one, e0, e1);
        const int reps = std::max(2, (int)(seconds * 1e3 / std::max(one, 0.05f)));
        (void)hipEventRecord(e1, stream);
        for (int i = 0; i < reps; ++i) launch();
        (void)hipEventRecord(e2, stream);
        if (hipEventSynchronize(e2) != hipSuccess) break;
        float ms = 0; (void)hipEventElapsedTime(&ms, e1, e2);
        *gbs = (float)(2.0 * (double)bytes * reps / (ms * 1e-3) / 1e9);
        rc = YOLO_OK;
    } while (0);
    if (rc != YOLO_OK) (void)hipGetLastError();
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (e2) (void)hipEventDestroy(e2);
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    return rc;
}

extern "C" int yolo_calibrate(int device, void *stream_, int f16, double seconds, float *tflops, float *clock_ghz)
{
    if (seconds <= 0 || seconds > 10) return YOLO_ERR_INVALID;
    if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return YOLO_ERR_HIP; }
    hipStream_t stream = (hipStream_t)stream_;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { (void)hipGetLastError(); return YOLO_ERR_HIP; }
    const int cus = prop.multiProcessorCount, blocks = cus * (kWavesPerCU / 4), waves = blocks * 4;
    // random operands: sign, a 3-bit window of the exponent around 1.0, random significand (no NaN / Inf / subnormals; products stay far
    // inside fp32 whatever the iteration count, because half of the terms are negative and the accumulators random-walk)
    std::vector<uint16_t> h(8 * 64 * 8);
    uint64_t x = 0x9E3779B97F4A7C15ull;
    for (auto &v : h) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        const unsigned sign = (x >> 40) & 1, e3 = (x >> 20) & 7, man = (unsigned)(x & 0x3FF);
        v = f16 ? (uint16_t)((sign << 15) | ((11 + e3) << 10) | man)                 // fp16: bias 15
                : (uint16_t)((sign << 15) | ((123 + e3) << 7) | (man >> 3));         // bf16: bias 127
    }
    cal_u32x4 *d_op = nullptr; float *d_sink = nullptr; unsigned long long *d_st = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
    int rc = YOLO_ERR_HIP;
    std::vector<unsigned long long> st((size_t)waves * 2);
    do {
        if (hipMalloc((void **)&d_op, h.size() * 2) != hipSuccess || hipMalloc((void **)&d_sink, 64) != hipSuccess ||
            hipMalloc((void **)&d_st, st.size() * 8) != hipSuccess) break;
        if (hipMemcpyAsync(d_op, h.data(), h.size() * 2, hipMemcpyHostToDevice, stream) != hipSuccess) break;
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess || hipEventCreate(&e2) != hipSuccess) break;
        const int iters = 20000;                                   // 8 MFMAs x 16 cycles x 2 waves per SIMD x 20000 = 5.1 M cycles: ~2.5 ms per launch
        auto launch = [&]() {
            if (f16) hipLaunchKernelGGL(k_mfma_calib<true>, dim3(blocks), dim3(256), 0, stream, d_op, iters, d_sink, d_st);
            else hipLaunchKernelGGL(k_mfma_calib<false>, dim3(blocks), dim3(256), 0, stream, d_op, iters, d_sink, d_st);
        };
        // one launch to size the run, then half of the time as ramp and the other half measured
        (void)hipEventRecord(e0, stream); launch(); (void)hipEventRecord(e1, stream);
        if (hipEventSynchronize(e1) != hipSuccess) break;
        float one = 0; (void)hipEventElapsedTime(&one, e0, e1);
        const int n = std::max(2, (int)(seconds * 1e3 / std::max(one, 0.05f)) / 2);
        for (int i = 0; i < n; ++i) launch();
        (void)hipEventRecord(e1, stream);
        for (int i = 0; i < n; ++i) launch();
        (void)hipEventRecord(e2, stream);
        if (hipEventSynchronize(e2) != hipSuccess) break;
        float ms = 0; (void)hipEventElapsedTime(&ms, e1, e2);
        if (hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) break;
        const double flop = (double)n * waves * (double)iters * kAcc * (2.0 * 16 * 16 * 32);
        if (tflops) *tflops = (float)(flop / (ms * 1e-3) / 1e12);
        std::vector<double> ghz;
        for (int w = 0; w < waves; ++w) if (st[2 * w + 1]) ghz.push_back((double)st[2 * w] / (double)st[2 * w + 1] * 0.1);
        std::sort(ghz.begin(), ghz.end());
        if (clock_ghz) *clock_ghz = ghz.empty() ? 0.f : (float)ghz[ghz.size() / 2];
        rc = YOLO_OK;
    } while (0);
    if (rc != YOLO_OK) (void)hipGetLastError();
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (e2) (void)hipEventDestroy(e2);
    if (d_op) (void)hipFree(d_op);
    if (d_sink) (void)hipFree(d_sink);
    if (d_st) (void)hipFree(d_st);
    return rc;
}
