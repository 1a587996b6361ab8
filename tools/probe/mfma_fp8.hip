// Probe (not product code): exact-integer check of v_mfma_f32_16x16x128_f8f6f4 with e4m3 operands when every lane
// takes its 32 operand bytes as "row = lane & 15, bytes [32*(lane>>4), +32) of that row's 128-byte K slice" for BOTH
// operands, with scale arguments 0 and 127; plus the back-to-back issue rate.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cmath>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k_mm(const uint8_t *A, const uint8_t *B, float *D, int scale)
{
    const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
    v8i a = *(const v8i *)(A + r * 128 + g * 32);
    v8i b = *(const v8i *)(B + r * 128 + g * 32);
    f32x4 c = {0, 0, 0, 0};
    if (scale == 0) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0, 0, 0);
    else c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 127, 0, 127);
    // C/D: col = lane & 15 (B row index), row = 4*(lane>>4) + reg (A row index)
    for (int q = 0; q < 4; ++q) D[(4 * g + q) * 16 + r] = c[q];
}

__global__ void k_rate(float *out, long long *cyc, int iters)
{
    v8i a, b;
    for (int i = 0; i < 8; ++i) { a[i] = 0x38383838 + threadIdx.x; b[i] = 0x38383838; }
    f32x4 c[8];
    for (int i = 0; i < 8; ++i) c[i] = f32x4{0, 0, 0, 0};
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c[i], 0, 0, 0, 0, 0, 0);
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

static uint8_t enc_e4m3(int v)      // small integers only (|v| <= 8): exact
{
    if (v == 0) return 0;
    uint8_t s = v < 0 ? 0x80 : 0; int m = abs(v), e = 0;
    while ((1 << (e + 1)) <= m) ++e;               // m in [2^e, 2^(e+1))
    int frac = ((m << 3) >> e) & 7;                // 3 mantissa bits; exact for m <= 15 when low bits vanish
    return s | (uint8_t)((e + 7) << 3) | (uint8_t)frac;
}

int main()
{
    uint8_t hA[16 * 128], hB[16 * 128]; int iA[16 * 128], iB[16 * 128];
    srand(1);
    for (int i = 0; i < 16 * 128; ++i) { iA[i] = rand() % 9 - 4; iB[i] = rand() % 7 - 3; hA[i] = enc_e4m3(iA[i]); hB[i] = enc_e4m3(iB[i]); }
    uint8_t *dA, *dB; float *dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, 256 * 4);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    for (int scale = 0; scale < 2; ++scale) {
        k_mm<<<1, 64>>>(dA, dB, dD, scale);
        float hD[256]; hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
        int bad = 0; double ratio = 0;
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                int ref = 0;
                for (int k = 0; k < 128; ++k) ref += iA[i * 128 + k] * iB[j * 128 + k];
                if (hD[i * 16 + j] != (float)ref) { ++bad; if (ref) ratio = hD[i * 16 + j] / ref; }
            }
        printf("scale_arg=%d: mismatches %d of 256 (sample got/ref ratio %g)\n", scale ? 127 : 0, bad, ratio);
    }
    // internal precision: one product of 2^16 plus 127 equal products of 2^-t in the same instruction
    for (int t = 0; t <= 12; ++t) {
        uint8_t a[16 * 128] = {0}, b[16 * 128] = {0};
        auto pow2 = [](int e) { return (uint8_t)((e + 7) << 3); };       // e4m3 code of 2^e, -6 <= e <= 8
        const int ea = -(t / 2), eb = -(t - t / 2);
        for (int k = 0; k < 128; ++k) { a[k] = k ? pow2(ea) : pow2(8); b[k] = k ? pow2(eb) : pow2(8); }
        hipMemcpy(dA, a, sizeof a, hipMemcpyHostToDevice); hipMemcpy(dB, b, sizeof b, hipMemcpyHostToDevice);
        k_mm<<<1, 64>>>(dA, dB, dD, 0);
        float hD[256]; hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
        const double exact = 65536.0 + 127.0 * ldexp(1.0, -t);
        printf("2^16 + 127 x 2^-%-2d: device %.6f  fp32(exact) %.6f  lost %.6f\n", t, hD[0], (double)(float)exact, (double)(float)exact - hD[0]);
    }
    // alignment window: one product of 2^E and 127 products of 2^-6 (E - (-6) = distance in bits)
    for (int E = 2; E <= 16; E += 1) {
        uint8_t a[16 * 128] = {0}, b[16 * 128] = {0};
        auto pow2 = [](int e) { return (uint8_t)((e + 7) << 3); };
        for (int k = 0; k < 128; ++k) { a[k] = k ? pow2(-3) : pow2(E / 2); b[k] = k ? pow2(-3) : pow2(E - E / 2); }
        hipMemcpy(dA, a, sizeof a, hipMemcpyHostToDevice); hipMemcpy(dB, b, sizeof b, hipMemcpyHostToDevice);
        k_mm<<<1, 64>>>(dA, dB, dD, 0);
        float hD[256]; hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
        const double big = ldexp(1.0, E), small = ldexp(1.0, -6);
        printf("2^%-2d + 127 x 2^-6 (distance %2d bits): small terms counted %.2f of 127\n", E, E + 6, (hD[0] - big) / small);
    }
    float *dO; long long *dC, hC; hipMalloc(&dO, 1024 * 256 * 4); hipMalloc(&dC, 8);
    for (int rep = 0; rep < 2; ++rep) {
        k_rate<<<1024, 256>>>(dO, dC, 1000);
        hipMemcpy(&hC, dC, 8, hipMemcpyDeviceToHost);
        printf("16x16x128 e4m3: %.2f cycles per MFMA per wave (one wave per SIMD, 8 independent accumulators)\n", hC / 8000.0);
    }
    return 0;
}
