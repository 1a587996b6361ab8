// Probe (MI355X): how many bytes per cycle can one CU take in from L2-resident data (a filter block every workgroup re-reads), by path?
//   dma   buffer_load_dwordx4 ... lds  (1 KiB per wave instruction, the conv kernels' filter path)
//   reg   global_load_dwordx4 into registers (MFMA-fragment-order filters would be read like this, 1 KiB per wave instruction)
// One 512-thread workgroup per CU, every workgroup streams the same S bytes R times (S = 256 KiB .. 2 MiB: L2 / Infinity-Cache resident),
// DEPTH loads in flight per wave.  Prints GB/s over the chip and bytes / cycle / CU at the clock measured with s_memtime / wall time.
//   hipcc --offload-arch=gfx950 -O3 -o ingest.bin ingest.hip && ./ingest.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int DEPTH>
__global__ __launch_bounds__(512) void k_dma(const char *src, size_t bytes, int reps, unsigned long long *cyc, unsigned *sink)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, (unsigned)bytes, 0x00020000);
    const unsigned pieces = (unsigned)(bytes / 1024);                 // 1 KiB pieces, piece p by wave p % 8
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r)
        for (unsigned p = wave; p < pieces; p += 8 * DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)(smem + (wave * DEPTH + d) * 1024), 16, (p + 8 * d) * 1024 + lane * 16, 0, 0, 0);
            __builtin_amdgcn_s_waitcnt(0x0f70);                       // vmcnt(0)
        }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    if (tid == 0) { cyc[blockIdx.x] = t1 - t0; sink[blockIdx.x] = *(unsigned *)(smem + 64); }
}

template <int DEPTH>
__global__ __launch_bounds__(512) void k_reg(const char *src, size_t bytes, int reps, unsigned long long *cyc, unsigned *sink)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned pieces = (unsigned)(bytes / 1024);
    u32x4 acc = {0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r)
        for (unsigned p = wave; p < pieces; p += 8 * DEPTH) {
            u32x4 v[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) v[d] = __builtin_nontemporal_load((const u32x4 *)(src + (size_t)(p + 8 * d) * 1024 + lane * 16));
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) acc ^= v[d];
        }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) sink[blockIdx.x] = 1;
}

template <int DEPTH>
__global__ __launch_bounds__(512) void k_reg_plain(const char *src, size_t bytes, int reps, unsigned long long *cyc, unsigned *sink)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned pieces = (unsigned)(bytes / 1024);
    u32x4 acc = {0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r)
        for (unsigned p = wave; p < pieces; p += 8 * DEPTH) {
            u32x4 v[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) v[d] = *(const u32x4 *)(src + (size_t)(p + 8 * d) * 1024 + lane * 16);
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) acc ^= v[d];
        }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) sink[blockIdx.x] = 1;
}

template <typename K>
static void run(const char *name, K kern, size_t lds, const char *d, size_t bytes, int reps, int blocks, unsigned long long *dc, unsigned *ds)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds, 0, d, bytes, 2, dc, ds);          // warm
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds, 0, d, bytes, reps, dc, ds);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c(blocks);
    hipMemcpy(c.data(), dc, blocks * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : c) mean += (double)v; mean /= blocks;
    const double total = (double)bytes * reps * blocks;
    // s_memtime ticks at 100 MHz on this part: report the wall-clock rate and bytes per shader cycle at 2.1 GHz
    printf("%-14s S %7zu KiB x %3d reps: %7.3f ms  %8.1f GB/s chip  %6.1f B/cycle/CU at 2.1 GHz  (memtime ticks %.0f)\n", name, bytes / 1024, reps, ms,
           total / ms / 1e6, total / blocks / (ms * 1e-3 * 2.1e9), mean);
}

int main(int argc, char **argv)
{
    const int blocks = 256;
    char *d; unsigned long long *dc; unsigned *ds;
    const size_t maxb = 512u << 20;
    hipMalloc(&d, maxb); hipMemset(d, 1, maxb); hipMalloc(&dc, blocks * 8); hipMalloc(&ds, blocks * 4);
    for (size_t kb : {256, 4096, 16384, 65536, 524288}) {
        const size_t bytes = kb * 1024; const int reps = bytes >= (64u << 20) ? 1 : (int)((64u << 20) / bytes);
        run("dma depth 4", k_dma<4>, 8 * 4 * 1024, d, bytes, reps, blocks, dc, ds);
        run("dma depth 8", k_dma<8>, 8 * 8 * 1024, d, bytes, reps, blocks, dc, ds);
        run("reg depth 8", k_reg_plain<8>, 0, d, bytes, reps, blocks, dc, ds);
    }
    return 0;
}
