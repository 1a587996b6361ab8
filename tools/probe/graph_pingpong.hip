// Probe (MI355X, ROCm 7): does hipGraphLaunch keep the host until the stream's previous graph replay has finished -- and does alternating
// between two streams (chained by events, so the replays still run one after the other on the GPU) let the host run a replay ahead?
// Each graph = NK launches of a ~30 us spin kernel (a stand-in for the 66 launches of a detect step).  Prints the wall time per replay and the
// host time spent inside hipGraphLaunch for: one stream / one exec; one stream / two execs; two streams / two execs with event chaining.
//   hipcc --offload-arch=gfx950 -O3 -o graph_pingpong.bin graph_pingpong.hip && ./graph_pingpong.bin
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
struct Big { unsigned long long cycles; unsigned *sink; char pad[240]; };
__global__ void spin_big(const Big b)
{
    extern __shared__ char smem[];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < b.cycles) { }
    if (b.sink && threadIdx.x == 9999) *b.sink = smem[b.pad[0]];
}
__global__ void spin(unsigned long long cycles, unsigned *sink)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < cycles) { }
    if (sink && threadIdx.x == 9999) *sink = 1;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const int NK = 66, REPS = 200;
    hipStream_t s[2]; CK(hipStreamCreate(&s[0])); CK(hipStreamCreate(&s[1]));
    char *dbuf; CK(hipMalloc((void **)&dbuf, 4096));
    CK(hipFuncSetAttribute((const void *)spin_big, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    hipGraphExec_t ex[2];
    for (int g = 0; g < 2; ++g) {
        hipGraph_t gr;
        CK(hipStreamBeginCapture(s[g], hipStreamCaptureModeThreadLocal));
        if (getenv("PP_MEMSET")) CK(hipMemsetAsync(dbuf, 0, 64, s[g]));
        for (int k = 0; k < NK; ++k) {
            if (getenv("PP_BIG")) { Big b; memset(&b, 0, sizeof b); b.cycles = 60000ull; hipLaunchKernelGGL(spin_big, dim3(256), dim3(256), getenv("PP_LDS") ? 100 * 1024 : 0, s[g], b); }
            else hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s[g], 60000ull, (unsigned *)nullptr);
        }
        if (getenv("PP_MEMCPY")) CK(hipMemcpyAsync(dbuf + 64, dbuf, 64, hipMemcpyDeviceToDevice, s[g]));
        CK(hipStreamEndCapture(s[g], &gr));
        CK(hipGraphInstantiate(&ex[g], gr, nullptr, nullptr, 0));
        CK(hipGraphDestroy(gr));
    }
    hipEvent_t ev[2]; CK(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
    for (int mode = 0; mode < 3; ++mode) {
        for (int w = 0; w < 3; ++w) { CK(hipGraphLaunch(ex[0], s[0])); CK(hipGraphLaunch(ex[1], s[1])); }
        CK(hipDeviceSynchronize());
        double host = 0; const double t0 = now();
        for (int r = 0; r < REPS; ++r) {
            const int g = mode == 0 ? 0 : (r & 1), st = mode == 2 ? (r & 1) : 0;
            if (mode == 2 && r > 0) CK(hipStreamWaitEvent(s[st], ev[st ^ 1], 0));
            const double h0 = now();
            CK(hipGraphLaunch(ex[g], s[st]));
            host += now() - h0;
            if (mode == 2) CK(hipEventRecord(ev[st], s[st]));
        }
        CK(hipDeviceSynchronize());
        const double t1 = now();
        printf("%-44s wall %.3f ms / replay, host inside hipGraphLaunch %.3f ms / replay\n",
               mode == 0 ? "one stream, one exec" : mode == 1 ? "one stream, two execs alternating" : "two streams + events, two execs alternating", (t1 - t0) / REPS * 1e3, host / REPS * 1e3);
    }
    {   // the NULL stream (what torch's default stream is)
        CK(hipDeviceSynchronize());
        double host = 0; const double t0 = now();
        for (int r = 0; r < REPS; ++r) { const double h0 = now(); CK(hipGraphLaunch(ex[0], nullptr)); host += now() - h0; }
        CK(hipDeviceSynchronize());
        printf("%-44s wall %.3f ms / replay, host inside hipGraphLaunch %.3f ms / replay\n", "NULL stream, one exec", (now() - t0) / REPS * 1e3, host / REPS * 1e3);
    }
    {   // one stream, one exec, a timing event recorded after every replay (what a per-step latency probe does)
        hipEvent_t te[REPS + 1]; for (auto &e : te) CK(hipEventCreate(&e));
        CK(hipDeviceSynchronize());
        const double t0 = now();
        CK(hipEventRecord(te[0], s[0]));
        for (int r = 0; r < REPS; ++r) { CK(hipGraphLaunch(ex[0], s[0])); CK(hipEventRecord(te[r + 1], s[0])); }
        CK(hipDeviceSynchronize());
        printf("%-44s wall %.3f ms / replay\n", "one stream, timing event after every replay", (now() - t0) / REPS * 1e3);
    }
    // the same work launched eagerly on one stream
    for (int w = 0; w < 2; ++w) for (int k = 0; k < NK; ++k) hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s[0], 60000ull, (unsigned *)nullptr);
    CK(hipDeviceSynchronize());
    const double t0 = now();
    for (int r = 0; r < REPS; ++r) for (int k = 0; k < NK; ++k) hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s[0], 60000ull, (unsigned *)nullptr);
    CK(hipDeviceSynchronize());
    printf("%-44s wall %.3f ms / replay\n", "eager launches, one stream", (now() - t0) / REPS * 1e3);
    return 0;
}
