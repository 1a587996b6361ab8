// Probe: cost of one dependent kernel launch in a stream vs replayed from a HIP graph (empty kernel, and a kernel that
// keeps every CU busy for ~10 us), MI355X.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_empty() {}
__global__ void k_busy(float *p, int iters)
{
    float v = threadIdx.x;
    for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
    if (v == 12345.f) p[0] = v;
}
int main()
{
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float *d; hipMalloc(&d, 4);
    const int N = 100;
    for (int busy = 0; busy < 2; ++busy) {
        auto body = [&]() { for (int i = 0; i < N; ++i) { if (busy) k_busy<<<256, 256, 0, s>>>(d, 2000); else k_empty<<<1, 64, 0, s>>>(); } };
        body(); hipStreamSynchronize(s);
        hipEventRecord(e0, s); for (int r = 0; r < 10; ++r) body(); hipEventRecord(e1, s); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%s kernel, stream launches: %.2f us per launch\n", busy ? "busy " : "empty", ms * 1000 / (10 * N));
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal); body(); hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, s); hipStreamSynchronize(s);
        hipEventRecord(e0, s); for (int r = 0; r < 10; ++r) hipGraphLaunch(ge, s); hipEventRecord(e1, s); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("%s kernel, graph replay   : %.2f us per launch\n", busy ? "busy " : "empty", ms * 1000 / (10 * N));
        // single launch of the busy kernel for reference
        if (busy) {
            hipEventRecord(e0, s); k_busy<<<256, 256, 0, s>>>(d, 2000); hipEventRecord(e1, s); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1); printf("busy kernel alone (event pair around one launch): %.2f us\n", ms * 1000);
        }
    }
    return 0;
}
