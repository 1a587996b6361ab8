// Probe (MI355X): what does an out-of-range lane of `buffer_load_dwordx4 ... offen lds` leave in LDS?
// Expected by the conv kernel: zeros (hardware zero-fill of padding taps).  Prints "zero-fill" or "untouched".
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
__global__ void k(const char *a, uint4 *o, int nbytes, int soff)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int tid = threadIdx.x;
    int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a, 0, nbytes, 0x00020000);
    *(uint4 *)(smem + tid * 16) = uint4{0xdeadbeef, 0xdeadbeef, 0xdeadbeef, 0xdeadbeef};
    __syncthreads();
    unsigned voff = (tid & 1) ? 0x80000000u : (unsigned)(tid * 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void *)(smem + wave * 1024), 16, voff, soff, 0, 0);
    __syncthreads();
    o[tid] = *(uint4 *)(smem + tid * 16);
}
int main()
{
    const int n = 256;
    std::vector<unsigned> h(n * 4 + 64);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x1000 + i;
    char *d; uint4 *o;
    hipMalloc(&d, h.size() * 4); hipMalloc(&o, n * 16);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(n), n * 16, 0, d, o, n * 16, 16);
    std::vector<unsigned> r(n * 4);
    hipMemcpy(r.data(), o, n * 16, hipMemcpyDeviceToHost);
    int ok_in = 0, zero = 0, poison = 0, other = 0;
    for (int t = 0; t < n; ++t) {
        unsigned v = r[t * 4];
        if (t & 1) { if (v == 0) zero++; else if (v == 0xdeadbeef) poison++; else other++; }
        else { if (v == 0x1000u + t * 4 + 4) ok_in++; else other++; }   // soffset 16 B = 4 words
    }
    printf("in-range ok %d/128, out-of-range: zero %d poison %d, other %d -> %s\n", ok_in, zero, poison, other,
           zero == 128 ? "zero-fill" : (poison == 128 ? "untouched" : "mixed"));
    // range check: last 16 bytes in range with soffset? voffset+16 > nbytes - soffset
    return 0;
}
