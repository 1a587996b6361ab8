// HBM-bound NHWC operators around the conv stack: input conversion / stretch-resize (row P), 2x
// upsample (row U), max-pool (row M), space-to-depth / darknet reorg (row R), residual add and strided
// copies (rows B, Rt when they cannot be fused into a conv epilogue), dtype conversion for introspection.
// One thread moves an 8-channel granule (16 B of bf16, 32 B of fp32) so global accesses are 16-B vectors
// and consecutive lanes touch consecutive addresses along the channel axis.
#include "kernels.h"
#include <math.h>

template <typename T> struct Elt;
template <> struct Elt<bf16_t> {
    static __device__ __forceinline__ void load8(const bf16_t *p, float *v)
    {
        uint4 u = *(const uint4 *)p;
        uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __builtin_bit_cast(float, w[i] << 16);
            v[2 * i + 1] = __builtin_bit_cast(float, w[i] & 0xffff0000u);
        }
    }
    static __device__ __forceinline__ uint32_t cvt(float f)
    {
        __bf16 b = (__bf16)f;
        return (uint32_t)__builtin_bit_cast(uint16_t, b);
    }
    static __device__ __forceinline__ void store8(bf16_t *p, const float *v)
    {
        uint4 u;
        u.x = cvt(v[0]) | (cvt(v[1]) << 16); u.y = cvt(v[2]) | (cvt(v[3]) << 16);
        u.z = cvt(v[4]) | (cvt(v[5]) << 16); u.w = cvt(v[6]) | (cvt(v[7]) << 16);
        *(uint4 *)p = u;
    }
    static __device__ __forceinline__ float load1(const bf16_t *p) { return __builtin_bit_cast(float, (uint32_t)(*p) << 16); }
    static __device__ __forceinline__ void store1(bf16_t *p, float f) { *p = (bf16_t)cvt(f); }
};
template <> struct Elt<f16_t> {       // IEEE binary16: decode exact, encode round-to-nearest-even, saturating at +-65504 (as the conv epilogues do)
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ void load8(const f16_t *p, float *v)
    {
        uint4 u = *(const uint4 *)p;
        uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { const h2 h = __builtin_bit_cast(h2, w[i]); v[2 * i] = (float)h[0]; v[2 * i + 1] = (float)h[1]; }
    }
    static __device__ __forceinline__ uint32_t pk(float a, float b)
    {
        a = __builtin_amdgcn_fmed3f(a, -65504.f, 65504.f); b = __builtin_amdgcn_fmed3f(b, -65504.f, 65504.f);
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{a, b}, h2));
    }
    static __device__ __forceinline__ void store8(f16_t *p, const float *v)
    {
        *(uint4 *)p = uint4{pk(v[0], v[1]), pk(v[2], v[3]), pk(v[4], v[5]), pk(v[6], v[7])};
    }
    static __device__ __forceinline__ float load1(const f16_t *p) { return (float)__builtin_bit_cast(_Float16, p->b); }
    static __device__ __forceinline__ void store1(f16_t *p, float f) { p->b = (uint16_t)(pk(f, 0.f) & 0xffffu); }
};
template <> struct Elt<float> {
    static __device__ __forceinline__ void load8(const float *p, float *v)
    {
        float4 a = *(const float4 *)p, b = *(const float4 *)(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
    static __device__ __forceinline__ void store8(float *p, const float *v)
    {
        *(float4 *)p = float4{v[0], v[1], v[2], v[3]};
        *(float4 *)(p + 4) = float4{v[4], v[5], v[6], v[7]};
    }
    static __device__ __forceinline__ float load1(const float *p) { return *p; }
    static __device__ __forceinline__ void store1(float *p, float f) { *p = f; }
};

template <> struct Elt<fp8_t> {      // OCP e4m3: decode is exact, encode is round-to-nearest-even with saturation at +-448
    template <bool HI> static __device__ __forceinline__ uint32_t enc2(float a, float b, uint32_t old)
    {
        a = __builtin_amdgcn_fmed3f(a, -FP8_MAX, FP8_MAX); b = __builtin_amdgcn_fmed3f(b, -FP8_MAX, FP8_MAX);
        return (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, (int)old, HI);
    }
    static __device__ __forceinline__ void load8(const fp8_t *p, float *v)
    {
        uint2 u = *(const uint2 *)p;
        v[0] = __builtin_amdgcn_cvt_f32_fp8((int)u.x, 0); v[1] = __builtin_amdgcn_cvt_f32_fp8((int)u.x, 1);
        v[2] = __builtin_amdgcn_cvt_f32_fp8((int)u.x, 2); v[3] = __builtin_amdgcn_cvt_f32_fp8((int)u.x, 3);
        v[4] = __builtin_amdgcn_cvt_f32_fp8((int)u.y, 0); v[5] = __builtin_amdgcn_cvt_f32_fp8((int)u.y, 1);
        v[6] = __builtin_amdgcn_cvt_f32_fp8((int)u.y, 2); v[7] = __builtin_amdgcn_cvt_f32_fp8((int)u.y, 3);
    }
    static __device__ __forceinline__ void store8(fp8_t *p, const float *v)
    {
        uint2 u;
        u.x = enc2<true>(v[2], v[3], enc2<false>(v[0], v[1], 0));
        u.y = enc2<true>(v[6], v[7], enc2<false>(v[4], v[5], 0));
        *(uint2 *)p = u;
    }
    static __device__ __forceinline__ float load1(const fp8_t *p) { return __builtin_amdgcn_cvt_f32_fp8((int)p->b, 0); }
    static __device__ __forceinline__ void store1(fp8_t *p, float f) { p->b = (uint8_t)(enc2<false>(f, 0.f, 0) & 0xff); }
};

// run `stmt` with T bound to the element type of `dt`
#define WITH_DT(dt, ...)                                                         \
    do {                                                                         \
        if ((dt) == DT_F32) { typedef float T; __VA_ARGS__; }                    \
        else if ((dt) == DT_FP8) { typedef fp8_t T; __VA_ARGS__; }               \
        else if ((dt) == DT_F16) { typedef f16_t T; __VA_ARGS__; }               \
        else { typedef bf16_t T; __VA_ARGS__; }                                  \
    } while (0)

static inline dim3 grid_for(size_t n, int block = 256) { return dim3((unsigned)((n + block - 1) / block)); }

// ---- row P: uint8/float image at network size -> 8-channel (3 real + 5 zero) activation ---------
template <typename T>
__global__ void k_preprocess(const void *img, int fmt, size_t npix, size_t hw, float scale, T *out, int out_stride, float post_mul, float post_add)
{
    size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (fmt == 0) {
        const uint8_t *s = (const uint8_t *)img + p * 3;
        v[0] = (float)s[0] * scale; v[1] = (float)s[1] * scale; v[2] = (float)s[2] * scale;
    } else if (fmt == 1) {
        const float *s = (const float *)img + p * 3;
        v[0] = s[0] * scale; v[1] = s[1] * scale; v[2] = s[2] * scale;
    } else {                                   // planar float [n][3][hw]: darknet's `image` layout (DN/image.c get_pixel)
        const size_t b = p / hw, q = p - b * hw;
        const float *s = (const float *)img + b * 3 * hw + q;
        v[0] = s[0] * scale; v[1] = s[hw] * scale; v[2] = s[2 * hw] * scale;
    }
    // YOLOv1's input normalisation `(x / 255) * 2 - 1` (V1/YOLO_V1_Inference.py:67-71): an affine map of the three real channels
    if (post_mul != 1.0f || post_add != 0.0f) { v[0] = v[0] * post_mul + post_add; v[1] = v[1] * post_mul + post_add; v[2] = v[2] * post_mul + post_add; }
    Elt<T>::store8(out + p * out_stride, v);
}

hipError_t launch_preprocess(const void *img, int fmt, int n, int hw, float scale, void *out, int out_dt,
                             int out_stride, hipStream_t s, float post_mul, float post_add)
{
    size_t npix = (size_t)n * hw;
    WITH_DT(out_dt, hipLaunchKernelGGL(k_preprocess<T>, grid_for(npix), dim3(256), 0, s, img, fmt, npix, (size_t)hw, scale, (T *)out, out_stride, post_mul, post_add));
    return hipGetLastError();
}

// legacy TF bilinear (no half-pixel offset): src = dst * (in/out); value/255 first (D2T _input_process)
template <typename T>
__global__ void k_resize_u8(const uint8_t *img, int h, int w, int so, T *out, int out_stride, int out_c, float post_scale, float post_add)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= so * so) return;
    int oy = p / so, ox = p - oy * so;
    const float hs = (float)h / (float)so, ws = (float)w / (float)so;
    float fy = (float)oy * hs, fx = (float)ox * ws;
    int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
    int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
    float yl = fy - (float)y0, xl = fx - (float)x0;
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float tl = (float)img[((size_t)y0 * w + x0) * 3 + c] / 255.0f;
        float tr = (float)img[((size_t)y0 * w + x1) * 3 + c] / 255.0f;
        float bl = (float)img[((size_t)y1 * w + x0) * 3 + c] / 255.0f;
        float br = (float)img[((size_t)y1 * w + x1) * 3 + c] / 255.0f;
        float top = tl + (tr - tl) * xl;
        float bot = bl + (br - bl) * xl;
        v[c] = top + (bot - top) * yl;
        if (post_scale != 1.0f) v[c] *= post_scale;
        if (post_add != 0.0f) v[c] += post_add;
    }
    if (out_c >= 8) Elt<T>::store8(out + (size_t)p * out_stride, v);
    else
        for (int c = 0; c < out_c; ++c) Elt<T>::store1(out + (size_t)p * out_stride + c, v[c]);
}

hipError_t launch_resize_u8(const uint8_t *img, int h, int w, int s_out, void *out, int out_dt, int out_stride,
                            int out_c, hipStream_t s, float post_scale, float post_add)
{
    size_t np = (size_t)s_out * s_out;
    WITH_DT(out_dt, hipLaunchKernelGGL(k_resize_u8<T>, grid_for(np), dim3(256), 0, s, img, h, w, s_out, (T *)out, out_stride, out_c, post_scale, post_add));
    return hipGetLastError();
}

// `cv2.resize(image_float32, (ow, oh))` as V2/utils.py:13-27 calls it (INTER_LINEAR on a float32 image, after BGR -> RGB) followed by the
// reference's `/ 225.0` -- restated from OpenCV's published resize (imgproc/resize.cpp, the CV_32F linear path): half-pixel centres,
//   fx = (float)((dx + 0.5) * (double)(src_w / dst_w as 1 / (dst_w / src_w)) - 0.5); sx = floor(fx); fx -= sx;
//   sx < 0 -> (sx, fx) = (0, 0);  sx >= src_w - 1 -> (sx, fx) = (src_w - 1, 0);     likewise in y,
// a horizontal pass S[sx] * (1 - fx) + S[sx + 1] * fx on both rows, then the vertical one R0 * (1 - fy) + R1 * fy (separately rounded
// float operations: this file is compiled with -ffp-contract=off).  cv2 is absent here: parity unpinned, oracle.resize_cv2_linear is the
// same closed form.  swap_rb: output channel c reads input channel 2 - c (cv2.cvtColor(..., COLOR_BGR2RGB)).
__global__ void k_resize_cv2_u8(const uint8_t *img, int h, int w, int oh, int ow, int swap_rb, float divisor, float *out)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= oh * ow) return;
    const int dy = p / ow, dx = p - dy * ow;
    const double scale_x = 1.0 / ((double)ow / (double)w), scale_y = 1.0 / ((double)oh / (double)h);
    float fx = (float)(((double)dx + 0.5) * scale_x - 0.5), fy = (float)(((double)dy + 0.5) * scale_y - 0.5);
    int sx = (int)floorf(fx), sy = (int)floorf(fy);
    fx -= (float)sx; fy -= (float)sy;
    if (sx < 0) { sx = 0; fx = 0.f; }
    if (sx >= w - 1) { sx = w - 1; fx = 0.f; }
    if (sy < 0) { sy = 0; fy = 0.f; }
    if (sy >= h - 1) { sy = h - 1; fy = 0.f; }
    const int sx1 = min(sx + 1, w - 1), sy1 = min(sy + 1, h - 1);
    const float a0 = 1.f - fx, a1 = fx, b0 = 1.f - fy, b1 = fy;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int ci = swap_rb ? 2 - c : c;
        const float p00 = (float)img[((size_t)sy * w + sx) * 3 + ci], p01 = (float)img[((size_t)sy * w + sx1) * 3 + ci];
        const float p10 = (float)img[((size_t)sy1 * w + sx) * 3 + ci], p11 = (float)img[((size_t)sy1 * w + sx1) * 3 + ci];
        const float r0 = p00 * a0 + p01 * a1, r1 = p10 * a0 + p11 * a1;
        out[(size_t)p * 3 + c] = (r0 * b0 + r1 * b1) / divisor;
    }
}
hipError_t launch_resize_cv2_u8(const uint8_t *img, int h, int w, int oh, int ow, int swap_rb, float divisor, float *out, hipStream_t s)
{
    hipLaunchKernelGGL(k_resize_cv2_u8, grid_for((size_t)oh * ow), dim3(256), 0, s, img, h, w, oh, ow, swap_rb, divisor, out);
    return hipGetLastError();
}

// ---- row U: 2x upsample.  bilinear = closed form of pad(SYMMETRIC 1) -> legacy resize_bilinear -> crop,
//      evaluated with TF's lerp order (x then y, a + (b-a)*t, t in {0, .5}); else nearest (darknet). ----
template <typename T>
__global__ void k_upsample2x(const T *in, int is, T *out, int os, int n, int h, int w, int c8, int bilinear)
{
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)n * 2 * h * 2 * w * c8;
    if (idx >= total) return;
    int g = (int)(idx % c8); size_t p = idx / c8;
    int ox = (int)(p % (2 * w)); p /= (2 * w);
    int oy = (int)(p % (2 * h)); int b = (int)(p / (2 * h));
    int iy = oy >> 1, ix = ox >> 1;
    const T *base = in + (size_t)b * h * w * is + g * 8;
    float r[8];
    if (!bilinear) {
        Elt<T>::load8(base + ((size_t)iy * w + ix) * is, r);
    } else {
        int iy1 = min(iy + 1, h - 1), ix1 = min(ix + 1, w - 1);
        float xl = (ox & 1) ? 0.5f : 0.f, yl = (oy & 1) ? 0.5f : 0.f;
        float tl[8], tr[8], bl[8], br[8];
        Elt<T>::load8(base + ((size_t)iy * w + ix) * is, tl);
        Elt<T>::load8(base + ((size_t)iy * w + ix1) * is, tr);
        Elt<T>::load8(base + ((size_t)iy1 * w + ix) * is, bl);
        Elt<T>::load8(base + ((size_t)iy1 * w + ix1) * is, br);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float top = tl[i] + (tr[i] - tl[i]) * xl;
            float bot = bl[i] + (br[i] - bl[i]) * xl;
            r[i] = top + (bot - top) * yl;
        }
    }
    Elt<T>::store8(out + (((size_t)b * 2 * h + oy) * 2 * w + ox) * os + g * 8, r);
}

hipError_t launch_upsample2x(const TView &in, const TView &out, int bilinear, hipStream_t s)
{
    size_t total = (size_t)in.n * 4 * in.h * in.w * (in.c / 8);
    WITH_DT(in.dt, hipLaunchKernelGGL(k_upsample2x<T>, grid_for(total), dim3(256), 0, s, (const T *)in.ptr, in.stride, (T *)out.ptr, out.stride, in.n, in.h, in.w, in.c / 8, bilinear));
    return hipGetLastError();
}

// ---- row M: max-pool, window origin -pad, out-of-range = -inf (DN/maxpool_layer.c:79-111 == TF) ----
template <typename T>
__global__ void k_maxpool(const T *in, int is, T *out, int os, int n, int h, int w, int ho, int wo, int c8,
                          int size, int stride, int pad)
{
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)n * ho * wo * c8;
    if (idx >= total) return;
    int g = (int)(idx % c8); size_t p = idx / c8;
    int ox = (int)(p % wo); p /= wo;
    int oy = (int)(p % ho); int b = (int)(p / ho);
    float m[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) m[i] = -INFINITY;
    for (int dy = 0; dy < size; ++dy)
        for (int dx = 0; dx < size; ++dx) {
            int iy = oy * stride + dy - pad, ix = ox * stride + dx - pad;
            if ((unsigned)iy < (unsigned)h && (unsigned)ix < (unsigned)w) {
                float v[8];
                Elt<T>::load8(in + (((size_t)b * h + iy) * w + ix) * is + g * 8, v);
#pragma unroll
                for (int i = 0; i < 8; ++i) m[i] = v[i] > m[i] ? v[i] : m[i];
            }
        }
    Elt<T>::store8(out + (((size_t)b * ho + oy) * wo + ox) * os + g * 8, m);
}

hipError_t launch_maxpool(const TView &in, const TView &out, int size, int stride, int pad, hipStream_t s)
{
    size_t total = (size_t)out.n * out.h * out.w * (in.c / 8);
    WITH_DT(in.dt, hipLaunchKernelGGL(k_maxpool<T>, grid_for(total), dim3(256), 0, s, (const T *)in.ptr, in.stride, (T *)out.ptr, out.stride, in.n, in.h, in.w, out.h, out.w, in.c / 8, size, stride, pad));
    return hipGetLastError();
}

// ---- row R: tf.space_to_depth (granule copy) or darknet reorg_cpu(forward=0) (element scramble) ----
template <typename T>
__global__ void k_space_to_depth(const T *in, int is, T *out, int os, int n, int h, int w, int c8, int st)
{
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)n * h * w * c8;
    if (idx >= total) return;
    int g = (int)(idx % c8); size_t p = idx / c8;
    int ix = (int)(p % w); p /= w;
    int iy = (int)(p % h); int b = (int)(p / h);
    float v[8];
    Elt<T>::load8(in + (((size_t)b * h + iy) * w + ix) * is + g * 8, v);
    int oy = iy / st, dy = iy - oy * st, ox = ix / st, dx = ix - ox * st;
    int oc = (dy * st + dx) * (c8 * 8) + g * 8;
    Elt<T>::store8(out + (((size_t)b * (h / st) + oy) * (w / st) + ox) * os + oc, v);
}

// Darknet: with x,out as NCHW flat buffers of the *input* geometry (w,h,c): out[in_index] = x[out_index]
// (DN/blas.c:9-30, forward=0), the result then reinterpreted as [c*s*s, h/s, w/s].
template <typename T>
__global__ void k_reorg_darknet(const T *in, int is, T *out, int os, int n, int h, int w, int c, int st)
{
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t per = (size_t)h * w * c;
    if (idx >= per * n) return;
    int b = (int)(idx / per); int t = (int)(idx - (size_t)b * per);      // t = in_index
    int i = t % w, j = (t / w) % h, k = t / (w * h);
    int out_c = c / (st * st);
    int c2 = k % out_c, off = k / out_c;
    int w2 = i * st + off % st, h2 = j * st + off / st;
    int src = w2 + w * st * (h2 + h * st * c2);                          // flat NCHW index into x viewed as [out_c, h*st, w*st]
    // x is the input tensor [c, h, w] NCHW flat: decode src as (cs, ys, xs) of that geometry
    int xs = src % w, ys = (src / w) % h, cs = src / (w * h);
    float v = Elt<T>::load1(in + (((size_t)b * h + ys) * w + xs) * is + cs);
    // destination flat index t of a buffer reinterpreted as [c*st*st, h/st, w/st]
    int wo = w / st, ho = h / st;
    int xo = t % wo, yo = (t / wo) % ho, co = t / (wo * ho);
    Elt<T>::store1(out + (((size_t)b * ho + yo) * wo + xo) * os + co, v);
}

hipError_t launch_reorg(const TView &in, const TView &out, int stride, int darknet, hipStream_t s)
{
    if (!darknet) {
        size_t total = (size_t)in.n * in.h * in.w * (in.c / 8);
        WITH_DT(in.dt, hipLaunchKernelGGL(k_space_to_depth<T>, grid_for(total), dim3(256), 0, s, (const T *)in.ptr, in.stride, (T *)out.ptr, out.stride, in.n, in.h, in.w, in.c / 8, stride));
    } else {
        size_t total = (size_t)in.n * in.h * in.w * in.c;
        WITH_DT(in.dt, hipLaunchKernelGGL(k_reorg_darknet<T>, grid_for(total), dim3(256), 0, s, (const T *)in.ptr, in.stride, (T *)out.ptr, out.stride, in.n, in.h, in.w, in.c, stride));
    }
    return hipGetLastError();
}

// ---- rows B / Rt fallbacks: residual add and strided copy --------------------------------------
template <typename T>
__global__ void k_add(const T *a, int as, const T *b, int bs, T *o, int os, size_t npix, int c8, float sa, float sb, float so)
{
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= npix * c8) return;
    size_t p = idx / c8; int g = (int)(idx - p * c8);
    float x[8], y[8];
    Elt<T>::load8(a + p * as + g * 8, x);
    Elt<T>::load8(b + p * bs + g * 8, y);
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = (x[i] * sa + y[i] * sb) * so;       // scales are exactly 1 unless fp8
    Elt<T>::store8(o + p * os + g * 8, x);
}
hipError_t launch_add(const TView &a, const TView &b, const TView &out, hipStream_t s, float sa, float sb, float so)
{
    size_t npix = (size_t)a.n * a.h * a.w; int c8 = a.c / 8;
    WITH_DT(a.dt, hipLaunchKernelGGL(k_add<T>, grid_for(npix * c8), dim3(256), 0, s, (const T *)a.ptr, a.stride, (const T *)b.ptr, b.stride, (T *)out.ptr, out.stride, npix, c8, sa, sb, so));
    return hipGetLastError();
}

template <typename T>
__global__ void k_copy(const T *a, int as, T *o, int os, size_t npix, int c8)
{
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= npix * c8) return;
    size_t p = idx / c8; int g = (int)(idx - p * c8);
    float x[8];
    Elt<T>::load8(a + p * as + g * 8, x);
    Elt<T>::store8(o + p * os + g * 8, x);
}
hipError_t launch_copy(const TView &in, const TView &out, hipStream_t s)
{
    size_t npix = (size_t)in.n * in.h * in.w; int c8 = in.c / 8;
    WITH_DT(in.dt, hipLaunchKernelGGL(k_copy<T>, grid_for(npix * c8), dim3(256), 0, s, (const T *)in.ptr, in.stride, (T *)out.ptr, out.stride, npix, c8));
    return hipGetLastError();
}

// ---- dtype conversion between dense fp32 NHWC host-staging buffers and device views -------------
template <typename T>
__global__ void k_to_f32(const T *in, int is, float *out, size_t npix, int c, float scale)
{
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= npix * c) return;
    size_t p = idx / c; int ch = (int)(idx - p * c);
    out[idx] = Elt<T>::load1(in + p * is + ch) * scale;
}
hipError_t launch_to_f32(const TView &in, float *out, hipStream_t s, float scale)
{
    size_t npix = (size_t)in.n * in.h * in.w;
    WITH_DT(in.dt, hipLaunchKernelGGL(k_to_f32<T>, grid_for(npix * in.c), dim3(256), 0, s, (const T *)in.ptr, in.stride, out, npix, in.c, scale));
    return hipGetLastError();
}

template <typename T>
__global__ void k_from_f32(const float *in, T *out, int os, size_t npix, int c, float scale)
{
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= npix * c) return;
    size_t p = idx / c; int ch = (int)(idx - p * c);
    Elt<T>::store1(out + p * os + ch, in[idx] * scale);
}
hipError_t launch_from_f32(const float *in, const TView &out, hipStream_t s, float scale)
{
    size_t npix = (size_t)out.n * out.h * out.w;
    WITH_DT(out.dt, hipLaunchKernelGGL(k_from_f32<T>, grid_for(npix * out.c), dim3(256), 0, s, in, (T *)out.ptr, out.stride, npix, out.c, scale));
    return hipGetLastError();
}

// ---- split fp16 storage (YOLO_FP16X2): a value v is the pair hi = f16(v), lo = f16(v - hi) -- 22 significant bits out of two 11-bit
//      halves.  A layer output is stored INTERLEAVED (round 6): per 32-channel group 32 hi then 32 lo, [pixel][2 * Cp], Cp = channels rounded
//      up to 32 -- a 128-byte run of a pixel is one K-step row [hi | lo] of the conv that reads it, which forms W_hi x_hi + W_lo x_hi +
//      W_hi x_lo from it (conv_igemm_kernel.h, PAIRK; what is dropped is W_lo x_lo, 2^-22 of the product).  The network input keeps rounds
//      4-5's three blocks hi | lo | hi of its 8 padded channels (PAIR_B3), read by an ordinary fp16 K loop against W_hi | W_hi | W_lo.
//      These are the memory-bound pieces around the convs. ----
__device__ __forceinline__ void split8(const float *v, uint4 &H, uint4 &L)
{
    typedef Elt<f16_t> E;
    H = uint4{E::pk(v[0], v[1]), E::pk(v[2], v[3]), E::pk(v[4], v[5]), E::pk(v[6], v[7])};
    const uint32_t w[4] = {H.x, H.y, H.z, H.w};
    float lo[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const E::h2 h = __builtin_bit_cast(E::h2, w[i]); lo[2 * i] = v[2 * i] - (float)h[0]; lo[2 * i + 1] = v[2 * i + 1] - (float)h[1]; }
    L = uint4{E::pk(lo[0], lo[1]), E::pk(lo[2], lo[3]), E::pk(lo[4], lo[5]), E::pk(lo[6], lo[7])};
}
// element offset of the hi piece of channels [g * 8, g * 8 + 8) inside a pixel, and the distance to its lo piece
__device__ __forceinline__ int pair_hi_off(int g, int Cp, int layout) { const int c = g * 8; return layout == PAIR_B3 ? c : (c >> 5) * 64 + (c & 31); }
__device__ __forceinline__ int pair_lo_dist(int Cp, int layout) { return layout == PAIR_B3 ? Cp : 32; }
__device__ __forceinline__ void join8(const f16_t *p, int lo_dist, float *v)
{
    float h[8], l[8];
    Elt<f16_t>::load8(p, h); Elt<f16_t>::load8(p + lo_dist, l);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = h[i] + l[i];
}
__device__ __forceinline__ void put_split8(f16_t *p, int Cp, int layout, const float *v)
{
    uint4 H, L; split8(v, H, L);
    *(uint4 *)p = H; *(uint4 *)(p + pair_lo_dist(Cp, layout)) = L;
    if (layout == PAIR_B3) *(uint4 *)(p + 2 * Cp) = H;
}
__global__ void k_split_from_f32(const float *in, int is, f16_t *out, int os, int Cp, size_t npix, int layout)
{
    const int c8 = Cp / 8;
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= npix * c8) return;
    size_t p = idx / c8; int g = (int)(idx - p * c8);
    float v[8]; Elt<float>::load8(in + p * is + g * 8, v);
    put_split8(out + p * os + pair_hi_off(g, Cp, layout), Cp, layout, v);
}
__global__ void k_split_to_f32(const f16_t *in, int is, int Cp, float *out, int os, size_t npix, int layout)
{
    const int c8 = Cp / 8;
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= npix * c8) return;
    size_t p = idx / c8; int g = (int)(idx - p * c8);
    float v[8]; join8(in + p * is + pair_hi_off(g, Cp, layout), pair_lo_dist(Cp, layout), v);
    Elt<float>::store8(out + p * os + g * 8, v);
}
// shortcut on split tensors: (a_hi + a_lo) + (b_hi + b_lo), split again (each parenthesis is exact in fp32 when the pair came from split8)
__global__ void k_add_split(const f16_t *a, int as, const f16_t *b, int bs, f16_t *o, int os, int Cp, size_t npix)
{
    const int c8 = Cp / 8;
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= npix * c8) return;
    size_t p = idx / c8; int g = (int)(idx - p * c8);
    const int ho = pair_hi_off(g, Cp, PAIR_ILV);
    float x[8], y[8];
    join8(a + p * as + ho, 32, x); join8(b + p * bs + ho, 32, y);
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = x[i] + y[i];
    put_split8(o + p * os + ho, Cp, PAIR_ILV, x);
}
// 2x upsample of an interleaved pair tensor in ONE launch: join (hi + lo), the fp32 kernel's arithmetic (k_upsample2x: TF's lerp order, or
// nearest), split -- bit-identical to the join / fp32 upsample / split sequence it replaces (three launches through fp32 staging)
__global__ void k_upsample2x_pair(const f16_t *in, int is, f16_t *out, int os, int n, int h, int w, int Cp, int bilinear)
{
    const int c8 = Cp / 8;
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)n * 2 * h * 2 * w * c8;
    if (idx >= total) return;
    const int g = (int)(idx % c8); size_t p = idx / c8;
    const int ox = (int)(p % (2 * w)); p /= (2 * w);
    const int oy = (int)(p % (2 * h)); const int b = (int)(p / (2 * h));
    const int iy = oy >> 1, ix = ox >> 1, ho = pair_hi_off(g, Cp, PAIR_ILV);
    const f16_t *base = in + (size_t)b * h * w * is + ho;
    f16_t *o = out + (((size_t)b * 2 * h + oy) * 2 * w + ox) * os + ho;
    if (!bilinear) {          // nearest: join and split again, as the fp32 sequence does (split(join(hi, lo)) may choose another pair for the same value at a tie)
        float v[8]; join8(base + ((size_t)iy * w + ix) * is, 32, v);
        put_split8(o, Cp, PAIR_ILV, v);
        return;
    }
    const int iy1 = min(iy + 1, h - 1), ix1 = min(ix + 1, w - 1);
    const float xl = (ox & 1) ? 0.5f : 0.f, yl = (oy & 1) ? 0.5f : 0.f;
    float tl[8], tr[8], bl[8], br[8], r[8];
    join8(base + ((size_t)iy * w + ix) * is, 32, tl); join8(base + ((size_t)iy * w + ix1) * is, 32, tr);
    join8(base + ((size_t)iy1 * w + ix) * is, 32, bl); join8(base + ((size_t)iy1 * w + ix1) * is, 32, br);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float top = tl[i] + (tr[i] - tl[i]) * xl;
        const float bot = bl[i] + (br[i] - bl[i]) * xl;
        r[i] = top + (bot - top) * yl;
    }
    put_split8(o, Cp, PAIR_ILV, r);
}
hipError_t launch_upsample2x_pair(const TView &in, const TView &out, int bilinear, hipStream_t s)
{
    const int Cp = (in.c + 31) / 32 * 32;
    if (in.dt != DT_F16 || out.dt != DT_F16 || in.stride < 2 * Cp || out.stride < 2 * Cp) return hipErrorInvalidValue;
    const size_t total = (size_t)in.n * 4 * in.h * in.w * (Cp / 8);
    hipLaunchKernelGGL(k_upsample2x_pair, grid_for(total), dim3(256), 0, s, (const f16_t *)in.ptr, in.stride, (f16_t *)out.ptr, out.stride, in.n, in.h, in.w, Cp, bilinear);
    return hipGetLastError();
}
hipError_t launch_split_from_f32(const float *in, int in_stride, void *out, int out_stride, int Cp, size_t npix, hipStream_t s, int layout)
{
    hipLaunchKernelGGL(k_split_from_f32, grid_for(npix * (Cp / 8)), dim3(256), 0, s, in, in_stride, (f16_t *)out, out_stride, Cp, npix, layout);
    return hipGetLastError();
}
hipError_t launch_split_to_f32(const void *in, int in_stride, int Cp, float *out, int out_stride, size_t npix, hipStream_t s, int layout)
{
    hipLaunchKernelGGL(k_split_to_f32, grid_for(npix * (Cp / 8)), dim3(256), 0, s, (const f16_t *)in, in_stride, Cp, out, out_stride, npix, layout);
    return hipGetLastError();
}
hipError_t launch_add_split(const void *a, int a_stride, const void *b, int b_stride, void *out, int out_stride, int Cp, size_t npix, hipStream_t s)
{
    hipLaunchKernelGGL(k_add_split, grid_for(npix * (Cp / 8)), dim3(256), 0, s, (const f16_t *)a, a_stride, (const f16_t *)b, b_stride, (f16_t *)out, out_stride, Cp, npix);
    return hipGetLastError();
}

// ---- darknet letterbox_image (DN/image.c:960-981) fused with the layout change: a planar float image of any size ->
//      aspect-preserving resize_image (DN/image.c:1347-1393: horizontal pass then vertical pass, scales (in-1)/(out-1),
//      last column / row copied) embedded at the centre of a 0.5-filled S x S canvas, written as the 8-channel
//      network input.  Operation order of the two passes is kept: part = (1-dx)*a + dx*b, then (1-dy)*p0 (+ dy*p1). ----
template <typename T>
__global__ void k_letterbox_chw(const float *img, int iw, int ih, int S, int new_w, int new_h, int off_x, int off_y, T *out, int out_stride)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= S * S) return;
    const int oy = p / S, ox = p - oy * S;
    const int r = oy - off_y, c = ox - off_x;
    float v[8] = {0.5f, 0.5f, 0.5f, 0, 0, 0, 0, 0};
    if ((unsigned)r < (unsigned)new_h && (unsigned)c < (unsigned)new_w) {
        const float w_scale = (float)(iw - 1) / (float)(new_w - 1), h_scale = (float)(ih - 1) / (float)(new_h - 1);
        const float sy = (float)r * h_scale; const int iy = (int)sy; const float dy = sy - (float)iy;
        const bool last_c = c == new_w - 1 || iw == 1, last_r = r == new_h - 1 || ih == 1;
        const float sx = (float)c * w_scale; const int ix = (int)sx; const float dx = sx - (float)ix;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float *pl = img + (size_t)k * iw * ih;
            auto part = [&](int row) {
                if (last_c) return pl[(size_t)row * iw + (iw - 1)];
                return (1 - dx) * pl[(size_t)row * iw + ix] + dx * pl[(size_t)row * iw + ix + 1];
            };
            float val = (1 - dy) * part(iy);
            if (!last_r) val = val + dy * part(iy + 1);
            v[k] = val;
        }
    }
    Elt<T>::store8(out + (size_t)p * out_stride, v);
}
hipError_t launch_letterbox_chw(const float *img, int iw, int ih, int S, void *out, int out_dt, int out_stride, hipStream_t s)
{
    int new_w, new_h;
    if (((float)S / iw) < ((float)S / ih)) { new_w = S; new_h = (ih * S) / iw; } else { new_h = S; new_w = (iw * S) / ih; }
    if (new_w < 1 || new_h < 1) return hipErrorInvalidValue;
    WITH_DT(out_dt, hipLaunchKernelGGL(k_letterbox_chw<T>, grid_for((size_t)S * S), dim3(256), 0, s, img, iw, ih, S, new_w, new_h, (S - new_w) / 2, (S - new_h) / 2, (T *)out, out_stride));
    return hipGetLastError();
}

// ---- darknet letterbox_image (DN/image.c:960-981: resize_image to the aspect-preserving size, embedded in a 0.5-grey w x h
//      canvas) and resize_image itself (DN/image.c:1347-1389; embed == 0: the image is stretched to w x h), planar float in,
//      planar float out.  Same two-pass operation order as k_letterbox_chw above. ----
__global__ void k_letterbox_planar(const float *img, int iw, int ih, int W, int H, int new_w, int new_h, int off_x, int off_y, float *out)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= W * H) return;
    const int oy = p / W, ox = p - oy * W;
    const int r = oy - off_y, c = ox - off_x;
    float v[3] = {0.5f, 0.5f, 0.5f};
    if ((unsigned)r < (unsigned)new_h && (unsigned)c < (unsigned)new_w) {
        const float w_scale = (float)(iw - 1) / (float)(new_w - 1), h_scale = (float)(ih - 1) / (float)(new_h - 1);
        const float sy = (float)r * h_scale; const int iy = (int)sy; const float dy = sy - (float)iy;
        const bool last_c = c == new_w - 1 || iw == 1, last_r = r == new_h - 1 || ih == 1;
        const float sx = (float)c * w_scale; const int ix = (int)sx; const float dx = sx - (float)ix;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float *pl = img + (size_t)k * iw * ih;
            auto part = [&](int row) {
                if (last_c) return pl[(size_t)row * iw + (iw - 1)];
                return (1 - dx) * pl[(size_t)row * iw + ix] + dx * pl[(size_t)row * iw + ix + 1];
            };
            float val = (1 - dy) * part(iy);
            if (!last_r) val = val + dy * part(iy + 1);
            v[k] = val;
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) out[(size_t)k * W * H + p] = v[k];
}
hipError_t launch_letterbox_planar(const float *img, int iw, int ih, int w, int h, int embed, float *out, hipStream_t s)
{
    int new_w = w, new_h = h;
    if (embed) { if (((float)w / iw) < ((float)h / ih)) { new_w = w; new_h = (ih * w) / iw; } else { new_h = h; new_w = (iw * h) / ih; } }
    if (new_w < 1 || new_h < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_letterbox_planar, grid_for((size_t)w * h), dim3(256), 0, s, img, iw, ih, w, h, new_w, new_h, (w - new_w) / 2, (h - new_h) / 2, out);
    return hipGetLastError();
}

// ---- [local]: locally connected layer (DN/local_layer.c:91-120): a conv whose filters are NOT shared between output locations.
//      out[n, oy, ox, f] = act(bias[loc][f] + sum_{kh,kw,c} W[loc][f][kh][kw][c] * in[n, oy s - p + kh, ox s - p + kw, c]), loc = oy Wo + ox.
//      Every filter value is used once per image: the layer streams its weights (darknet's yolov1.cfg: 49 locations x 256 x 9216) and
//      is HBM-bound by construction.  One wave per (location, filter): lanes run along the 8-channel granules of a tap (16-byte loads
//      of filter and activation), fp32 accumulation, shuffle reduction; the image loop is inside so the filter row is read once. ----
template <typename T>
__global__ __launch_bounds__(256) void k_local(const T *in, int in_stride, const T *w, const float *bias, T *out, int out_stride,
                                               int n, int H, int W, int C, int Ho, int Wo, int F, int k, int stride, int pad, int act)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int loc = blockIdx.x, f = blockIdx.y * 4 + wv;
    if (f >= F) {                // channels F .. roundup(F, 8) of the stored pixel: zero, so that a consumer's zero filters never meet stale Inf / NaN
        if (f < ((F + 7) & ~7) && lane == 0) for (int b = 0; b < n; ++b) Elt<T>::store1(out + ((size_t)b * Ho * Wo + loc) * out_stride + f, 0.f);
        return;
    }
    const int oy = loc / Wo, ox = loc - oy * Wo;
    const int c8n = C / 8, K8 = k * k * c8n;                      // 8-channel granules per tap / per filter row
    const T *wr = w + ((size_t)loc * F + f) * (size_t)K8 * 8;
    for (int b = 0; b < n; ++b) {
        float acc = 0.f;
        for (int g = lane; g < K8; g += 64) {
            const int t = g / c8n, c8 = g - t * c8n;
            const int kh = t / k, kw = t - kh * k;
            const int iy = oy * stride - pad + kh, ix = ox * stride - pad + kw;
            if ((unsigned)iy >= (unsigned)H || (unsigned)ix >= (unsigned)W) continue;
            float xv[8], wv8[8];
            Elt<T>::load8(in + ((size_t)(b * H + iy) * W + ix) * in_stride + c8 * 8, xv);
            Elt<T>::load8(wr + (size_t)g * 8, wv8);
#pragma unroll
            for (int q = 0; q < 8; ++q) acc = fmaf(xv[q], wv8[q], acc);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
        if (lane == 0) {
            float v = acc + bias[(size_t)loc * F + f];
            if (act == ACT_LEAKY) v = v > 0.f ? v : 0.1f * v;
            Elt<T>::store1(out + ((size_t)b * Ho * Wo + loc) * out_stride + f, v);
        }
    }
}

hipError_t launch_local(const TView &in, const TView &out, const void *w, const float *bias, int k, int stride, int pad, int act, hipStream_t s)
{
    if (in.c % 8 || in.dt != out.dt || in.dt == DT_FP8) return hipErrorInvalidValue;
    dim3 grid((unsigned)(out.h * out.w), (unsigned)((((out.c + 7) & ~7) + 3) / 4));
    WITH_DT(in.dt, hipLaunchKernelGGL(k_local<T>, grid, dim3(256), 0, s, (const T *)in.ptr, in.stride, (const T *)w, bias, (T *)out.ptr, out.stride,
                                      in.n, in.h, in.w, in.c, out.h, out.w, out.c, k, stride, pad, act));
    return hipGetLastError();
}
