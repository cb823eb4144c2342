// Lookup-free quantisation (reference: genie/module/quantization.py:77-133).
//
//   quantize : quant = sign(z) (sign(0) = 0), idx = sum_i (z_i > 0) << (d-1-i)   -- MSB first, int64
//   loss     : commit MSE + per-token entropy + entropy of the batch-mean distribution over all 2^d
//              codes, with the reference's clamp *inside* the log (eps = 1e-6), forward and backward.
//
// The 2^d-way softmax factorises: p[code] = prod_i s(+-x_i), so with the bits split into a high and a
// low half p = A (x) B with |A| = 2^dh, |B| = 2^dl.  The clamp forbids a closed form, so the per-code
// terms are enumerated -- from A (x) B held in LDS, never from an N x 2^d matrix in HBM (the reference
// materialises 1 GiB at N=1024, d=18).
#include "common.h"
#include "genie_hip.h"

template <typename T>
__device__ __forceinline__ float ld(const T* p);
template <>
__device__ __forceinline__ float ld<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float ld<bf16_t>(const bf16_t* p) { return bf16_to_f32(*p); }
template <typename T>
__device__ __forceinline__ void st(T* p, float v);
template <>
__device__ __forceinline__ void st<float>(float* p, float v) { *p = v; }
template <>
__device__ __forceinline__ void st<bf16_t>(bf16_t* p, float v) { *p = f32_to_bf16(v); }

// one lane = one (token, codebook)
template <typename T>
__global__ void __launch_bounds__(256) lfq_quantize_kernel(const T* __restrict__ z, T* __restrict__ quant,
                                                           long long* __restrict__ idx, long long ntok, int ncb, int d,
                                                           long long pitch) {
    const long long total = ntok * ncb;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long tok = i / ncb;
        const int cb = (int)(i % ncb);
        const T* zp = z + tok * pitch + (long long)cb * d;
        T* qp = quant ? quant + tok * pitch + (long long)cb * d : nullptr;
        long long code = 0;
        for (int j = 0; j < d; ++j) {
            const float v = ld<T>(zp + j);
            code = (code << 1) | (v > 0.f ? 1 : 0);                    // NaN and 0 -> bit 0
            if (qp) st<T>(qp + j, v > 0.f ? 1.f : (v < 0.f ? -1.f : (v == 0.f ? 0.f : v)));   // sign(); NaN passes through
        }
        idx[i] = code;
    }
}

extern "C" int genie_lfq_quantize(const void* z, int dtype, int64_t ntok, int num_codebook, int codebook_dim, int64_t pitch,
                                  void* quant, int64_t* idx, void* stream) {
    GENIE_CHECK_ARG(z && idx, "genie_lfq_quantize: null pointer");
    GENIE_CHECK_ARG(codebook_dim >= 1 && codebook_dim <= 62, "genie_lfq_quantize: codebook_dim %d out of range [1, 62]", codebook_dim);
    GENIE_CHECK_ARG(num_codebook >= 1 && pitch >= (int64_t)num_codebook * codebook_dim, "genie_lfq_quantize: pitch %lld < %d x %d", (long long)pitch, num_codebook, codebook_dim);
    if (ntok == 0) return GENIE_OK;
    const long long total = ntok * num_codebook;
    int grid = (int)((total + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GENIE_BF16)
        lfq_quantize_kernel<bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)z, (bf16_t*)quant, (long long*)idx, ntok, num_codebook, codebook_dim, pitch);
    else if (dtype == GENIE_F32)
        lfq_quantize_kernel<float><<<grid, 256, 0, s>>>((const float*)z, (float*)quant, (long long*)idx, ntok, num_codebook, codebook_dim, pitch);
    else
        GENIE_CHECK_ARG(false, "genie_lfq_quantize: unsupported dtype %d", dtype);
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}
