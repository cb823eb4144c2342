// HBM-bound elementwise / reduction kernels: layout conversion at the model boundary, SiLU,
// residual add, MSE loss, fused AdamW, fp32 -> bf16 weight packing.
//
// Internal activation layout ("CL"): bf16, channels-last (N, T, H, W, Cp) with the channel pitch
// Cp a multiple of 8 (16 B), pad channels kept at zero.  Every kernel here moves 16 B per lane.
#include "common.h"
#include "genie_hip.h"

// ------------------------------------------------------------------------------------------------
// strided (fp32 | bf16) NCTHW  ->  CL bf16        one lane = one pixel x 8 channels
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ float load_as_f32(const T* p);
template <>
__device__ __forceinline__ float load_as_f32<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float load_as_f32<bf16_t>(const bf16_t* p) { return bf16_to_f32(*p); }

template <typename T>
__global__ void __launch_bounds__(256) to_cl_kernel(const T* __restrict__ src, bf16_t* __restrict__ dst,
                                                    int N, int C, int T_, int H, int W, long long sN, long long sC,
                                                    long long sT, long long sH, long long sW, int Cp) {
    const long long npix = (long long)N * T_ * H * W;
    const int nchunk = Cp >> 3;
    const long long total = npix * nchunk;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        // pixel fastest within a chunk plane: lanes read consecutive w for one channel (coalesced for NCTHW)
        const long long pix = i % npix;
        const int chunk = (int)(i / npix);
        long long r = pix;
        const int w = (int)(r % W); r /= W;
        const int h = (int)(r % H); r /= H;
        const int t = (int)(r % T_); r /= T_;
        const int n = (int)r;
        const T* base = src + n * sN + t * sT + h * sH + w * sW;
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = chunk * 8 + j;
            f[j] = c < C ? load_as_f32<T>(base + c * sC) : 0.f;
        }
        *reinterpret_cast<u32x4_t*>(dst + pix * Cp + chunk * 8) = pack8(f);
    }
}

template <typename T>
__device__ __forceinline__ void store_from_f32(T* p, float v);
template <>
__device__ __forceinline__ void store_from_f32<float>(float* p, float v) { *p = v; }
template <>
__device__ __forceinline__ void store_from_f32<bf16_t>(bf16_t* p, float v) { *p = f32_to_bf16(v); }

template <typename T>
__global__ void __launch_bounds__(256) from_cl_kernel(const bf16_t* __restrict__ src, T* __restrict__ dst, int N, int C,
                                                      int T_, int H, int W, long long sN, long long sC, long long sT,
                                                      long long sH, long long sW, int Cp) {
    const long long npix = (long long)N * T_ * H * W;
    const int nchunk = (C + 7) >> 3;
    const long long total = npix * nchunk;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long pix = i % npix;
        const int chunk = (int)(i / npix);
        long long r = pix;
        const int w = (int)(r % W); r /= W;
        const int h = (int)(r % H); r /= H;
        const int t = (int)(r % T_); r /= T_;
        const int n = (int)r;
        float f[8];
        unpack8(*reinterpret_cast<const u32x4_t*>(src + pix * Cp + chunk * 8), f);
        T* base = dst + n * sN + t * sT + h * sH + w * sW;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = chunk * 8 + j;
            if (c < C) store_from_f32<T>(base + c * sC, f[j]);
        }
    }
}

static int ew_grid(long long total) {
    long long b = (total + 255) / 256;
    if (b > 2048) b = 2048;   // 256 CUs x 8 blocks: grid-stride the rest
    if (b < 1) b = 1;
    return (int)b;
}

extern "C" int genie_to_channels_last(const void* src, int src_dtype, const int64_t* dims, const int64_t* strides, void* dst,
                                      int cpitch, void* stream) {
    GENIE_CHECK_ARG(src && dst && dims && strides, "genie_to_channels_last: null pointer");
    GENIE_CHECK_ARG(cpitch % 8 == 0 && cpitch >= dims[1], "genie_to_channels_last: channel pitch %d must be a multiple of 8 and >= C=%lld", cpitch, (long long)dims[1]);
    const int N = (int)dims[0], C = (int)dims[1], T = (int)dims[2], H = (int)dims[3], W = (int)dims[4];
    const long long total = (long long)N * T * H * W * (cpitch / 8);
    if (total == 0) return GENIE_OK;
    hipStream_t s = (hipStream_t)stream;
    if (src_dtype == GENIE_F32)
        to_cl_kernel<float><<<ew_grid(total), 256, 0, s>>>((const float*)src, (bf16_t*)dst, N, C, T, H, W, strides[0], strides[1], strides[2], strides[3], strides[4], cpitch);
    else if (src_dtype == GENIE_BF16)
        to_cl_kernel<bf16_t><<<ew_grid(total), 256, 0, s>>>((const bf16_t*)src, (bf16_t*)dst, N, C, T, H, W, strides[0], strides[1], strides[2], strides[3], strides[4], cpitch);
    else
        GENIE_CHECK_ARG(false, "genie_to_channels_last: unsupported dtype %d", src_dtype);
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}

// Decoded video frames -> the internal activation layout, on the DEVICE (data path in front of the hot path): the reference
// divides the uint8 frames by 255 on the host and rearranges 't h w c -> c t h w' (genie/module/data.py:218-231); at the ~145 clips/s one
// MI355X consumes that is 113 MB/s of fp32 through the loader's pipes, the pinned staging and PCIe, plus a layout pass on arrival.  The
// frames stay uint8 until they are in HBM (a quarter of the bytes) and become the model's own bf16 channels-last tensor in ONE pass:
// dst[n][t][h][w][c] = bf16(float(src[n][t][h][w][c]) / 255) for c < C, zeros up to the pitch -- the very numbers the reference's fp32
// division followed by this library's to_channels_last produces (same fp32 quotient, same round-to-nearest-even).  One pixel per thread.
__global__ void __launch_bounds__(256) u8_frames_to_cl_kernel(const uint8_t* __restrict__ src, bf16_t* __restrict__ dst, long long npix, int C, int cpitch) {
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long long)gridDim.x * 256) {
        const uint8_t* s = src + p * C;
        for (int c0 = 0; c0 < cpitch; c0 += 8) {
            float f[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = (c0 + j < C) ? (float)s[c0 + j] / 255.f : 0.f;
            *reinterpret_cast<u32x4_t*>(dst + p * cpitch + c0) = pack8(f);
        }
    }
}

extern "C" int genie_u8_frames_to_cl(const void* src_u8, int64_t npix, int C, void* dst_cl, int cpitch, void* stream) {
    GENIE_CHECK_ARG(src_u8 && dst_cl, "genie_u8_frames_to_cl: null pointer");
    GENIE_CHECK_ARG(C >= 1 && cpitch % 8 == 0 && cpitch >= C, "genie_u8_frames_to_cl: channel pitch %d must be a multiple of 8 and >= C=%d", cpitch, C);
    if (npix == 0) return GENIE_OK;
    u8_frames_to_cl_kernel<<<ew_grid(npix), 256, 0, (hipStream_t)stream>>>((const uint8_t*)src_u8, (bf16_t*)dst_cl, npix, C, cpitch);
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}

// Inverse of the depth-to-space-time rearrange on a CL tensor (reference video.py:403-408 'b (c p q r) t h w -> b c (t p)(h q)(w r)'):
// src CL [N][T P][H Q][W R][sp] (cf channels) -> dst CL [N][T][H][W][P Q R cf], channel ((p Q + q) R + r) cf + c -- the sub-pixel-major
// order of the transposed weight pack, so that the backward-data pass of an upsample conv becomes a PLAIN conv over dst (kw-triple
// kernels) instead of a gather through the shuffle.  One 16-byte chunk per thread; reads and writes are whole cf * 2-byte runs.
__global__ void __launch_bounds__(256) unshuffle_cl_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int N, int T, int H, int W,
                                                            int cf, int sp, int P, int Q, int R, long long total) {
    const int cc = cf / 8, sub = P * Q * R;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c8 = (int)(i % cc);
        long long j = i / cc;
        const int s = (int)(j % sub); j /= sub;
        const int w = (int)(j % W); j /= W;
        const int h = (int)(j % H); j /= H;
        const int t = (int)(j % T);
        const int n = (int)(j / T);
        const int r = s % R, q = (s / R) % Q, p = s / (R * Q);
        const long long so = ((((long long)n * T * P + (t * P + p)) * (H * Q) + (h * Q + q)) * (long long)(W * R) + (w * R + r)) * sp + c8 * 8;
        *reinterpret_cast<u32x4_t*>(dst + i * 8) = *reinterpret_cast<const u32x4_t*>(src + so);
    }
}

extern "C" int genie_unshuffle_cl(const void* src_cl, int src_pitch, void* dst_cl, int N, int T, int H, int W, int cf, int P, int Q, int R,
                                  void* stream) {
    GENIE_CHECK_ARG(src_cl && dst_cl, "genie_unshuffle_cl: null pointer");
    GENIE_CHECK_ARG(cf >= 8 && cf % 8 == 0 && src_pitch >= cf && src_pitch % 8 == 0, "genie_unshuffle_cl: channels %d (pitch %d) must be a multiple of 8", cf, src_pitch);
    GENIE_CHECK_ARG(N >= 1 && T >= 1 && H >= 1 && W >= 1 && P >= 1 && Q >= 1 && R >= 1, "genie_unshuffle_cl: bad geometry");
    const long long total = (long long)N * T * H * W * P * Q * R * (cf / 8);
    unshuffle_cl_kernel<<<ew_grid(total), 256, 0, (hipStream_t)stream>>>((const bf16_t*)src_cl, (bf16_t*)dst_cl, N, T, H, W, cf, src_pitch, P, Q, R, total);
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}

extern "C" int genie_from_channels_last(const void* src, int cpitch, const int64_t* dims, void* dst, int dst_dtype,
                                        const int64_t* strides, void* stream) {
    GENIE_CHECK_ARG(src && dst && dims && strides, "genie_from_channels_last: null pointer");
    GENIE_CHECK_ARG(cpitch % 8 == 0 && cpitch >= dims[1], "genie_from_channels_last: bad channel pitch %d", cpitch);
    const int N = (int)dims[0], C = (int)dims[1], T = (int)dims[2], H = (int)dims[3], W = (int)dims[4];
    const long long total = (long long)N * T * H * W * ((C + 7) / 8);
    if (total == 0) return GENIE_OK;
    hipStream_t s = (hipStream_t)stream;
    if (dst_dtype == GENIE_F32)
        from_cl_kernel<float><<<ew_grid(total), 256, 0, s>>>((const bf16_t*)src, (float*)dst, N, C, T, H, W, strides[0], strides[1], strides[2], strides[3], strides[4], cpitch);
    else if (dst_dtype == GENIE_BF16)
        from_cl_kernel<bf16_t><<<ew_grid(total), 256, 0, s>>>((const bf16_t*)src, (bf16_t*)dst, N, C, T, H, W, strides[0], strides[1], strides[2], strides[3], strides[4], cpitch);
    else
        GENIE_CHECK_ARG(false, "genie_from_channels_last: unsupported dtype %d", dst_dtype);
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}

// ------------------------------------------------------------------------------------------------
// SiLU forward / backward and a + b on flat bf16 buffers (numel multiple of 8)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) silu_fwd_kernel(const u32x4_t* __restrict__ x, u32x4_t* __restrict__ y, long long n16) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long long)gridDim.x * blockDim.x) {
        float f[8];
        unpack8(x[i], f);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = silu_f(f[j]);
        y[i] = pack8(f);
    }
}

__global__ void __launch_bounds__(256) silu_bwd_kernel(const u32x4_t* __restrict__ x, const u32x4_t* __restrict__ dy,
                                                       u32x4_t* __restrict__ dx, long long n16) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long long)gridDim.x * blockDim.x) {
        float f[8], g[8];
        unpack8(x[i], f);
        unpack8(dy[i], g);
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] *= silu_grad_f(f[j]);
        dx[i] = pack8(g);
    }
}

__global__ void __launch_bounds__(256) add_kernel(const u32x4_t* __restrict__ a, const u32x4_t* __restrict__ b,
                                                  u32x4_t* __restrict__ y, long long n16) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long long)gridDim.x * blockDim.x) {
        float f[8], g[8];
        unpack8(a[i], f);
        unpack8(b[i], g);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] += g[j];
        y[i] = pack8(f);
    }
}

__global__ void __launch_bounds__(256) leaky_fwd_kernel(const u32x4_t* __restrict__ x, u32x4_t* __restrict__ y, long long n16, float slope) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) {
        float f[8];
        unpack8(x[i], f);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = f[j] > 0.f ? f[j] : slope * f[j];
        y[i] = pack8(f);
    }
}
__global__ void __launch_bounds__(256) leaky_bwd_kernel(const u32x4_t* __restrict__ x, const u32x4_t* __restrict__ dy, u32x4_t* __restrict__ dx, long long n16,
                                                        float slope) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) {
        float f[8], g[8];
        unpack8(x[i], f);
        unpack8(dy[i], g);
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] = f[j] > 0.f ? g[j] : slope * g[j];
        dx[i] = pack8(g);
    }
}

extern "C" int genie_silu_fwd(const void* x, void* y, int64_t numel, void* stream) {
    GENIE_CHECK_ARG(numel % 8 == 0, "genie_silu_fwd: numel %lld not a multiple of 8", (long long)numel);
    if (numel == 0) return GENIE_OK;
    silu_fwd_kernel<<<ew_grid(numel / 8), 256, 0, (hipStream_t)stream>>>((const u32x4_t*)x, (u32x4_t*)y, numel / 8);
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}

extern "C" int genie_silu_bwd(const void* x, const void* dy, void* dx, int64_t numel, void* stream) {
    GENIE_CHECK_ARG(numel % 8 == 0, "genie_silu_bwd: numel %lld not a multiple of 8", (long long)numel);
    if (numel == 0) return GENIE_OK;
    silu_bwd_kernel<<<ew_grid(numel / 8), 256, 0, (hipStream_t)stream>>>((const u32x4_t*)x, (const u32x4_t*)dy, (u32x4_t*)dx, numel / 8);
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}

extern "C" int genie_add(const void* a, const void* b, void* y, int64_t numel, void* stream) {
    GENIE_CHECK_ARG(numel % 8 == 0, "genie_add: numel %lld not a multiple of 8", (long long)numel);
    if (numel == 0) return GENIE_OK;
    add_kernel<<<ew_grid(numel / 8), 256, 0, (hipStream_t)stream>>>((const u32x4_t*)a, (const u32x4_t*)b, (u32x4_t*)y, numel / 8);
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}

// ------------------------------------------------------------------------------------------------
// MSE:  loss = mean((rec - target)^2);  rec is CL bf16, target any strided fp32/bf16 NCTHW.
// Deterministic two-stage reduction (fixed partial count, fixed order).
// ------------------------------------------------------------------------------------------------
#define MSE_PARTIALS 1024

template <typename T>
__global__ void __launch_bounds__(256) mse_partial_kernel(const bf16_t* __restrict__ rec, int Cp, const T* __restrict__ tgt,
                                                          int N, int C, int T_, int H, int W, long long sN, long long sC,
                                                          long long sT, long long sH, long long sW, float* __restrict__ partial) {
    const long long npix = (long long)N * T_ * H * W;
    const int nchunk = (C + 7) >> 3;
    const long long total = npix * nchunk;
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long pix = i % npix;
        const int chunk = (int)(i / npix);
        long long r = pix;
        const int w = (int)(r % W); r /= W;
        const int h = (int)(r % H); r /= H;
        const int t = (int)(r % T_); r /= T_;
        const int n = (int)r;
        float f[8];
        unpack8(*reinterpret_cast<const u32x4_t*>(rec + pix * Cp + chunk * 8), f);
        const T* base = tgt + n * sN + t * sT + h * sH + w * sW;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = chunk * 8 + j;
            if (c < C) {
                const float d = f[j] - load_as_f32<T>(base + c * sC);
                acc += d * d;
            }
        }
    }
    __shared__ float red[4];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ void __launch_bounds__(256) mse_final_kernel(const float* __restrict__ partial, int np, float inv_count,
                                                        float* __restrict__ loss) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < np; i += 256) acc += (double)partial[i];
    __shared__ double red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss = (float)(red[0] * (double)inv_count);
}

// drec = (rec - target) * 2/numel * (*gscale)
template <typename T>
__global__ void __launch_bounds__(256) mse_bwd_kernel(const bf16_t* __restrict__ rec, int Cp, const T* __restrict__ tgt, int N,
                                                      int C, int T_, int H, int W, long long sN, long long sC, long long sT,
                                                      long long sH, long long sW, const float* __restrict__ gscale,
                                                      float coef, bf16_t* __restrict__ drec) {
    const long long npix = (long long)N * T_ * H * W;
    const int nchunk = Cp >> 3;
    const long long total = npix * nchunk;
    const float k = coef * (gscale ? *gscale : 1.f);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long pix = i % npix;
        const int chunk = (int)(i / npix);
        long long r = pix;
        const int w = (int)(r % W); r /= W;
        const int h = (int)(r % H); r /= H;
        const int t = (int)(r % T_); r /= T_;
        const int n = (int)r;
        float f[8];
        unpack8(*reinterpret_cast<const u32x4_t*>(rec + pix * Cp + chunk * 8), f);
        const T* base = tgt + n * sN + t * sT + h * sH + w * sW;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = chunk * 8 + j;
            f[j] = c < C ? (f[j] - load_as_f32<T>(base + c * sC)) * k : 0.f;
        }
        *reinterpret_cast<u32x4_t*>(drec + pix * Cp + chunk * 8) = pack8(f);
    }
}

extern "C" int genie_mse_fwd(const void* rec, int cpitch, const void* target, int target_dtype, const int64_t* dims,
                             const int64_t* strides, float* partial_ws, float* loss, void* stream) {
    GENIE_CHECK_ARG(rec && target && partial_ws && loss, "genie_mse_fwd: null pointer");
    const int N = (int)dims[0], C = (int)dims[1], T = (int)dims[2], H = (int)dims[3], W = (int)dims[4];
    const long long count = (long long)N * C * T * H * W;
    GENIE_CHECK_ARG(count > 0, "genie_mse_fwd: empty tensor");
    hipStream_t s = (hipStream_t)stream;
    const long long total = (long long)N * T * H * W * ((C + 7) / 8);
    int grid = ew_grid(total);
    if (grid > MSE_PARTIALS) grid = MSE_PARTIALS;
    if (target_dtype == GENIE_F32)
        mse_partial_kernel<float><<<grid, 256, 0, s>>>((const bf16_t*)rec, cpitch, (const float*)target, N, C, T, H, W, strides[0], strides[1], strides[2], strides[3], strides[4], partial_ws);
    else if (target_dtype == GENIE_BF16)
        mse_partial_kernel<bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)rec, cpitch, (const bf16_t*)target, N, C, T, H, W, strides[0], strides[1], strides[2], strides[3], strides[4], partial_ws);
    else
        GENIE_CHECK_ARG(false, "genie_mse_fwd: unsupported dtype %d", target_dtype);
    GENIE_CHECK_LAUNCH();
    mse_final_kernel<<<1, 256, 0, s>>>(partial_ws, grid, 1.f / (float)count, loss);
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}

extern "C" int genie_mse_bwd(const void* rec, int cpitch, const void* target, int target_dtype, const int64_t* dims,
                             const int64_t* strides, const float* grad_loss, void* drec, void* stream) {
    GENIE_CHECK_ARG(rec && target && drec, "genie_mse_bwd: null pointer");
    const int N = (int)dims[0], C = (int)dims[1], T = (int)dims[2], H = (int)dims[3], W = (int)dims[4];
    const long long count = (long long)N * C * T * H * W;
    GENIE_CHECK_ARG(count > 0, "genie_mse_bwd: empty tensor");
    hipStream_t s = (hipStream_t)stream;
    const long long total = (long long)N * T * H * W * (cpitch / 8);
    const float coef = 2.f / (float)count;
    if (target_dtype == GENIE_F32)
        mse_bwd_kernel<float><<<ew_grid(total), 256, 0, s>>>((const bf16_t*)rec, cpitch, (const float*)target, N, C, T, H, W, strides[0], strides[1], strides[2], strides[3], strides[4], grad_loss, coef, (bf16_t*)drec);
    else if (target_dtype == GENIE_BF16)
        mse_bwd_kernel<bf16_t><<<ew_grid(total), 256, 0, s>>>((const bf16_t*)rec, cpitch, (const bf16_t*)target, N, C, T, H, W, strides[0], strides[1], strides[2], strides[3], strides[4], grad_loss, coef, (bf16_t*)drec);
    else
        GENIE_CHECK_ARG(false, "genie_mse_bwd: unsupported dtype %d", target_dtype);
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}

// ------------------------------------------------------------------------------------------------
// Fused AdamW over a flat fp32 arena (torch.optim.AdamW semantics, decoupled weight decay).
// One pass: reads p, g, m, v; writes p, m, v (+ optionally zeroes g: the wgrad kernels accumulate
// into g with atomics, so "consume and clear" saves a separate memset pass over 1.5 GB).
// ------------------------------------------------------------------------------------------------
// `dev` != nullptr (genie_adamw_step_graph): step size, decay and the second-moment correction come from device memory ({lr / bc1,
// 1 - lr * wd, 1 / sqrt(bc2)}, written by adamw_advance_kernel just before), so a captured launch stays valid from one replay to the next.
__global__ void __launch_bounds__(256) adamw_kernel(float4* __restrict__ p, float4* __restrict__ g, float4* __restrict__ m,
                                                    float4* __restrict__ v, long long n4, float lr, float beta1, float beta2,
                                                    float eps, float wd, float bc1, float rsqrt_bc2, float gscale,
                                                    int zero_grad, u32x2_t* __restrict__ mirror, const float* __restrict__ dev) {
    const float step = dev ? dev[0] : lr / bc1;
    const float decay = dev ? dev[1] : 1.f - lr * wd;
    if (dev) rsqrt_bc2 = dev[2];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        float4 pp = p[i], gg = g[i], mm = m[i], vv = v[i];
        float* P = &pp.x; float* G = &gg.x; float* M = &mm.x; float* V = &vv.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gr = G[j] * gscale;
            P[j] *= decay;
            M[j] = beta1 * M[j] + (1.f - beta1) * gr;
            V[j] = beta2 * V[j] + (1.f - beta2) * gr * gr;
            const float denom = sqrtf(V[j]) * rsqrt_bc2 + eps;
            P[j] -= step * (M[j] / denom);
        }
        p[i] = pp; m[i] = mm; v[i] = vv;
        if (zero_grad) g[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (mirror) {                                   // bf16 image of the updated parameters (the conv kernels' weight packs)
            u32x2_t o;
            o[0] = pack_bf16x2(pp.x, pp.y);
            o[1] = pack_bf16x2(pp.z, pp.w);
            mirror[i] = o;
        }
    }
}

static int adamw_launch(float* p, float* g, float* m, float* v, int64_t numel, float lr, float beta1, float beta2, float eps,
                        float weight_decay, int step, float grad_scale, int zero_grad, void* mirror, void* stream) {
    GENIE_CHECK_ARG(p && g && m && v, "genie_adamw_step: null pointer");
    GENIE_CHECK_ARG(numel % 4 == 0, "genie_adamw_step: arena numel %lld must be a multiple of 4", (long long)numel);
    GENIE_CHECK_ARG(step >= 1, "genie_adamw_step: step must be >= 1");
    if (numel == 0) return GENIE_OK;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    adamw_kernel<<<ew_grid(numel / 4), 256, 0, (hipStream_t)stream>>>((float4*)p, (float4*)g, (float4*)m, (float4*)v, numel / 4, lr, beta1, beta2, eps, weight_decay, (float)bc1, (float)(1.0 / sqrt(bc2)), grad_scale, zero_grad, (u32x2_t*)mirror, nullptr);
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}

// state: [0] step count (int32 bits), [1] lr, [2] weight decay (host-written), [3..5] {lr / bc1, 1 - lr * wd, 1 / sqrt(bc2)} of the step
__global__ void adamw_advance_kernel(float* __restrict__ state, float beta1, float beta2) {
    int* cnt = reinterpret_cast<int*>(state);
    const int step = *cnt + 1;
    *cnt = step;
    const float lr = state[1], wd = state[2];
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    state[3] = (float)((double)lr / bc1);
    state[4] = 1.f - lr * wd;
    state[5] = (float)(1.0 / sqrt(bc2));
}

extern "C" int genie_adamw_step_graph(float* p, float* g, float* m, float* v, void* p_bf16, int64_t numel, float* state, float beta1,
                                      float beta2, float eps, float grad_scale, int zero_grad, void* stream) {
    GENIE_CHECK_ARG(p && g && m && v && state, "genie_adamw_step_graph: null pointer");
    GENIE_CHECK_ARG(numel % 4 == 0, "genie_adamw_step_graph: arena numel %lld must be a multiple of 4", (long long)numel);
    if (numel == 0) return GENIE_OK;
    adamw_advance_kernel<<<1, 1, 0, (hipStream_t)stream>>>(state, beta1, beta2);
    GENIE_CHECK_LAUNCH();
    adamw_kernel<<<ew_grid(numel / 4), 256, 0, (hipStream_t)stream>>>((float4*)p, (float4*)g, (float4*)m, (float4*)v, numel / 4, 0.f, beta1, beta2, eps, 0.f, 1.f, 1.f, grad_scale, zero_grad, (u32x2_t*)p_bf16, state + 3);
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}

extern "C" int genie_adamw_step(float* p, float* g, float* m, float* v, int64_t numel, float lr, float beta1, float beta2,
                                float eps, float weight_decay, int step, float grad_scale, int zero_grad, void* stream) {
    return adamw_launch(p, g, m, v, numel, lr, beta1, beta2, eps, weight_decay, step, grad_scale, zero_grad, nullptr, stream);
}

extern "C" int genie_adamw_step_mirror(float* p, float* g, float* m, float* v, void* p_bf16, int64_t numel, float lr, float beta1,
                                       float beta2, float eps, float weight_decay, int step, float grad_scale, int zero_grad,
                                       void* stream) {
    GENIE_CHECK_ARG(p_bf16, "genie_adamw_step_mirror: null mirror");
    return adamw_launch(p, g, m, v, numel, lr, beta1, beta2, eps, weight_decay, step, grad_scale, zero_grad, p_bf16, stream);
}

// ------------------------------------------------------------------------------------------------
// Weight packing: strided fp32 (R, J, K) -> dense bf16 [R][J][Kp], Kp = roundup8(K), zero padded.
// Optional permutation of the K index: natural k = (k' % permC) * permF + k' / permC  (depth-to-space
// column order, DESIGN.md "pixel shuffle").  Used for the forward pack (R=cout, K=cin) and the
// transposed pack for dgrad (R=cin, K=cout).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) pack_weight_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int R, int J,
                                                          int K, int Kp, long long sR, long long sJ, long long sK, int permC,
                                                          int permF) {
    const long long total = (long long)R * J * (Kp >> 3);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int kc = (int)(i % (Kp >> 3));
        const long long rj = i / (Kp >> 3);
        const int j = (int)(rj % J);
        const int r = (int)(rj / J);
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int kk = kc * 8 + e;
            if (kk < K) {
                const int kn = permF > 1 ? (kk % permC) * permF + kk / permC : kk;
                f[e] = src[r * sR + j * sJ + kn * sK];
            } else {
                f[e] = 0.f;
            }
        }
        *reinterpret_cast<u32x4_t*>(dst + i * 8) = pack8(f);
    }
}

extern "C" int genie_pack_weight(const float* src, void* dst, int R, int J, int K, int64_t sR, int64_t sJ, int64_t sK, int perm_c,
                                 int perm_f, void* stream) {
    GENIE_CHECK_ARG(src && dst, "genie_pack_weight: null pointer");
    GENIE_CHECK_ARG(R > 0 && J > 0 && K > 0, "genie_pack_weight: bad dims %d %d %d", R, J, K);
    GENIE_CHECK_ARG(perm_f >= 1 && (perm_f == 1 || (perm_c >= 1 && perm_c * perm_f == K)), "genie_pack_weight: bad permutation %d x %d for K=%d", perm_c, perm_f, K);
    const int Kp = (K + 7) & ~7;
    const long long total = (long long)R * J * (Kp >> 3);
    pack_weight_kernel<<<ew_grid(total), 256, 0, (hipStream_t)stream>>>(src, (bf16_t*)dst, R, J, K, Kp, sR, sJ, sK, perm_c, perm_f);
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}

// fp32 -> bf16 flat cast (arena mirror)
__global__ void __launch_bounds__(256) cast_bf16_kernel(const float4* __restrict__ src, u32x2_t* __restrict__ dst, long long n4) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const float4 f = src[i];
        u32x2_t o;
        o[0] = pack_bf16x2(f.x, f.y);
        o[1] = pack_bf16x2(f.z, f.w);
        dst[i] = o;
    }
}

extern "C" int genie_cast_f32_to_bf16(const float* src, void* dst, int64_t numel, void* stream) {
    GENIE_CHECK_ARG(numel % 4 == 0, "genie_cast_f32_to_bf16: numel %lld must be a multiple of 4", (long long)numel);
    if (numel == 0) return GENIE_OK;
    cast_bf16_kernel<<<ew_grid(numel / 4), 256, 0, (hipStream_t)stream>>>((const float4*)src, (u32x2_t*)dst, numel / 4);
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}

// ------------------------------------------------------------------------------------------------
// Batched transposed weight pack: for every job, bf16 [R = cout][J = tap][K = cin] (dense, the layout of a channels_last_3d
// nn.Conv3d weight inside the bf16 parameter mirror) -> bf16 [cin][tap][Rp], Rp = roundup8(cout), zero padded, with the optional
// depth-to-space column order (column r' holds natural row (r' % permC) * permF + r' / permC).  One launch for ALL convolutions
// of a model: blocks are dealt to jobs through the `first_block` prefix (binary search), each block transposes a 64 x 64 tile
// of one tap through LDS so that both the reads (64 cin = 128 B) and the writes (64 cout = 128 B) are row-contiguous.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) pack_transpose_batched_kernel(const GeniePackJob* __restrict__ jobs, int njobs,
                                                                     const bf16_t* __restrict__ src, bf16_t* __restrict__ dst) {
    __shared__ bf16_t tile[64][66];
    int lo = 0, hi = njobs - 1;
    const int bid = blockIdx.x;
    while (lo < hi) {                                   // last job whose first_block <= bid
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first_block <= bid) lo = mid; else hi = mid - 1;
    }
    const GeniePackJob jb = jobs[lo];
    int t = bid - jb.first_block;
    const int tk = t % jb.tiles_k; t /= jb.tiles_k;
    const int j = t % jb.J;
    const int tr = t / jb.J;
    const int r0 = tr * 64, k0 = tk * 64;
    const int Rp = (jb.R + 7) & ~7;
    const bf16_t* s = src + jb.src_off;
    bf16_t* d = dst + jb.dst_off;
    const int tid = threadIdx.x;
    // load: rows r' = r0 + 32 * pass + tid / 8 (destination column order), 8 consecutive k per thread
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int rl = pass * 32 + (tid >> 3), kc = (tid & 7) * 8;
        const int rp = r0 + rl;
        u32x4_t v = {0u, 0u, 0u, 0u};
        if (rp < jb.R && k0 + kc < jb.K) {
            const int rn = jb.perm_f > 1 ? (rp % jb.perm_c) * jb.perm_f + rp / jb.perm_c : rp;
            v = *reinterpret_cast<const u32x4_t*>(s + ((long long)rn * jb.J + j) * jb.K + k0 + kc);
        }
        uint32_t* row = reinterpret_cast<uint32_t*>(&tile[rl][kc]);    // rows are 132 B: 4-byte aligned, kc even
        row[0] = v[0]; row[1] = v[1]; row[2] = v[2]; row[3] = v[3];
    }
    __syncthreads();
    // store: rows k = k0 + 32 * pass + tid / 8, 8 consecutive r' per thread
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int kl = pass * 32 + (tid >> 3), rc = (tid & 7) * 8;
        const int k = k0 + kl;
        if (k < jb.K && r0 + rc < Rp) {
            u32x4_t o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (uint32_t)tile[rc + 2 * e][kl] | ((uint32_t)tile[rc + 2 * e + 1][kl] << 16);
            *reinterpret_cast<u32x4_t*>(d + ((long long)k * jb.J + j) * Rp + r0 + rc) = o;
        }
    }
}

extern "C" int genie_pack_transpose_batched(const GeniePackJob* jobs_dev, int njobs, int total_blocks, const void* src_bf16,
                                            void* dst_bf16, void* stream) {
    GENIE_CHECK_ARG(jobs_dev && src_bf16 && dst_bf16, "genie_pack_transpose_batched: null pointer");
    GENIE_CHECK_ARG(njobs >= 1 && total_blocks >= 1, "genie_pack_transpose_batched: empty job list");
    pack_transpose_batched_kernel<<<total_blocks, 256, 0, (hipStream_t)stream>>>(jobs_dev, njobs, (const bf16_t*)src_bf16, (bf16_t*)dst_bf16);
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}

// ------------------------------------------------------------------------------------------------
// BlurPooling3d (reference genie/module/video.py:487-537).  With num_groups = 1 the reference runs a DENSE conv3d whose every
// (out, in) tap is the same Pascal blur kernel, i.e. every output channel is the strided blur of the SUM over the input
// channels.  Written as what it is -- HBM-bound: one pass that sums the channels of each pixel
// (reads x once), a stencil over that C-times-smaller fp32 field, and a broadcast store of the result to all output channels --
// instead of a 2 M C O 27-FLOP GEMM.  Backward is the same three steps transposed.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) chan_sum_kernel(const bf16_t* __restrict__ x, long long npix, int C, int Cp, float* __restrict__ out) {
    // a group of L = 8 lanes per pixel walks the pixel's 16-B chunks; 8 pixels per wave-instruction
    // only the chunks that hold channels: x may be a channel-slice VIEW (the grouped blur: C = one group, Cp = the pitch of the whole tensor),
    // and walking the pitch from the view's first channel ran past the end of the tensor on its last pixels (found with rocgdb as a memory
    // fault that depended on where the allocator had placed the tensor)
    const int lane = threadIdx.x & 63, sub = lane & 7;
#ifdef GENIE_REINTRODUCE_CHANSUM_OOB
    const int nch = Cp >> 3;                              // round 4's bug, kept ONLY for lib/libgenie_hip_oobprobe.so: the guard harness must catch it
#else
    const int nch = (C + 7) >> 3;
#endif
    for (long long p = ((long long)blockIdx.x * 256 + threadIdx.x) >> 3; p < npix; p += ((long long)gridDim.x * 256) >> 3) {
        float s = 0.f;
        for (int ch = sub; ch < nch; ch += 8) {
            float f[8];
            unpack8(*reinterpret_cast<const u32x4_t*>(x + p * Cp + ch * 8), f);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += (ch * 8 + j < C) ? f[j] : 0.f;
        }
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 4, 64);
        if (sub == 0) out[p] = s;
    }
}

struct BlurGeom {
    int N, T, H, W, To, Ho, Wo;
    int kt, kh, kw, st, sh, sw, pt, ph, pw;
};

// out[n, :, to, ho, wo] = sum_taps k[a, b, c] * s[n, to st + a - pt, ho sh + b - ph, wo sw + c - pw]   (zero outside)
__global__ void __launch_bounds__(256) blur_stencil_fwd_kernel(const float* __restrict__ s, const float* __restrict__ taps, BlurGeom g,
                                                               bf16_t* __restrict__ out, int O, int Op) {
    const int nch = Op >> 3;
    const long long total = (long long)g.N * g.To * g.Ho * g.Wo * nch;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ch = (int)(i % nch);
        long long p = i / nch;
        const int wo = (int)(p % g.Wo); p /= g.Wo;
        const int ho = (int)(p % g.Ho); p /= g.Ho;
        const int to = (int)(p % g.To);
        const int n = (int)(p / g.To);
        float v = 0.f;
        for (int a = 0; a < g.kt; ++a) {
            const int t = to * g.st + a - g.pt;
            if ((unsigned)t >= (unsigned)g.T) continue;
            for (int b = 0; b < g.kh; ++b) {
                const int h = ho * g.sh + b - g.ph;
                if ((unsigned)h >= (unsigned)g.H) continue;
                for (int c = 0; c < g.kw; ++c) {
                    const int w = wo * g.sw + c - g.pw;
                    if ((unsigned)w >= (unsigned)g.W) continue;
                    v += taps[(a * g.kh + b) * g.kw + c] * s[(((long long)n * g.T + t) * g.H + h) * g.W + w];
                }
            }
        }
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = (ch * 8 + j < O) ? v : 0.f;
        *reinterpret_cast<u32x4_t*>(out + i * 8) = pack8(f);
    }
}

// dx[n, :, t, h, w] = sum over taps and output positions with to st + a - pt == t (...) of k[a, b, c] * sdy[n, to, ho, wo]
__global__ void __launch_bounds__(256) blur_stencil_bwd_kernel(const float* __restrict__ sdy, const float* __restrict__ taps, BlurGeom g,
                                                               bf16_t* __restrict__ dx, int C, int Cp) {
    const int nch = Cp >> 3;
    const long long total = (long long)g.N * g.T * g.H * g.W * nch;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ch = (int)(i % nch);
        long long p = i / nch;
        const int w = (int)(p % g.W); p /= g.W;
        const int h = (int)(p % g.H); p /= g.H;
        const int t = (int)(p % g.T);
        const int n = (int)(p / g.T);
        float v = 0.f;
        for (int a = 0; a < g.kt; ++a) {
            const int tn = t + g.pt - a;
            if (tn < 0 || tn % g.st != 0 || tn / g.st >= g.To) continue;
            for (int b = 0; b < g.kh; ++b) {
                const int hn = h + g.ph - b;
                if (hn < 0 || hn % g.sh != 0 || hn / g.sh >= g.Ho) continue;
                for (int c = 0; c < g.kw; ++c) {
                    const int wn = w + g.pw - c;
                    if (wn < 0 || wn % g.sw != 0 || wn / g.sw >= g.Wo) continue;
                    v += taps[(a * g.kh + b) * g.kw + c] * sdy[(((long long)n * g.To + tn / g.st) * g.Ho + hn / g.sh) * g.Wo + wn / g.sw];
                }
            }
        }
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = (ch * 8 + j < C) ? v : 0.f;
        *reinterpret_cast<u32x4_t*>(dx + i * 8) = pack8(f);
    }
}

static int blur_check(const int64_t* dims, const int* kernel, const int* stride, const int* pad, BlurGeom& g) {
    GENIE_CHECK_ARG(dims && kernel && stride && pad, "genie_blur_pool3d: null geometry");
    g.N = (int)dims[0]; g.T = (int)dims[2]; g.H = (int)dims[3]; g.W = (int)dims[4];
    g.kt = kernel[0]; g.kh = kernel[1]; g.kw = kernel[2]; g.st = stride[0]; g.sh = stride[1]; g.sw = stride[2];
    g.pt = pad[0]; g.ph = pad[1]; g.pw = pad[2];
    GENIE_CHECK_ARG(g.kt >= 1 && g.kh >= 1 && g.kw >= 1 && g.st >= 1 && g.sh >= 1 && g.sw >= 1, "genie_blur_pool3d: bad kernel / stride");
    g.To = (g.T + 2 * g.pt - g.kt) / g.st + 1; g.Ho = (g.H + 2 * g.ph - g.kh) / g.sh + 1; g.Wo = (g.W + 2 * g.pw - g.kw) / g.sw + 1;
    GENIE_CHECK_ARG(g.To >= 1 && g.Ho >= 1 && g.Wo >= 1, "genie_blur_pool3d: input smaller than the kernel");
    return GENIE_OK;
}

extern "C" int genie_blur_pool3d_fwd(const void* x_cl, int cpitch, const int64_t* dims, const float* taps, const int* kernel,
                                     const int* stride, const int* pad, void* out_cl, int out_channels, int out_pitch, float* ws,
                                     void* stream) {
    GENIE_CHECK_ARG(x_cl && taps && out_cl && ws, "genie_blur_pool3d_fwd: null pointer");
    BlurGeom g;
    if (int rc = blur_check(dims, kernel, stride, pad, g)) return rc;
    const int C = (int)dims[1];
    GENIE_CHECK_ARG(cpitch % 8 == 0 && cpitch >= C && out_pitch % 8 == 0 && out_pitch >= out_channels, "genie_blur_pool3d_fwd: bad channel pitch");
    const long long npix = (long long)g.N * g.T * g.H * g.W;
    hipStream_t s = (hipStream_t)stream;
    chan_sum_kernel<<<ew_grid(npix * 8), 256, 0, s>>>((const bf16_t*)x_cl, npix, C, cpitch, ws);
    GENIE_CHECK_LAUNCH();
    const long long total = (long long)g.N * g.To * g.Ho * g.Wo * (out_pitch >> 3);
    blur_stencil_fwd_kernel<<<ew_grid(total), 256, 0, s>>>(ws, taps, g, (bf16_t*)out_cl, out_channels, out_pitch);
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}

extern "C" int genie_blur_pool3d_bwd(const void* dy_cl, int out_channels, int out_pitch, const int64_t* dims, const float* taps,
                                     const int* kernel, const int* stride, const int* pad, void* dx_cl, int cpitch, float* ws,
                                     void* stream) {
    GENIE_CHECK_ARG(dy_cl && taps && dx_cl && ws, "genie_blur_pool3d_bwd: null pointer");
    BlurGeom g;
    if (int rc = blur_check(dims, kernel, stride, pad, g)) return rc;
    const int C = (int)dims[1];
    GENIE_CHECK_ARG(cpitch % 8 == 0 && cpitch >= C && out_pitch % 8 == 0 && out_pitch >= out_channels, "genie_blur_pool3d_bwd: bad channel pitch");
    const long long nout = (long long)g.N * g.To * g.Ho * g.Wo;
    hipStream_t s = (hipStream_t)stream;
    chan_sum_kernel<<<ew_grid(nout * 8), 256, 0, s>>>((const bf16_t*)dy_cl, nout, out_channels, out_pitch, ws);
    GENIE_CHECK_LAUNCH();
    const long long total = (long long)g.N * g.T * g.H * g.W * (cpitch >> 3);
    blur_stencil_bwd_kernel<<<ew_grid(total), 256, 0, s>>>(ws, taps, g, (bf16_t*)dx_cl, C, cpitch);
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}

// GELU, the exact (erf) form of nn.GELU() -- the default activation of the reference's ForwardBlock (misc.py:78), reached through
// SpaceTimeAttention(hid_dim=...) (attention.py:429-438).  gelu(x) = x Phi(x);  gelu'(x) = Phi(x) + x phi(x).
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.39894228040143268f * __expf(-0.5f * x * x);
}
__global__ void __launch_bounds__(256) gelu_fwd_kernel(const u32x4_t* __restrict__ x, u32x4_t* __restrict__ y, long long n16) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) {
        float f[8];
        unpack8(x[i], f);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = gelu_f(f[j]);
        y[i] = pack8(f);
    }
}
__global__ void __launch_bounds__(256) gelu_bwd_kernel(const u32x4_t* __restrict__ x, const u32x4_t* __restrict__ dy, u32x4_t* __restrict__ dx, long long n16) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) {
        float f[8], g[8];
        unpack8(x[i], f);
        unpack8(dy[i], g);
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] *= gelu_grad_f(f[j]);
        dx[i] = pack8(g);
    }
}
extern "C" int genie_gelu_fwd(const void* x, void* y, int64_t numel, void* stream) {
    GENIE_CHECK_ARG(x && y && numel % 8 == 0, "genie_gelu_fwd: null pointer or numel %lld not a multiple of 8", (long long)numel);
    if (numel == 0) return GENIE_OK;
    gelu_fwd_kernel<<<ew_grid(numel / 8), 256, 0, (hipStream_t)stream>>>((const u32x4_t*)x, (u32x4_t*)y, numel / 8);
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}
extern "C" int genie_gelu_bwd(const void* x, const void* dy, void* dx, int64_t numel, void* stream) {
    GENIE_CHECK_ARG(x && dy && dx && numel % 8 == 0, "genie_gelu_bwd: null pointer or numel %lld not a multiple of 8", (long long)numel);
    if (numel == 0) return GENIE_OK;
    gelu_bwd_kernel<<<ew_grid(numel / 8), 256, 0, (hipStream_t)stream>>>((const u32x4_t*)x, (const u32x4_t*)dy, (u32x4_t*)dx, numel / 8);
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}

// LeakyReLU over a CL buffer (reference: nn.LeakyReLU in ImageResidualBlock image.py:118-131 and FrameDiscriminator.to_logits discriminator.py:91)
extern "C" int genie_leaky_relu_fwd(const void* x, void* y, int64_t numel, float slope, void* stream) {
    GENIE_CHECK_ARG(x && y && numel % 8 == 0, "genie_leaky_relu_fwd: null pointer or numel %lld not a multiple of 8", (long long)numel);
    if (numel == 0) return GENIE_OK;
    leaky_fwd_kernel<<<ew_grid(numel / 8), 256, 0, (hipStream_t)stream>>>((const u32x4_t*)x, (u32x4_t*)y, numel / 8, slope);
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}
extern "C" int genie_leaky_relu_bwd(const void* x, const void* dy, void* dx, int64_t numel, float slope, void* stream) {
    GENIE_CHECK_ARG(x && dy && dx && numel % 8 == 0, "genie_leaky_relu_bwd: null pointer or numel %lld not a multiple of 8", (long long)numel);
    if (numel == 0) return GENIE_OK;
    leaky_bwd_kernel<<<ew_grid(numel / 8), 256, 0, (hipStream_t)stream>>>((const u32x4_t*)x, (const u32x4_t*)dy, (u32x4_t*)dx, numel / 8, slope);
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}

// ------------------------------------------------------------------------------------------------------------------------------
// Embedding lookup and its sparse backward (reference: nn.Embedding tok_emb / act_emb of DynamicsModel, genie/dynamics.py:31-38, :52-55).
// Forward: out[n][:] = weight[idx[n]][:] (fp32 rows, D % 4 == 0), one wave per row.  Backward: grad[idx[n]][:] += dy[n][:] for n < N, straight into
// the parameter's gradient buffer.  The MaskGIT loss fills three quarters of the grid with ONE token id (dynamics.py:86), so a scatter of one
// atomic per (row, column) is ~25 000 same-address atomics per column; here a workgroup takes 64 consecutive rows, sorts their (index, row)
// pairs in LDS and walks them in index order: one atomic per DISTINCT index per column and workgroup (torch's index_add_: 0.6 ms at 32768 rows
// x 512 columns; this: the two passes over dy).
// ------------------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) embedding_fwd_kernel(const long long* __restrict__ idx, const float* __restrict__ weight, float* __restrict__ out,
                                                            long long N, int D, long long V) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
    for (long long n = wave0; n < N; n += nwaves) {
        const long long t = idx[n];
        const bool ok = t >= 0 && t < V;                     // nn.Embedding raises on an index outside [0, V); a kernel cannot: the row is poisoned
        const float4* src = reinterpret_cast<const float4*>(weight + (ok ? t : 0) * D);
        float4* dst = reinterpret_cast<float4*>(out + n * D);
        for (int c = lane; c < D / 4; c += 64) {
            float4 v = src[c];
            if (!ok) v = make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""));
            dst[c] = v;
        }
    }
}

__device__ __forceinline__ float emb_ld(const float* p, long long i) { return p[i]; }
__device__ __forceinline__ float emb_ld(const bf16_t* p, long long i) { return bf16_to_f32(p[i]); }
template <typename T>
__global__ void __launch_bounds__(256) embedding_bwd_kernel(const long long* __restrict__ idx, const T* __restrict__ dy, float* __restrict__ grad,
                                                            long long N, int D, long long V) {
    __shared__ unsigned long long key[64];                   // (index << 6 | row in the chunk): sorting the keys sorts by index, then by row
    const long long n0 = (long long)blockIdx.x * 64;
    const int tid = threadIdx.x;
    if (tid < 64) {
        const long long n = n0 + tid;
        long long t = n < N ? idx[n] : -1;
        if (t < 0 || t >= V) t = V;                          // out of range / past the end: sorted to the back, skipped
        key[tid] = ((unsigned long long)t << 6) | (unsigned)tid;
    }
    __syncthreads();
    // bitonic sort of 64 keys by 32 threads
    for (int k = 2; k <= 64; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (tid < 32) {
                const int i = 2 * tid - (tid & (j - 1));     // lower element of the pair (i, i + j)
                const bool up = (i & k) == 0;
                const unsigned long long a = key[i], b = key[i + j];
                if ((a > b) == up) { key[i] = b; key[i + j] = a; }
            }
            __syncthreads();
        }
    for (int c = tid; c < D; c += 256) {
        float acc = 0.f;
        long long cur = -1;
        for (int p = 0; p < 64; ++p) {
            const unsigned long long kk = key[p];
            const long long t = (long long)(kk >> 6);
            if (t >= V) break;
            if (t != cur) {
                if (cur >= 0) atomicAdd(grad + cur * D + c, acc);
                cur = t; acc = 0.f;
            }
            acc += emb_ld(dy, (n0 + (long long)(kk & 63)) * D + c);
        }
        if (cur >= 0) atomicAdd(grad + cur * D + c, acc);
    }
}

extern "C" int genie_embedding_fwd(const int64_t* idx, const float* weight, float* out, int64_t N, int D, int64_t V, void* stream) {
    GENIE_CHECK_ARG(idx && weight && out, "genie_embedding_fwd: null pointer");
    GENIE_CHECK_ARG(D >= 4 && D % 4 == 0 && V >= 1 && N >= 0, "genie_embedding_fwd: D=%d must be a multiple of 4, V=%lld", D, (long long)V);
    if (N == 0) return GENIE_OK;
    const long long blocks = (N + 3) / 4;
    embedding_fwd_kernel<<<(unsigned)(blocks < 8192 ? blocks : 8192), 256, 0, (hipStream_t)stream>>>((const long long*)idx, weight, out, N, D, V);
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}

extern "C" int genie_embedding_bwd(const int64_t* idx, const void* dy, int dy_dtype, float* grad, int64_t N, int D, int64_t V, void* stream) {
    GENIE_CHECK_ARG(idx && dy && grad, "genie_embedding_bwd: null pointer");
    GENIE_CHECK_ARG(D >= 1 && V >= 1 && V < (1ll << 56) && N >= 0 && (N + 63) / 64 < (1ll << 31), "genie_embedding_bwd: bad geometry");
    GENIE_CHECK_ARG(dy_dtype == GENIE_F32 || dy_dtype == GENIE_BF16, "genie_embedding_bwd: dy dtype %d (fp32 or bf16)", dy_dtype);
    if (N == 0) return GENIE_OK;
    const unsigned blocks = (unsigned)((N + 63) / 64);
    if (dy_dtype == GENIE_F32) embedding_bwd_kernel<float><<<blocks, 256, 0, (hipStream_t)stream>>>((const long long*)idx, (const float*)dy, grad, N, D, V);
    else embedding_bwd_kernel<bf16_t><<<blocks, 256, 0, (hipStream_t)stream>>>((const long long*)idx, (const bf16_t*)dy, grad, N, D, V);
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}
