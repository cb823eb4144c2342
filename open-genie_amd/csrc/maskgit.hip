// MaskGIT sampling step on the device (reference genie/dynamics.py:138-158, with the multinomial draw replaced by an inverse-CDF
// draw from injected uniforms so token ids are reproducible across devices -- oracle/genie_oracle.py::sample_from_uniform).
//
//   sample : per logits row (one of the B*h*w positions of the frame being generated)
//              x_j = logit_j / temp;  m = max_j x_j;  e_j = exp(x_j - m);  total = sum_j e_j (fp32);  p_j = e_j / total (fp32)
//              cdf_j = sum_{i<=j} (double) p_i;  pred = #{ j : cdf_j <= u * cdf_{V-1} }  (clamped to V-1);  conf = p_pred
//            = torch.softmax(logits / temp) -> cumsum(double) -> count, the oracle's rule, without ever writing the (rows, V)
//            probability matrix: the row (512 KB of bf16 at V = 2^18) is streamed four times, three of them out of L2.
//   paint  : per sample, the k positions with the highest confidence among the still-masked ones receive their sampled token
//            (topk + gather + scatter_ of dynamics.py:146-158); rank by counting inside one workgroup, ties -> lower index.
//
// Both kernels only enqueue; the generate loop runs without a device->host sync (the reference syncs twice per step).
#include "common.h"
#include "genie_hip.h"

namespace {

constexpr int SAMPLE_THREADS = 256;
constexpr int SAMPLE_WAVES = SAMPLE_THREADS / 64;

template <typename T>
__device__ __forceinline__ void load8(const T* p, float* f);
template <>
__device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, float* f) {
    unpack8(*reinterpret_cast<const u32x4_t*>(p), f);
}
template <>
__device__ __forceinline__ void load8<float>(const float* p, float* f) {
    const f32x4_t a = *reinterpret_cast<const f32x4_t*>(p), b = *reinterpret_cast<const f32x4_t*>(p + 4);
    f[0] = a[0]; f[1] = a[1]; f[2] = a[2]; f[3] = a[3]; f[4] = b[0]; f[5] = b[1]; f[6] = b[2]; f[7] = b[3];
}
template <typename T>
__device__ __forceinline__ float load1(const T* p);
template <>
__device__ __forceinline__ float load1<bf16_t>(const bf16_t* p) { return bf16_to_f32(*p); }
template <>
__device__ __forceinline__ float load1<float>(const float* p) { return *p; }

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// inclusive prefix sum over the 64 lanes of a wave
__device__ __forceinline__ double wave_scan_f64(double v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}

// the eight x_j of one 16-B group (IEEE division, as torch's `logits / temp`)
template <typename T>
__device__ __forceinline__ void group_x(const T* row, long long j0, long long V, float temp, float* x, bool vec) {
    if (vec && j0 + 8 <= V) {
        load8<T>(row + j0, x);
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = (j0 + i < V) ? load1<T>(row + j0 + i) : -INFINITY;
    }
    if (temp != 1.f) {
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = x[i] / temp;
    }
}

template <typename T>
__global__ void __launch_bounds__(SAMPLE_THREADS) maskgit_sample_kernel(const T* __restrict__ logits, long long rows_per_sample, long long sample_stride,
                                                                          long long pitch, long long V, const float* __restrict__ u, float temp,
                                                                          long long* __restrict__ pred, float* __restrict__ conf, bool vec) {
    __shared__ float s_f[SAMPLE_WAVES];
    __shared__ double s_d[SAMPLE_WAVES];
    const long long r = blockIdx.x;
    const T* row = logits + (r / rows_per_sample) * sample_stride + (r % rows_per_sample) * pitch;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long ngroups = (V + 7) >> 3;

    // pass 1: row maximum
    float m = -INFINITY;
    for (long long g = tid; g < ngroups; g += SAMPLE_THREADS) {
        float x[8];
        group_x<T>(row, g << 3, V, temp, x, vec);
#pragma unroll
        for (int i = 0; i < 8; ++i) m = fmaxf(m, x[i]);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (lane == 0) s_f[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(s_f[0], s_f[1]), fmaxf(s_f[2], s_f[3]));
    __syncthreads();

    // pass 2: softmax denominator.  torch's CPU softmax sums the fp32 exponentials with a vectorised cascade that is accurate to an
    // ulp or two; a plain fp32 running sum per thread over up to 2^18 / 256 terms is not (measured 2e-5 relative), so the partial sums
    // are kept in double and the total is rounded to fp32 once -- the nearest fp32 to the true sum, which is what torch's also is.
    double tot_d = 0.0;
    for (long long g = tid; g < ngroups; g += SAMPLE_THREADS) {
        float x[8];
        group_x<T>(row, g << 3, V, temp, x, vec);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += expf(x[i] - m);          // exp(-inf) = 0 for the tail of a ragged last group
        tot_d += (double)s;
    }
    tot_d = wave_sum_f64(tot_d);
    if (lane == 0) s_d[wave] = tot_d;
    __syncthreads();
    const float tot = (float)((s_d[0] + s_d[1]) + (s_d[2] + s_d[3]));
    __syncthreads();

    // pass 3: each wave owns a contiguous quarter of the row; sum of the fp32 probabilities of that range, in double
    const long long gpw = (ngroups + SAMPLE_WAVES - 1) / SAMPLE_WAVES;       // groups per wave
    const long long g_lo = wave * gpw, g_hi = (g_lo + gpw < ngroups) ? g_lo + gpw : ngroups;
    double part = 0.0;
    for (long long g = g_lo + lane; g < g_hi; g += 64) {
        float x[8];
        group_x<T>(row, g << 3, V, temp, x, vec);
#pragma unroll
        for (int i = 0; i < 8; ++i) part += (double)(expf(x[i] - m) / tot);
    }
    part = wave_sum_f64(part);
    if (lane == 0) s_d[wave] = part;
    __syncthreads();
    double before = 0.0, cdf_last = 0.0;
#pragma unroll
    for (int w = 0; w < SAMPLE_WAVES; ++w) {
        if (w < wave) before += s_d[w];
        cdf_last += s_d[w];
    }
    const double thr = (double)u[r] * cdf_last;
    // the crossing (first j with cdf_j > thr) lies in the first wave range whose inclusive end exceeds thr; when no range
    // does (u * total >= total: u = 1 or rounding), the count is V and clamps to V - 1
    const double after = before + s_d[wave];
    const bool mine = (before <= thr) && (after > thr);
    // exactly one wave can have `mine` (ranges are disjoint and monotone); the last wave handles the clamp case
    if (!mine) {
        if (wave == SAMPLE_WAVES - 1 && lane == 0 && !(cdf_last > thr)) {
            const long long j = V - 1;
            pred[r] = j;
            conf[r] = expf((temp != 1.f ? load1<T>(row + j) / temp : load1<T>(row + j)) - m) / tot;
        }
        return;
    }
    // pass 4 (one wave): walk the range 64 groups at a time with a wave scan of the per-lane group sums
    double run = before;
    for (long long gb = g_lo; gb < g_hi; gb += 64) {          // wave-uniform loop
        const long long g = gb + lane;
        float p[8];
        double gs = 0.0;
        if (g < g_hi) {
            float x[8];
            group_x<T>(row, g << 3, V, temp, x, vec);
#pragma unroll
            for (int i = 0; i < 8; ++i) { p[i] = expf(x[i] - m) / tot; gs += (double)p[i]; }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) p[i] = 0.f;
        }
        const double incl = wave_scan_f64(gs, lane);
        const double step_total = __shfl(incl, 63, 64);
        const bool last_step = gb + 64 >= g_hi;
        if (run + step_total > thr || last_step) {
            // the crossing is in this step: the first lane whose inclusive prefix exceeds thr holds it
            const bool over = (run + incl > thr) && (g < g_hi);
            const unsigned long long ball = __ballot(over);
            int src = ball ? __ffsll((long long)ball) - 1 : -1;
            if (src < 0) {
                // rounding of the regrouped sums moved the crossing past this range's end: take the range's last element
                if (lane == 0) {
                    long long j = (g_hi << 3) - 1;
                    if (j > V - 1) j = V - 1;
                    pred[r] = j;
                    conf[r] = expf((temp != 1.f ? load1<T>(row + j) / temp : load1<T>(row + j)) - m) / tot;
                }
                return;
            }
            if (lane == src) {
                double c = run + (incl - gs);                 // cdf just before this group
                int hit = -1;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    c += (double)p[i];
                    if (hit < 0 && c > thr) hit = i;
                }
                if (hit < 0) hit = 7;                         // (re-association of the double sums: at most one element off)
                long long j = (g << 3) + hit;
                if (j > V - 1) { hit -= (int)(j - (V - 1)); j = V - 1; }
                float ph = p[0];
#pragma unroll
                for (int i = 1; i < 8; ++i) ph = (i == hit) ? p[i] : ph;
                pred[r] = j;
                conf[r] = ph;
            }
            return;
        }
        run += step_total;
    }
}

// one workgroup per sample: rank every position by (confidence desc, index asc) among the masked ones and paint the top k
__global__ void __launch_bounds__(256) maskgit_paint_kernel(const float* __restrict__ conf, const long long* __restrict__ pred, int n, int k,
                                                            long long* __restrict__ code, unsigned char* __restrict__ mask) {
    extern __shared__ float s_conf[];
    const long long base = (long long)blockIdx.x * n;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s_conf[i] = mask[base + i] ? conf[base + i] : -INFINITY;   // conf[~mask] = -inf
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float c = s_conf[i];
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const float o = s_conf[j];                    // same address across the wave: LDS broadcast
            rank += (o > c) || (o == c && j < i);
        }
        if (rank < k) {
            code[base + i] = pred[base + i];
            mask[base + i] = 0;
        }
    }
}

}  // namespace

extern "C" int genie_maskgit_sample(const void* logits, int dtype, int64_t rows, int64_t rows_per_sample, int64_t sample_stride, int64_t pitch, int64_t V,
                                    const float* u, float temp, int64_t* pred, float* conf, void* stream) {
    GENIE_CHECK_ARG(logits && u && pred && conf, "genie_maskgit_sample: null pointer");
    GENIE_CHECK_ARG(dtype == GENIE_BF16 || dtype == GENIE_F32, "genie_maskgit_sample: dtype %d", dtype);
    GENIE_CHECK_ARG(rows >= 1 && rows_per_sample >= 1 && V >= 1 && rows < (1ll << 31), "genie_maskgit_sample: empty or oversized problem");
    GENIE_CHECK_ARG(pitch >= V, "genie_maskgit_sample: row pitch %lld < V %lld", (long long)pitch, (long long)V);
    const bool vec = pitch % 8 == 0 && sample_stride % 8 == 0 && ((uintptr_t)logits & 15) == 0;      // 16-B loads, else element loads
    GENIE_CHECK_ARG(temp > 0.f, "genie_maskgit_sample: temperature must be positive (got %g)", (double)temp);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GENIE_BF16)
        maskgit_sample_kernel<bf16_t><<<(unsigned)rows, SAMPLE_THREADS, 0, s>>>((const bf16_t*)logits, rows_per_sample, sample_stride, pitch, V, u, temp,
                                                                               (long long*)pred, conf, vec);
    else
        maskgit_sample_kernel<float><<<(unsigned)rows, SAMPLE_THREADS, 0, s>>>((const float*)logits, rows_per_sample, sample_stride, pitch, V, u, temp,
                                                                              (long long*)pred, conf, vec);
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}

extern "C" int genie_maskgit_paint(const float* conf, const int64_t* pred, int64_t batch, int64_t n, int64_t k, int64_t* code, unsigned char* mask,
                                   void* stream) {
    GENIE_CHECK_ARG(conf && pred && code && mask, "genie_maskgit_paint: null pointer");
    GENIE_CHECK_ARG(batch >= 1 && n >= 1 && n <= 32768 && k >= 0 && k <= n, "genie_maskgit_paint: batch %lld, n %lld (<= 32768), k %lld out of range",
                    (long long)batch, (long long)n, (long long)k);
    if (k == 0) return GENIE_OK;
    const size_t lds = (size_t)n * sizeof(float);
    if (lds > 65536)
        GENIE_CHECK_ARG(hipFuncSetAttribute((const void*)maskgit_paint_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess,
                        "hipFuncSetAttribute failed");
    maskgit_paint_kernel<<<(unsigned)batch, 256, lds, (hipStream_t)stream>>>(conf, (const long long*)pred, (int)n, (int)k, (long long*)code, mask);
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}
