// Skinny linears of the hot path -- y = x W^T + b with min(in_features, out_features) <= 32 -- in fp32 arithmetic (reference call sites:
//   AdaptiveGroupNorm's Linear(18 -> C) pair        genie/module/norm.py:55-69
//   LookupFreeQuantization.proj_inp / proj_out      genie/module/quantization.py:52-58   (Linear(512 <-> 10) of config/tokenize.yaml: 16384 B rows)
//   Adapter.to_k / to_v on a conditioning vector    genie/module/attention.py:128-129    (Linear(8 -> C))
//   LatentAction.to_act                              genie/action.py:83-90                (Linear(C/4 H W = 2^18 .. 2^20 -> 8))
// ).  Rounds 1-5 left these to F.linear: a few MFLOP each, but library GEMMs on the product path.  None of them is matrix-pipe work: one side of
// the weight is <= 32 wide, so every case is a stream over the LONG operand with the short one in registers.  Three bodies:
//   rowdot   (out <= 32): a wave owns a slice of the reduction axis (its weight slice in registers: out x 8 x steps floats per lane) and walks rows;
//            one slice -> wave reduction and a direct store; more slices (to_act: 256 x 1024 of 2^18) -> partials [slice][row][out] + a fixed-order sum
//   expand   (in <= 32):  a thread owns output columns (their weight rows in registers) and walks rows; the input row is a broadcast
//   wgrad    dW[n][k] = sum_m dy[m][n] x[m][k]: a thread owns an index of the long axis and <= 32 accumulators of the short one, rows are split
//            over workgroups, partials [split][..] are summed in a fixed order INTO the gradient buffer (no atomics: bit-reproducible)
// The backward-data pass is the forward with the weight's strides exchanged.  x / dy / y: fp32 or bf16 rows with a pitch; W, bias, gradients: fp32.
#include "common.h"
#include "genie_hip.h"

namespace {

constexpr int LS_MAXS = 32;            // the short side

template <typename T> __device__ __forceinline__ float ls_ld(const T* p, long long i);
template <> __device__ __forceinline__ float ls_ld<float>(const float* p, long long i) { return p[i]; }
template <> __device__ __forceinline__ float ls_ld<bf16_t>(const bf16_t* p, long long i) { return bf16_to_f32(p[i]); }
template <typename T> __device__ __forceinline__ void ls_st(T* p, long long i, float v);
template <> __device__ __forceinline__ void ls_st<float>(float* p, long long i, float v) { p[i] = v; }
template <> __device__ __forceinline__ void ls_st<bf16_t>(bf16_t* p, long long i, float v) { p[i] = f32_to_bf16(v); }

struct LsArgs {
    const void* x; long long x_pitch; long long M; int K;
    const float* W; long long w_sn, w_sk;          // W[n][k] = W[n * w_sn + k * w_sk]
    const float* bias;
    void* y; long long y_pitch; int N;
    float* part;                                   // rowdot with several slices: [nslice][M][N]
    int nslice, slice_k;                           // reduction-axis slices and their width (multiple of 512)
};

// ---- rowdot: N <= 32.  wave -> (slice, row group); lane owns k = slice0 + step * 512 + lane * 8 .. + 7 -------------------------------------------
// weight slice in registers: NN * VEC * STEPS floats per lane -- (10, 8, 2) = 160, (16, 8, 1) = 128, (32, 4, 1) = 128; slice width 64 * VEC * STEPS
template <typename TX, typename TY, int NN, int VEC, int STEPS>
__global__ void __launch_bounds__(256) ls_rowdot_kernel(const LsArgs a) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long groups = ((long long)gridDim.x * 4) / a.nslice;              // row groups per slice (>= 1 by construction)
    const int slice = (int)(wave / groups);
    const long long g = wave % groups;
    if (slice >= a.nslice) return;
    const int k0 = slice * a.slice_k;
    constexpr int STEPW = 64 * VEC;
    float w[NN][STEPS][VEC];
#pragma unroll
    for (int n = 0; n < NN; ++n)
#pragma unroll
        for (int s = 0; s < STEPS; ++s)
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const int k = k0 + s * STEPW + lane * VEC + e;
                w[n][s][e] = (n < a.N && k < a.K) ? a.W[n * a.w_sn + k * a.w_sk] : 0.f;
            }
    const TX* x = reinterpret_cast<const TX*>(a.x);
    for (long long m = g; m < a.M; m += groups) {
        float acc[NN];
#pragma unroll
        for (int n = 0; n < NN; ++n) acc[n] = 0.f;
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            float xv[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const int k = k0 + s * STEPW + lane * VEC + e;
                xv[e] = k < a.K ? ls_ld<TX>(x, m * a.x_pitch + k) : 0.f;
            }
#pragma unroll
            for (int n = 0; n < NN; ++n)
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[n] = __builtin_fmaf(xv[e], w[n][s][e], acc[n]);
        }
#pragma unroll
        for (int n = 0; n < NN; ++n) acc[n] = wave_sum(acc[n]);
        if (lane == 0) {
            if (a.nslice == 1) {
                TY* y = reinterpret_cast<TY*>(a.y);
#pragma unroll
                for (int n = 0; n < NN; ++n) if (n < a.N) ls_st<TY>(y, m * a.y_pitch + n, acc[n] + (a.bias ? a.bias[n] : 0.f));
            } else {
#pragma unroll
                for (int n = 0; n < NN; ++n) if (n < a.N) a.part[((long long)slice * a.M + m) * a.N + n] = acc[n];
            }
        }
    }
}
// y[m][n] = bias[n] + sum over slices (fixed order)
template <typename TY>
__global__ void __launch_bounds__(256) ls_rowdot_sum_kernel(const float* __restrict__ part, int nslice, long long M, int N, const float* __restrict__ bias,
                                                            TY* __restrict__ y, long long y_pitch) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= M * N) return;
    const long long m = i / N; const int n = (int)(i % N);
    float v = bias ? bias[n] : 0.f;
    for (int s = 0; s < nslice; ++s) v += part[(long long)s * M * N + i];
    ls_st<TY>(y, m * y_pitch + n, v);
}

// ---- expand: K <= 32.  thread -> one output column (its weight row in registers), rows walked by the block, four per trip through LDS ---------
template <typename TX, typename TY, int KK>
__global__ void __launch_bounds__(256) ls_expand_kernel(const LsArgs a) {
    __shared__ float xs[4][LS_MAXS];
    const int n = blockIdx.y * 256 + threadIdx.x;
    float w[KK];
#pragma unroll
    for (int k = 0; k < KK; ++k) w[k] = (n < a.N && k < a.K) ? a.W[n * a.w_sn + k * a.w_sk] : 0.f;
    const float b = (a.bias && n < a.N) ? a.bias[n] : 0.f;
    const TX* x = reinterpret_cast<const TX*>(a.x);
    TY* y = reinterpret_cast<TY*>(a.y);
    // four rows per trip through LDS (one barrier per four rows)
    for (long long m0 = (long long)blockIdx.x * 4; m0 < a.M; m0 += (long long)gridDim.x * 4) {
        __syncthreads();
        if (threadIdx.x < 4 * LS_MAXS) {
            const int r = threadIdx.x / LS_MAXS, k = threadIdx.x % LS_MAXS;
            xs[r][k] = (m0 + r < a.M && k < a.K) ? ls_ld<TX>(x, (m0 + r) * a.x_pitch + k) : 0.f;
        }
        __syncthreads();
        if (n < a.N) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (m0 + r >= a.M) break;
                float acc = b;
#pragma unroll
                for (int k = 0; k < KK; ++k) acc = __builtin_fmaf(xs[r][k], w[k], acc);
                ls_st<TY>(y, (m0 + r) * a.y_pitch + n, acc);
            }
        }
    }
}

// ---- wgrad: thread owns index L of the long axis, SS accumulators over the short axis; rows split over gridDim.y --------------------------------
struct LsWgradArgs {
    const void* dy; long long dy_pitch; const void* x; long long x_pitch;
    long long M; int N, K;
    float* part;            // [nsplit][N * K (+ N for the bias sums)]
    int nsplit; long long rows_per_split;
    int want_bias;
};
template <typename TD, typename TX, int SS, bool NLONG>      // NLONG: N is the long axis (thread -> n, accumulators over k); else thread -> k, accumulators over n
__global__ void __launch_bounds__(256) ls_wgrad_kernel(const LsWgradArgs a) {
    __shared__ float sh[4][LS_MAXS];
    const int L = blockIdx.x * 256 + threadIdx.x;
    const int nlong = NLONG ? a.N : a.K, nshort = NLONG ? a.K : a.N;
    const TD* dy = reinterpret_cast<const TD*>(a.dy);
    const TX* x = reinterpret_cast<const TX*>(a.x);
    float acc[SS];
#pragma unroll
    for (int s = 0; s < SS; ++s) acc[s] = 0.f;
    float bacc = 0.f;
    const long long m_lo = (long long)blockIdx.y * a.rows_per_split, m_hi = min(a.M, m_lo + a.rows_per_split);
    for (long long m0 = m_lo; m0 < m_hi; m0 += 4) {
        __syncthreads();
        if (threadIdx.x < 4 * LS_MAXS) {
            const int r = threadIdx.x / LS_MAXS, s = threadIdx.x % LS_MAXS;
            float v = 0.f;
            if (m0 + r < m_hi && s < nshort) v = NLONG ? ls_ld<TX>(x, (m0 + r) * a.x_pitch + s) : ls_ld<TD>(dy, (m0 + r) * a.dy_pitch + s);
            sh[r][s] = v;
        }
        __syncthreads();
        if (L < nlong) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (m0 + r >= m_hi) break;
                const float lv = NLONG ? ls_ld<TD>(dy, (m0 + r) * a.dy_pitch + L) : ls_ld<TX>(x, (m0 + r) * a.x_pitch + L);
#pragma unroll
                for (int s = 0; s < SS; ++s) acc[s] = __builtin_fmaf(lv, sh[r][s], acc[s]);
                if (NLONG) bacc += lv;
            }
        }
        if (!NLONG && a.want_bias && blockIdx.x == 0 && (int)threadIdx.x < a.N) {     // bias sums of the short axis: thread s of the first long-axis block
#pragma unroll
            for (int r = 0; r < 4; ++r) if (m0 + r < m_hi) bacc += sh[r][threadIdx.x];
        }
    }
    float* part = a.part + (long long)blockIdx.y * ((long long)a.N * a.K + a.N);
    if (L < nlong) {
#pragma unroll
        for (int s = 0; s < SS; ++s)
            if (s < nshort) part[NLONG ? (long long)L * a.K + s : (long long)s * a.K + L] = acc[s];        // [n][k]
        if (NLONG && a.want_bias) part[(long long)a.N * a.K + L] = bacc;
    }
    if (!NLONG && a.want_bias && blockIdx.x == 0 && (int)threadIdx.x < a.N) part[(long long)a.N * a.K + threadIdx.x] = bacc;
}
// dW[n * sn + k * sk] += sum over splits (fixed order); dbias[n] += ...
__global__ void __launch_bounds__(256) ls_wgrad_sum_kernel(const float* __restrict__ part, int nsplit, int N, int K, float* __restrict__ dW, long long sn, long long sk,
                                                           float* __restrict__ dbias) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long per = (long long)N * K + N;
    if (i >= (long long)N * K + (dbias ? N : 0)) return;
    float v = 0.f;
    for (int s = 0; s < nsplit; ++s) v += part[(long long)s * per + i];
    if (i < (long long)N * K) dW[(i / K) * sn + (i % K) * sk] += v;
    else dbias[i - (long long)N * K] += v;
}

int ls_slice_k(int N) { return N <= 10 ? 1024 : (N <= 16 ? 512 : 256); }      // rowdot: width of a wave's reduction-axis slice (see the kernel's register budget)
int ls_slices(int K, int N) { return (K + ls_slice_k(N) - 1) / ls_slice_k(N); }

template <typename TX, typename TY>
int ls_fwd_typed(const LsArgs& a0, hipStream_t s) {
    LsArgs a = a0;
    if (a.K > LS_MAXS) {
        // rowdot: out <= 32, long reduction axis
        a.nslice = ls_slices(a.K, a.N); a.slice_k = ls_slice_k(a.N);
        long long groups = a.M < 2048 ? a.M : 2048;
        if (a.nslice > 1 && groups * a.nslice > 8192) groups = 8192 / a.nslice > 0 ? 8192 / a.nslice : 1;
        if (groups < 1) groups = 1;
        const long long waves = groups * a.nslice;
        const unsigned blocks = (unsigned)((waves + 3) / 4);
        // (the kernel derives its row groups as gridDim * 4 / nslice >= groups: every (slice, group) pair has a wave)
#define LS_ROWDOT(NNv, VECv, STEPSv) ls_rowdot_kernel<TX, TY, NNv, VECv, STEPSv><<<blocks, 256, 0, s>>>(a)
        if (a.N <= 4) LS_ROWDOT(4, 8, 2); else if (a.N <= 8) LS_ROWDOT(8, 8, 2); else if (a.N <= 10) LS_ROWDOT(10, 8, 2); else if (a.N <= 16) LS_ROWDOT(16, 8, 1); else LS_ROWDOT(32, 4, 1);
#undef LS_ROWDOT
        if (a.nslice > 1) {
            const long long tot = a.M * a.N;
            ls_rowdot_sum_kernel<TY><<<(unsigned)((tot + 255) / 256), 256, 0, s>>>(a.part, a.nslice, a.M, a.N, a.bias, reinterpret_cast<TY*>(a.y), a.y_pitch);
        }
        return 0;
    }
    // expand: in <= 32 (any out)
    const unsigned gy = (unsigned)((a.N + 255) / 256);
    long long gx = (a.M + 3) / 4;
    if (gx > 4096) gx = 4096;
#define LS_EXPAND(KKv) ls_expand_kernel<TX, TY, KKv><<<dim3((unsigned)gx, gy), 256, 0, s>>>(a)
    if (a.K <= 8) LS_EXPAND(8); else if (a.K <= 16) LS_EXPAND(16); else LS_EXPAND(32);
#undef LS_EXPAND
    return 0;
}

long long ls_wgrad_splits(long long M, int nlong) {
    // enough workgroups to fill the chip, at least 256 rows per split
    const long long cols = (nlong + 255) / 256;
    long long want = (1024 + cols - 1) / cols;
    const long long cap = (M + 255) / 256;
    if (want > cap) want = cap;
    if (want < 1) want = 1;
    if (want > 512) want = 512;
    return want;
}

template <typename TD, typename TX>
int ls_wgrad_typed(LsWgradArgs a, float* dW, long long sn, long long sk, float* dbias, hipStream_t s) {
    const bool nlong = a.K <= LS_MAXS && (a.N > LS_MAXS || a.N >= a.K);
    const int nl = nlong ? a.N : a.K, ns = nlong ? a.K : a.N;
    const dim3 grid((unsigned)((nl + 255) / 256), (unsigned)a.nsplit);
#define LS_WG(SSv)                                                                       \
    do {                                                                                 \
        if (nlong) ls_wgrad_kernel<TD, TX, SSv, true><<<grid, 256, 0, s>>>(a);           \
        else ls_wgrad_kernel<TD, TX, SSv, false><<<grid, 256, 0, s>>>(a);                \
    } while (0)
    if (ns <= 8) LS_WG(8); else if (ns <= 16) LS_WG(16); else LS_WG(32);
#undef LS_WG
    const long long tot = (long long)a.N * a.K + (dbias ? a.N : 0);
    ls_wgrad_sum_kernel<<<(unsigned)((tot + 255) / 256), 256, 0, s>>>(a.part, a.nsplit, a.N, a.K, dW, sn, sk, dbias);
    return 0;
}

}  // namespace

extern "C" int64_t genie_linear_small_ws_floats(int64_t M, int K, int N) {
    if (K <= LS_MAXS || N > LS_MAXS) return 0;
    const int ns = ls_slices(K, N);
    return ns > 1 ? (int64_t)ns * M * N : 0;
}

extern "C" int genie_linear_small_fwd(const void* x, int x_dtype, int64_t x_pitch, int64_t M, int K, const float* W, int64_t w_sn, int64_t w_sk,
                                      const float* bias, void* y, int y_dtype, int64_t y_pitch, int N, float* ws, int64_t ws_floats, void* stream) {
    GENIE_CHECK_ARG(x && W && y, "genie_linear_small_fwd: null pointer");
    GENIE_CHECK_ARG(M >= 0 && K >= 1 && N >= 1 && (K <= LS_MAXS || N <= LS_MAXS), "genie_linear_small_fwd: min(in %d, out %d) must be <= 32", K, N);
    GENIE_CHECK_ARG(x_pitch >= K && y_pitch >= N, "genie_linear_small_fwd: pitch");
    GENIE_CHECK_ARG((x_dtype == GENIE_F32 || x_dtype == GENIE_BF16) && (y_dtype == GENIE_F32 || y_dtype == GENIE_BF16), "genie_linear_small_fwd: dtypes (fp32 / bf16)");
    const int64_t need = genie_linear_small_ws_floats(M, K, N);
    GENIE_CHECK_ARG(need == 0 || (ws && ws_floats >= need), "genie_linear_small_fwd: workspace of %lld floats needed (genie_linear_small_ws_floats)", (long long)need);
    if (M == 0) return GENIE_OK;
    LsArgs a;
    a.x = x; a.x_pitch = x_pitch; a.M = M; a.K = K; a.W = W; a.w_sn = w_sn; a.w_sk = w_sk; a.bias = bias; a.y = y; a.y_pitch = y_pitch; a.N = N;
    a.part = ws; a.nslice = 1; a.slice_k = 0;
    hipStream_t s = (hipStream_t)stream;
    if (x_dtype == GENIE_F32 && y_dtype == GENIE_F32) ls_fwd_typed<float, float>(a, s);
    else if (x_dtype == GENIE_F32) ls_fwd_typed<float, bf16_t>(a, s);
    else if (y_dtype == GENIE_F32) ls_fwd_typed<bf16_t, float>(a, s);
    else ls_fwd_typed<bf16_t, bf16_t>(a, s);
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}

extern "C" int64_t genie_linear_small_wgrad_ws_floats(int64_t M, int N, int K) {
    const bool nlong = K <= LS_MAXS && (N > LS_MAXS || N >= K);
    return ls_wgrad_splits(M, nlong ? N : K) * ((int64_t)N * K + N);
}

extern "C" int genie_linear_small_wgrad(const void* dy, int dy_dtype, int64_t dy_pitch, const void* x, int x_dtype, int64_t x_pitch, int64_t M, int N, int K,
                                        float* dW, int64_t w_sn, int64_t w_sk, float* dbias, float* ws, int64_t ws_floats, void* stream) {
    GENIE_CHECK_ARG(dy && x && dW && ws, "genie_linear_small_wgrad: null pointer");
    GENIE_CHECK_ARG(M >= 0 && K >= 1 && N >= 1 && (K <= LS_MAXS || N <= LS_MAXS), "genie_linear_small_wgrad: min(in %d, out %d) must be <= 32", K, N);
    GENIE_CHECK_ARG(dy_pitch >= N && x_pitch >= K, "genie_linear_small_wgrad: pitch");
    GENIE_CHECK_ARG((dy_dtype == GENIE_F32 || dy_dtype == GENIE_BF16) && (x_dtype == GENIE_F32 || x_dtype == GENIE_BF16), "genie_linear_small_wgrad: dtypes (fp32 / bf16)");
    GENIE_CHECK_ARG(ws_floats >= genie_linear_small_wgrad_ws_floats(M, N, K), "genie_linear_small_wgrad: workspace too small (genie_linear_small_wgrad_ws_floats)");
    if (M == 0) return GENIE_OK;
    const bool nlong = K <= LS_MAXS && (N > LS_MAXS || N >= K);
    LsWgradArgs a;
    a.dy = dy; a.dy_pitch = dy_pitch; a.x = x; a.x_pitch = x_pitch; a.M = M; a.N = N; a.K = K; a.part = ws;
    a.nsplit = (int)ls_wgrad_splits(M, nlong ? N : K);
    a.rows_per_split = ((M + a.nsplit - 1) / a.nsplit + 3) / 4 * 4;
    a.nsplit = (int)((M + a.rows_per_split - 1) / a.rows_per_split);
    a.want_bias = dbias ? 1 : 0;
    hipStream_t s = (hipStream_t)stream;
    if (dy_dtype == GENIE_F32 && x_dtype == GENIE_F32) ls_wgrad_typed<float, float>(a, dW, w_sn, w_sk, dbias, s);
    else if (dy_dtype == GENIE_F32) ls_wgrad_typed<float, bf16_t>(a, dW, w_sn, w_sk, dbias, s);
    else if (x_dtype == GENIE_F32) ls_wgrad_typed<bf16_t, float>(a, dW, w_sn, w_sk, dbias, s);
    else ls_wgrad_typed<bf16_t, bf16_t>(a, dW, w_sn, w_sk, dbias, s);
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}
