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
#include <stdlib.h>

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

// ================================================================================================
// LFQ training loss, forward + gradient in one sweep.
//
//   p[code] = softmax_code(2 beta z.c)  = A[hi(code)] * B[lo(code)]      (per token, per codebook)
//   inp_ent = mean_{tok,cb} -sum_e p_e log(max(p_e, eps))               e runs over rep * 2^d entries,
//   avg_ent = mean_cb       -sum_e P_e log(max(P_e, eps)),  P = mean_tok p   p_e = p[code(e)] / rep
//   commit  = mean (z - sign z)^2
//   loss    = (inp_ent + div * avg_ent) * w_e + commit * w_c            (quantization.py:116-131)
//
// `rep` reproduces a reference quirk: its codebook buffer has 2^d * num_codebook rows (quantization.py:
// 53,74-75) whose low d bits repeat, so every distinct code appears num_codebook times in the softmax.
//
// Kernels: K1 factors (A, B per token) -> K2 P = A^T B / N (tiled fp32) -> K3 G = dH/dP and H(P) partials
//          -> K4 per-token sweep over all 2^d codes: H(p), and d loss / d z via
//             dF/dz_i = 2 beta * sum_c p_c (g_c - m) c_i,  m = sum_c p_c g_c,  sum_c p_c c_i = 2 s_i - 1
//          -> K5 deterministic final reduction.
// ================================================================================================
#define LFQ_EPS 1e-6f

struct LfqGeom {
    long long ntok;
    int ncb, d, dh, dl, na, nb;   // na = 2^dh, nb = 2^dl
    long long pitch;
    long long nrow;               // ntok * ncb
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

// s[i] = P(bit i = 1) = sigmoid(4 beta z_i); factor tables for the hi (first dh dims) and lo parts
template <typename T>
__device__ __forceinline__ void lfq_tables(const T* zp, const LfqGeom& g, float beta, float* s_pos, float* s_neg, float* A, float* B) {
    const int tid = threadIdx.x;
    if (tid < g.d) {
        const float x = 4.f * beta * ld<T>(zp + tid);
        s_pos[tid] = sigmoidf_(x);
        s_neg[tid] = sigmoidf_(-x);
    }
    __syncthreads();
    for (int a = tid; a < g.na; a += blockDim.x) {
        float v = 1.f;
        for (int i = 0; i < g.dh; ++i) v *= ((a >> (g.dh - 1 - i)) & 1) ? s_pos[i] : s_neg[i];
        A[a] = v;
    }
    for (int b = tid; b < g.nb; b += blockDim.x) {
        float v = 1.f;
        for (int i = 0; i < g.dl; ++i) v *= ((b >> (g.dl - 1 - i)) & 1) ? s_pos[g.dh + i] : s_neg[g.dh + i];
        B[b] = v;
    }
    __syncthreads();
}

template <typename T>
__global__ void __launch_bounds__(256) lfq_factor_kernel(const T* __restrict__ z, LfqGeom g, float beta, float* __restrict__ Aout,
                                                         float* __restrict__ Bout) {
    __shared__ float s_pos[32], s_neg[32], A[1024], B[1024];
    const long long row = blockIdx.x;
    const long long tok = row / g.ncb;
    const int cb = (int)(row % g.ncb);
    lfq_tables<T>(z + tok * g.pitch + (long long)cb * g.d, g, beta, s_pos, s_neg, A, B);
    for (int a = threadIdx.x; a < g.na; a += 256) Aout[row * g.na + a] = A[a];
    for (int b = threadIdx.x; b < g.nb; b += 256) Bout[row * g.nb + b] = B[b];
}

// P[cb][a][b] = (1/ntok) sum_tok A[tok,cb][a] * B[tok,cb][b];  64x64 tile per block, 4x4 per thread.  ksplit > 1: blockIdx.z also
// carries a token range (a 512 x 512 table is only 64 tiles: a quarter of the chip; the token split fills it) and every split writes
// its OWN partial table P[ks]; lfq_avgent_kernel adds the ksplit tables in a fixed order.  (Round 2 added the partial tiles
// atomically into one table: the order of those fp32 adds changed from run to run, the 1e-7 wobble of d loss / d z reached the
// encoder's backward pass and bf16 rounding amplified it to 4e-3 of the stem's weight gradient within a dozen layers --
// tests/test_gpu_properties.py::test_gradient_determinism.)
__global__ void __launch_bounds__(256) lfq_avgprob_kernel(const float* __restrict__ Ain, const float* __restrict__ Bin, LfqGeom g,
                                                          float* __restrict__ P, int ksplit) {
    __shared__ float As[16][64], Bs[16][64];
    const int cb = blockIdx.z / ksplit, ks = blockIdx.z % ksplit, a0 = blockIdx.y * 64, b0 = blockIdx.x * 64;
    const long long tper = ((g.ntok + ksplit - 1) / ksplit + 15) / 16 * 16;
    const long long tbeg = ks * tper, tend = tbeg + tper < g.ntok ? tbeg + tper : g.ntok;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (long long t0 = tbeg; t0 < tend; t0 += 16) {
        for (int i = threadIdx.x; i < 16 * 64; i += 256) {
            const int r = i >> 6, c = i & 63;
            const long long tok = t0 + r;
            const long long row = tok * g.ncb + cb;
            As[r][c] = (tok < tend && a0 + c < g.na) ? Ain[row * g.na + a0 + c] : 0.f;
            Bs[r][c] = (tok < tend && b0 + c < g.nb) ? Bin[row * g.nb + b0 + c] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float av[4], bv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { av[i] = As[r][ty * 4 + i]; bv[i] = Bs[r][tx * 4 + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] += av[i] * bv[j];
        }
        __syncthreads();
    }
    const float inv = 1.f / (float)g.ntok;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int a = a0 + ty * 4 + i, b = b0 + tx * 4 + j;
            if (a < g.na && b < g.nb) {
                P[(((long long)ks * g.ncb + cb) * g.na + a) * g.nb + b] = acc[i][j] * inv;
            }
        }
}

// G = d H(P) / dP;  partial sums of H(P)
__global__ void __launch_bounds__(256) lfq_avgent_kernel(const float* __restrict__ P, float* __restrict__ Gm, long long ncode, float inv_rep,
                                                         float* __restrict__ partial, int ksplit) {
    const int cb = blockIdx.y;
    const long long table = (long long)gridDim.y * ncode;          // one partial table per token split
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < ncode; i += (long long)gridDim.x * 256) {
        float p = P[cb * ncode + i];
        for (int ks = 1; ks < ksplit; ++ks) p += P[ks * table + cb * ncode + i];          // fixed order: reproducible
        const float pe = p * inv_rep;
        const float lg = __logf(fmaxf(pe, LFQ_EPS));
        acc -= p * lg;
        Gm[cb * ncode + i] = -(lg + (pe >= LFQ_EPS ? 1.f : 0.f));
    }
    __shared__ float red[4];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[cb * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// per (token, codebook): entropy and the gradient of the whole loss w.r.t. z -- any d (one code per thread and iteration, d sign-selected
// accumulators per code); the kernel below is the fast path for 2^dl >= 256
template <typename T>
__global__ void __launch_bounds__(256) lfq_token_generic_kernel(const T* __restrict__ z, LfqGeom g, float beta, float inv_rep, float div_w,
                                                        float ent_scale /* w_e / nrow */, float commit_scale /* w_c * 2 / (nrow d) */,
                                                        const float* __restrict__ Gm, float* __restrict__ Htok, float* __restrict__ Ctok,
                                                        float* __restrict__ dz /* fp32 [nrow][d] */) {
    __shared__ float s_pos[32], s_neg[32], A[1024], B[1024];
    __shared__ float red[4][24];
    const long long row = blockIdx.x;
    const long long tok = row / g.ncb;
    const int cb = (int)(row % g.ncb);
    const T* zp = z + tok * g.pitch + (long long)cb * g.d;
    lfq_tables<T>(zp, g, beta, s_pos, s_neg, A, B);
    const long long ncode = (long long)g.na * g.nb;
    const float* Gc = Gm + cb * ncode;
    float H = 0.f, m = 0.f;
    float acc[20];
#pragma unroll
    for (int i = 0; i < 20; ++i) acc[i] = 0.f;
    for (int idx = threadIdx.x; idx < ncode; idx += 256) {
        const float p = A[idx >> g.dl] * B[idx & (g.nb - 1)];
        const float pe = p * inv_rep;
        const float lg = __logf(fmaxf(pe, LFQ_EPS));
        H -= p * lg;
        const float gc = -(lg + (pe >= LFQ_EPS ? 1.f : 0.f)) + div_w * Gc[idx];
        const float t = p * gc;
        m += t;
#pragma unroll
        for (int i = 0; i < 20; ++i)
            if (i < g.d) acc[i] += ((idx >> (g.d - 1 - i)) & 1) ? t : -t;
    }
    // block reduction of H, m, acc[0..d)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    H = wave_sum(H);
    m = wave_sum(m);
    if (lane == 0) { red[wave][20] = H; red[wave][21] = m; }
#pragma unroll
    for (int i = 0; i < 20; ++i) {
        if (i < g.d) {
            const float v = wave_sum(acc[i]);
            if (lane == 0) red[wave][i] = v;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) Htok[row] = red[0][20] + red[1][20] + red[2][20] + red[3][20];
    if (threadIdx.x < g.d) {
        const int i = threadIdx.x;
        const float mm = red[0][21] + red[1][21] + red[2][21] + red[3][21];
        const float si = red[0][i] + red[1][i] + red[2][i] + red[3][i];
        const float ec = s_pos[i] - s_neg[i];                              // E[c_i] = 2 s_i - 1
        const float x = ld<T>(zp + i);
        const float q = x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f);
        dz[row * g.d + i] = 2.f * beta * ent_scale * (si - mm * ec) + commit_scale * (x - q);
        red[0][i] = (x - q) * (x - q);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float c = 0.f;
        for (int i = 0; i < g.d; ++i) c += red[0][i];
        Ctok[row] = c;
    }
}


// The same sweep when the low half has nb = 64 * BPL codes (d = 16 .. 20): a wave walks rows a = wave, wave + 4, ... of the A (x) B
// table and a lane owns BPL consecutive columns, so
//   * the sign sums over the dh HIGH bits need the row sum only: bit i of a is wave-uniform, one +- per ROW and bit instead of one per
//     code and bit;
//   * the dl LOW bits are taken from BPL per-lane column sums after the sweep;
//   * dH(P)/dP is read as whole 16-byte pieces, rows of 4 * nb contiguous bytes per wave.
// 11 VALU + one v_log per code instead of 30 + one (d = 18: 5.3 -> 2.x ms per training step of the tokenizer at 32 clips).
template <typename T, int BPL>
__global__ void __launch_bounds__(256) lfq_token_kernel(const T* __restrict__ z, LfqGeom g, float beta, float inv_rep, float div_w,
                                                        float ent_scale /* w_e / nrow */, float commit_scale /* w_c * 2 / (nrow d) */,
                                                        const float* __restrict__ Gm, float* __restrict__ Htok, float* __restrict__ Ctok,
                                                        float* __restrict__ dz /* fp32 [nrow][d] */) {
    __shared__ float s_pos[32], s_neg[32], A[1024], B[1024];
    __shared__ float red[4][24];
    const long long row = blockIdx.x;
    const long long tok = row / g.ncb;
    const int cb = (int)(row % g.ncb);
    const T* zp = z + tok * g.pitch + (long long)cb * g.d;
    lfq_tables<T>(zp, g, beta, s_pos, s_neg, A, B);
    const long long ncode = (long long)g.na * g.nb;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* Gc = Gm + cb * ncode + lane * BPL;
    float Bv[BPL], cs[BPL];
#pragma unroll
    for (int j = 0; j < BPL; ++j) { Bv[j] = B[lane * BPL + j]; cs[j] = 0.f; }
    float H = 0.f, m = 0.f;
    float acc[20];
#pragma unroll
    for (int i = 0; i < 20; ++i) acc[i] = 0.f;
    for (int a = wave; a < g.na; a += 4) {
        const float Aa = A[a];
        float G[BPL];
#pragma unroll
        for (int j = 0; j < BPL; j += 4) *reinterpret_cast<f32x4_t*>(G + j) = *reinterpret_cast<const f32x4_t*>(Gc + (long long)a * g.nb + j);
        float r = 0.f;
#pragma unroll
        for (int j = 0; j < BPL; ++j) {
            const float p = Aa * Bv[j];
            const float pe = p * inv_rep;
            const float lg = __logf(fmaxf(pe, LFQ_EPS));
            H -= p * lg;
            const float gc = -(lg + (pe >= LFQ_EPS ? 1.f : 0.f)) + div_w * G[j];
            const float t = p * gc;
            r += t;
            cs[j] += t;
        }
        m += r;
#pragma unroll
        for (int i = 0; i < 10; ++i)
            if (i < g.dh) acc[i] += ((a >> (g.dh - 1 - i)) & 1) ? r : -r;
    }
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        if (i < g.dl) {
            float v = 0.f;
#pragma unroll
            for (int j = 0; j < BPL; ++j) v += (((lane * BPL + j) >> (g.dl - 1 - i)) & 1) ? cs[j] : -cs[j];
            acc[10 + i] = v;
        }
    }
    H = wave_sum(H);
    m = wave_sum(m);
    if (lane == 0) { red[wave][20] = H; red[wave][21] = m; }
#pragma unroll
    for (int i = 0; i < 20; ++i) {
        const bool used = i < 10 ? i < g.dh : i - 10 < g.dl;
        if (used) {
            const float v = wave_sum(acc[i]);
            if (lane == 0) red[wave][i < 10 ? i : g.dh + (i - 10)] = v;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) Htok[row] = red[0][20] + red[1][20] + red[2][20] + red[3][20];
    if (threadIdx.x < g.d) {
        const int i = threadIdx.x;
        const float mm = red[0][21] + red[1][21] + red[2][21] + red[3][21];
        const float si = red[0][i] + red[1][i] + red[2][i] + red[3][i];
        const float ec = s_pos[i] - s_neg[i];                              // E[c_i] = 2 s_i - 1
        const float x = ld<T>(zp + i);
        const float q = x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f);
        dz[row * g.d + i] = 2.f * beta * ent_scale * (si - mm * ec) + commit_scale * (x - q);
        red[0][i] = (x - q) * (x - q);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float c = 0.f;
        for (int i = 0; i < g.d; ++i) c += red[0][i];
        Ctok[row] = c;
    }
}

__global__ void __launch_bounds__(256) lfq_final_kernel(const float* __restrict__ Htok, const float* __restrict__ Ctok, long long nrow,
                                                        const float* __restrict__ avg_partial, int n_avg_partial, int ncb, int d,
                                                        float commit_w, float ent_w, float div_w, float* __restrict__ out) {
    __shared__ double rh[256], rc[256];
    double h = 0.0, c = 0.0;
    for (long long i = threadIdx.x; i < nrow; i += 256) { h += (double)Htok[i]; c += (double)Ctok[i]; }
    rh[threadIdx.x] = h; rc[threadIdx.x] = c;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { rh[threadIdx.x] += rh[threadIdx.x + o]; rc[threadIdx.x] += rc[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double avg = 0.0;
        for (int i = 0; i < n_avg_partial; ++i) avg += (double)avg_partial[i];
        const double inp_ent = rh[0] / (double)nrow;
        const double avg_ent = avg / (double)ncb;
        const double commit = rc[0] / ((double)nrow * (double)d);
        out[0] = (float)((inp_ent + (double)div_w * avg_ent) * (double)ent_w + commit * (double)commit_w);
        out[1] = (float)inp_ent;
        out[2] = (float)avg_ent;
        out[3] = (float)commit;
    }
}

#define LFQ_AVG_BLOCKS 64
#define LFQ_MAX_KSPLIT 16    // token splits of the average-probability product (one partial table each)

extern "C" int64_t genie_lfq_loss_ws_floats(int64_t ntok, int num_codebook, int codebook_dim) {
    const int dh = codebook_dim / 2, dl = codebook_dim - dh;
    const int64_t nrow = ntok * num_codebook;
    return nrow * ((1ll << dh) + (1ll << dl)) + (1ll + LFQ_MAX_KSPLIT) * num_codebook * (1ll << codebook_dim) + 2 * nrow + (int64_t)num_codebook * LFQ_AVG_BLOCKS + 64;
}

extern "C" int genie_lfq_loss(const void* z, int dtype, int64_t ntok, int num_codebook, int codebook_dim, int64_t pitch, float beta,
                              float commit_weight, float entropy_weight, float diversity_weight, float* ws, float* loss4, float* dz,
                              void* stream) {
    GENIE_CHECK_ARG(z && ws && loss4 && dz, "genie_lfq_loss: null pointer");
    GENIE_CHECK_ARG(codebook_dim >= 1 && codebook_dim <= 20, "genie_lfq_loss: codebook_dim %d out of range [1, 20]", codebook_dim);
    GENIE_CHECK_ARG(num_codebook >= 1 && pitch >= (int64_t)num_codebook * codebook_dim, "genie_lfq_loss: bad pitch");
    GENIE_CHECK_ARG(ntok >= 1, "genie_lfq_loss: no tokens");
    LfqGeom g;
    g.ntok = ntok; g.ncb = num_codebook; g.d = codebook_dim; g.dh = codebook_dim / 2; g.dl = codebook_dim - g.dh;
    g.na = 1 << g.dh; g.nb = 1 << g.dl; g.pitch = pitch; g.nrow = ntok * num_codebook;
    GENIE_CHECK_ARG(g.nrow < (1ll << 31), "genie_lfq_loss: too many tokens");
    const long long ncode = 1ll << codebook_dim;
    float* A = ws;
    float* B = A + g.nrow * g.na;
    float* P = B + g.nrow * g.nb;
    float* Gm = P + (long long)LFQ_MAX_KSPLIT * num_codebook * ncode;        // P holds up to LFQ_MAX_KSPLIT partial tables
    float* Htok = Gm + (long long)num_codebook * ncode;
    float* Ctok = Htok + g.nrow;
    float* avgp = Ctok + g.nrow;
    hipStream_t s = (hipStream_t)stream;
    const float inv_rep = 1.f / (float)num_codebook;
    const float ent_scale = entropy_weight / (float)g.nrow;
    const float commit_scale = commit_weight * 2.f / ((float)g.nrow * (float)codebook_dim);
    if (dtype == GENIE_BF16)
        lfq_factor_kernel<bf16_t><<<(unsigned)g.nrow, 256, 0, s>>>((const bf16_t*)z, g, beta, A, B);
    else if (dtype == GENIE_F32)
        lfq_factor_kernel<float><<<(unsigned)g.nrow, 256, 0, s>>>((const float*)z, g, beta, A, B);
    else
        GENIE_CHECK_ARG(false, "genie_lfq_loss: unsupported dtype %d", dtype);
    GENIE_CHECK_LAUNCH();
    int ksplit = 1;
    {
        const long long tiles = (long long)cdiv(g.nb, 64) * cdiv(g.na, 64) * num_codebook;
        while (tiles * ksplit < 256 && ntok / (ksplit * 2) >= 256 && ksplit < LFQ_MAX_KSPLIT) ksplit *= 2;
        lfq_avgprob_kernel<<<dim3(cdiv(g.nb, 64), cdiv(g.na, 64), num_codebook * ksplit), 256, 0, s>>>(A, B, g, P, ksplit);
    }
    GENIE_CHECK_LAUNCH();
    int ablk = (int)((ncode + 255) / 256);
    if (ablk > LFQ_AVG_BLOCKS) ablk = LFQ_AVG_BLOCKS;
    lfq_avgent_kernel<<<dim3(ablk, num_codebook), 256, 0, s>>>(P, Gm, ncode, inv_rep, avgp, ksplit);
    GENIE_CHECK_LAUNCH();
    static const int fast_tok = getenv("GENIE_LFQ_FAST") ? atoi(getenv("GENIE_LFQ_FAST")) : 1;
    const int bpl = (fast_tok && g.nb % 64 == 0 && g.dh <= 10 && g.dl <= 10) ? g.nb / 64 : 0;     // columns per lane of the fast sweep: 4, 8 or 16
#define LFQ_TOKEN_ARGS g, beta, inv_rep, diversity_weight, ent_scale, commit_scale, Gm, Htok, Ctok, dz
    if (dtype == GENIE_BF16) {
        if (bpl == 4) lfq_token_kernel<bf16_t, 4><<<(unsigned)g.nrow, 256, 0, s>>>((const bf16_t*)z, LFQ_TOKEN_ARGS);
        else if (bpl == 8) lfq_token_kernel<bf16_t, 8><<<(unsigned)g.nrow, 256, 0, s>>>((const bf16_t*)z, LFQ_TOKEN_ARGS);
        else if (bpl == 16) lfq_token_kernel<bf16_t, 16><<<(unsigned)g.nrow, 256, 0, s>>>((const bf16_t*)z, LFQ_TOKEN_ARGS);
        else lfq_token_generic_kernel<bf16_t><<<(unsigned)g.nrow, 256, 0, s>>>((const bf16_t*)z, LFQ_TOKEN_ARGS);
    } else {
        if (bpl == 4) lfq_token_kernel<float, 4><<<(unsigned)g.nrow, 256, 0, s>>>((const float*)z, LFQ_TOKEN_ARGS);
        else if (bpl == 8) lfq_token_kernel<float, 8><<<(unsigned)g.nrow, 256, 0, s>>>((const float*)z, LFQ_TOKEN_ARGS);
        else if (bpl == 16) lfq_token_kernel<float, 16><<<(unsigned)g.nrow, 256, 0, s>>>((const float*)z, LFQ_TOKEN_ARGS);
        else lfq_token_generic_kernel<float><<<(unsigned)g.nrow, 256, 0, s>>>((const float*)z, LFQ_TOKEN_ARGS);
    }
#undef LFQ_TOKEN_ARGS
    GENIE_CHECK_LAUNCH();
    lfq_final_kernel<<<1, 256, 0, s>>>(Htok, Ctok, g.nrow, avgp, ablk * num_codebook, num_codebook, codebook_dim, commit_weight, entropy_weight, diversity_weight, loss4);
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}

// dz_total = dy (straight-through) + gscale * dz_loss, written in the layout/dtype of z
template <typename T>
__global__ void __launch_bounds__(256) lfq_bwd_kernel(const T* __restrict__ dy, const float* __restrict__ dzl, const float* __restrict__ gscale,
                                                      T* __restrict__ out, long long ntok, int width, long long pitch) {
    const float gs = gscale ? *gscale : 0.f;
    const long long total = ntok * pitch;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long tok = i / pitch;
        const int c = (int)(i % pitch);
        float v = 0.f;
        if (c < width) v = (dy ? ld<T>(dy + i) : 0.f) + (dzl ? gs * dzl[tok * width + c] : 0.f);
        st<T>(out + i, v);
    }
}

extern "C" int genie_lfq_bwd(const void* dy, const float* dz_loss, const float* grad_loss, void* out, int dtype, int64_t ntok, int width,
                             int64_t pitch, void* stream) {
    GENIE_CHECK_ARG(out, "genie_lfq_bwd: null pointer");
    if (ntok == 0) return GENIE_OK;
    const long long total = ntok * pitch;
    int grid = (int)((total + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GENIE_BF16)
        lfq_bwd_kernel<bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)dy, dz_loss, grad_loss, (bf16_t*)out, ntok, width, pitch);
    else if (dtype == GENIE_F32)
        lfq_bwd_kernel<float><<<grid, 256, 0, s>>>((const float*)dy, dz_loss, grad_loss, (float*)out, ntok, width, pitch);
    else
        GENIE_CHECK_ARG(false, "genie_lfq_bwd: unsupported dtype %d", dtype);
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}

// ------------------------------------------------------------------------------------------------
// Masked token cross-entropy over bf16 logits rows (DynamicsModel.compute_loss, reference genie/dynamics.py:66-99: boolean-mask
// gather of the logits, F.cross_entropy(mean)).  The reference path materialises the gathered rows, an fp32 copy, the softmax and
// its backward and scatters the gradient back -- eight passes over up to 2 GiB at V = 2^18.  Here: forward = ONE read pass
// (per-row log-sum-exp + the loss), backward = one read + one write pass producing d logits (zero rows where the mask is off).
// One 256-thread block per row, grid-stride; 16-B loads.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float ce_block_max(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__device__ __forceinline__ float ce_block_sum(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__global__ void __launch_bounds__(256) masked_ce_fwd_kernel(const bf16_t* __restrict__ logits, long long pitch, long long nrow, int V,
                                                            const long long* __restrict__ target, const unsigned char* __restrict__ mask,
                                                            float* __restrict__ row_lse, float* __restrict__ loss_sum) {
    __shared__ float red[4];
    const int nch = (V + 7) >> 3;                                        // a ragged last chunk reads pad columns (pitch >= 8 * nch) and masks them
    for (long long r = blockIdx.x; r < nrow; r += gridDim.x) {
        if (mask && !mask[r]) continue;                                  // block-uniform
        const bf16_t* row = logits + r * pitch;
        float m = -INFINITY, s = 0.f;                                    // online max / sum of exp per thread
        for (int ch = threadIdx.x; ch < nch; ch += 256) {
            float f[8];
            unpack8(*reinterpret_cast<const u32x4_t*>(row + ch * 8), f);
            if (ch * 8 + 8 > V) {
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] = (ch * 8 + j < V) ? f[j] : -INFINITY;
            }
            float cm = f[0];
#pragma unroll
            for (int j = 1; j < 8; ++j) cm = fmaxf(cm, f[j]);
            if (cm > m) { s *= __expf(m - cm); m = cm; }
#pragma unroll
            for (int j = 0; j < 8; ++j) s += __expf(f[j] - m);
        }
        const float M = ce_block_max(m, red);
        const float S = ce_block_sum(s * __expf(m - M), red);
        if (threadIdx.x == 0) {
            const float lse = M + __logf(S);
            row_lse[r] = lse;
        }
    }
}

// sum over the masked rows of lse - logit[target], in ONE fixed order (thread t takes rows t, t + 1024, ...; fixed tree over the threads; a single
// writer): the loss is bit-reproducible run to run -- per-row atomics on the one loss word were neither that nor cheap (L2 serialises
// same-address atomics).  One block; 15 bytes per row.
__global__ void __launch_bounds__(1024) masked_ce_loss_kernel(const bf16_t* __restrict__ logits, long long pitch, long long nrow, int V,
                                                              const long long* __restrict__ target, const unsigned char* __restrict__ mask,
                                                              const float* __restrict__ row_lse, float* __restrict__ loss_sum) {
    __shared__ float red[1024];
    float acc = 0.f;
    for (long long r = threadIdx.x; r < nrow; r += 1024) {
        if (mask && !mask[r]) continue;
        const long long t = target[r];
        // F.cross_entropy raises on a target outside [0, V); a kernel cannot, so the loss is poisoned instead of reading out of bounds
        acc += (t >= 0 && t < V) ? row_lse[r] - bf16_to_f32(logits[r * pitch + t]) : __builtin_nanf("");
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss_sum += red[0];
}

__global__ void __launch_bounds__(256) masked_ce_bwd_kernel(const bf16_t* __restrict__ logits, long long pitch, long long nrow, int V,
                                                            const long long* __restrict__ target, const unsigned char* __restrict__ mask,
                                                            const float* __restrict__ row_lse, const float* __restrict__ scale,
                                                            bf16_t* __restrict__ dlogits, long long dpitch) {
    const int nch = (V + 7) >> 3;
    const float sc = scale[0];
    for (long long r = blockIdx.x; r < nrow; r += gridDim.x) {
        bf16_t* drow = dlogits + r * dpitch;
        if (mask && !mask[r]) {
            const u32x4_t z = {0u, 0u, 0u, 0u};
            for (int ch = threadIdx.x; ch < nch; ch += 256) *reinterpret_cast<u32x4_t*>(drow + ch * 8) = z;
            continue;
        }
        const bf16_t* row = logits + r * pitch;
        const float lse = row_lse[r];
        const int tgt = (int)target[r];
        for (int ch = threadIdx.x; ch < nch; ch += 256) {
            float f[8];
            unpack8(*reinterpret_cast<const u32x4_t*>(row + ch * 8), f);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = (ch * 8 + j < V) ? (__expf(f[j] - lse) - ((ch * 8 + j) == tgt ? 1.f : 0.f)) * sc : 0.f;      // pad columns stay zero
            *reinterpret_cast<u32x4_t*>(drow + ch * 8) = pack8(f);
        }
    }
}

extern "C" int genie_masked_ce_fwd(const void* logits_bf16, int64_t pitch, int64_t nrow, int V, const int64_t* target,
                                   const unsigned char* mask, float* row_lse, float* loss_sum, void* stream) {
    GENIE_CHECK_ARG(logits_bf16 && target && row_lse && loss_sum, "genie_masked_ce_fwd: null pointer");
    GENIE_CHECK_ARG(V >= 1 && pitch >= ((V + 7) & ~7) && pitch % 8 == 0, "genie_masked_ce_fwd: V=%d needs a row pitch that is a multiple of 8 and >= V rounded up to 8 (got %lld)", V, (long long)pitch);
    if (nrow == 0) return GENIE_OK;
    const unsigned grid = (unsigned)(nrow < 4096 ? nrow : 4096);
    masked_ce_fwd_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const bf16_t*)logits_bf16, pitch, nrow, V, (const long long*)target, mask, row_lse, loss_sum);
    masked_ce_loss_kernel<<<1, 1024, 0, (hipStream_t)stream>>>((const bf16_t*)logits_bf16, pitch, nrow, V, (const long long*)target, mask, row_lse, loss_sum);
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}

extern "C" int genie_masked_ce_bwd(const void* logits_bf16, int64_t pitch, int64_t nrow, int V, const int64_t* target,
                                   const unsigned char* mask, const float* row_lse, const float* scale, void* dlogits_bf16,
                                   int64_t dpitch, void* stream) {
    GENIE_CHECK_ARG(logits_bf16 && target && row_lse && scale && dlogits_bf16, "genie_masked_ce_bwd: null pointer");
    GENIE_CHECK_ARG(V >= 1 && pitch >= ((V + 7) & ~7) && dpitch >= ((V + 7) & ~7) && pitch % 8 == 0 && dpitch % 8 == 0, "genie_masked_ce_bwd: bad V / pitch");
    if (nrow == 0) return GENIE_OK;
    const unsigned grid = (unsigned)(nrow < 4096 ? nrow : 4096);
    masked_ce_bwd_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const bf16_t*)logits_bf16, pitch, nrow, V, (const long long*)target, mask, row_lse, scale, (bf16_t*)dlogits_bf16, dpitch);
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}
