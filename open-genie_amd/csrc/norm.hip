// GroupNorm (+ adaptive scale/shift, + SiLU) on CL bf16 tensors -- HBM-bound streaming kernels.
//
// Forward : stats pass (1 read) -> finalize (tiny) -> apply pass (1 read + 1 write)
// Backward: reduce pass (reads x, dy) -> finalize (tiny) -> apply pass (reads x, dy; writes dx)
//
// Thread mapping shared by all passes: a block owns (sample n, a contiguous pixel range); inside it
// thread -> (pixel row pr, 16-B channel chunk cc) with cc FIXED for the thread's lifetime, so the
// per-channel coefficients live in registers and per-channel partial sums need no atomics.  All
// reductions run in a fixed order (deterministic).  Statistics are fp32 per thread/block and fp64
// across blocks.
//
// Reference semantics: torch.nn.GroupNorm / F.group_norm (biased variance, eps inside the sqrt) as used
// at genie/module/video.py:578,612, genie/module/norm.py:58, genie/module/misc.py:92.
#include "common.h"
#include "genie_hip.h"
#include <stdlib.h>

#define GENIE_GN_CHUNK_DEFAULT_MB 0ll      // sample chunking for memory-side-cache reuse: off unless measured faster (GENIE_GN_CHUNK_MB)

#define GN_MAX_BLK 128

struct GnGeom {
    int N, C, Cp, G, CH;   // CH = Cp / 8
    long long npix;
    int nblk;              // blocks per sample
    long long pix_per_blk;
};

// Sweep order of the streaming passes (K = 16-byte accesses a thread issues together per stream):
//   legacy (GENIE_GN_SWEEP=0; rounds 1-3): a block owns ONE contiguous pixel range of its sample and walks it two accesses deep -- all four
//       passes sat at 5.0-5.4 TB/s, which matched the runtime's device copy and was taken for the ceiling;
//   default: the sample is cut into chunks of K block-rows (K x 4 KB when the 256 threads cover whole pixels) and read with non-temporal loads.
//       The passes that keep per-block partial sums (statistics K = 8, backward reduce K = 4) give block b the chunks b, b + nblk, ...: the
//       blocks of a sample read one dense window instead of nblk windows 0.5 MB apart.  The passes that write (apply, backward apply; K = 4)
//       run ONE chunk per block -- many short blocks dispatched in address order keep the read and the write stream dense, which persistent
//       blocks do not (scripts/probes/hbm_stream*.hip: a copy reaches 6.2 TB/s that way, 5.3 with a contiguous range per block).
//   Measured per pass on a 1.07-GB tensor (64 x 128ch x 16x64x64, rocprofv3): statistics 211 -> 162 us (6.6 TB/s), apply 390 -> 358
//   (6.0), backward reduce 383-428 -> 333 (6.4), backward apply 587 -> 533 us (6.0).
#define GENIE_GN_SWEEP_DEFAULT 1
static int gn_sweep() {
    static const int k = getenv("GENIE_GN_SWEEP") ? atoi(getenv("GENIE_GN_SWEEP")) : GENIE_GN_SWEEP_DEFAULT;
    return k != 0;
}
#define GN_K_STATS 8
#define GN_K_PASS 4
// geometry of the passes without per-block partial sums: one chunk per block
static GnGeom gn_geom_apply(GnGeom g) {
    if (gn_sweep()) {
        const int chb = g.CH < 256 ? g.CH : 256;
        const long long chunk_pix = (long long)GN_K_PASS * (256 / chb);
        const long long nb = (g.npix + chunk_pix - 1) / chunk_pix;
        if (nb <= 0x7fffffffll) g.nblk = (int)nb;
    }
    return g;
}
// activation selected at COMPILE time: with `act` as a run-time argument hipcc turned `act == 1 ? silu(z) : ...` into one scalar branch per
// ELEMENT (32 three-instruction basic blocks per access, each a serial exp -> rcp chain behind s_nop hazards), and the passes ran at the
// issue rate of that code, not at the rate of the memory system
template <int ACT> __device__ __forceinline__ float gn_act(float z) { return ACT == 1 ? silu_f(z) : (ACT == 2 ? (z > 0.f ? z : 0.01f * z) : z); }
template <int ACT> __device__ __forceinline__ float gn_act_bwd(float z, float d) {
    return ACT == 1 ? d * silu_grad_f(z) : (ACT == 2 ? (z > 0.f ? d : 0.01f * d) : d);
}
__device__ __forceinline__ u32x4_t gn_ld_stream(const bf16_t* p) { return __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p)); }

static GnGeom gn_geom(int N, long long npix, int C, int Cp, int G) {
    GnGeom g;
    g.N = N; g.C = C; g.Cp = Cp; g.G = G; g.CH = Cp / 8; g.npix = npix;
    static const long long gn_blocks = getenv("GENIE_GN_BLOCKS") ? atoll(getenv("GENIE_GN_BLOCKS")) : 2048;
    long long want = (gn_blocks + N - 1) / N;            // ~2048 blocks in flight over the whole batch
    long long by_size = (npix * g.CH + 2047) / 2048;     // at least ~8 x 16 B per thread
    long long nb = want < by_size ? want : by_size;
    if (nb > GN_MAX_BLK) nb = GN_MAX_BLK;
    if (nb < 1) nb = 1;
    g.pix_per_blk = (npix + nb - 1) / nb;
    g.nblk = (int)((npix + g.pix_per_blk - 1) / g.pix_per_blk);
    return g;
}

extern "C" int64_t genie_groupnorm_ws_floats(int N, int C, int G) {
    const int Cp = (C + 7) & ~7;
    return (int64_t)N * GN_MAX_BLK * Cp * 2 + (int64_t)N * Cp * 4 + (int64_t)N * G * 4 + 64;
}

// ---- stats: per (n, blk, channel) sum and sum of squares ------------------------------------------
template <int K>
__global__ void __launch_bounds__(256) gn_stats_kernel(const bf16_t* __restrict__ x, GnGeom g, float* __restrict__ part) {
    __shared__ float red[256 * 16];
    const int n = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x;
    const long long p0 = (long long)blk * g.pix_per_blk;
    long long p1 = p0 + g.pix_per_blk;
    if (p1 > g.npix) p1 = g.npix;
    const bf16_t* xs = x + (long long)n * g.npix * g.Cp;
    for (int cc0 = 0; cc0 < g.CH; cc0 += 256) {
        const int chb = g.CH - cc0 < 256 ? g.CH - cc0 : 256;
        const int R = 256 / chb;
        const int pr = tid / chb, cc = cc0 + tid % chb;
        float s[8], q[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] = q[j] = 0.f;
        if (pr < R && K == 0) {
    #pragma unroll 2
        for (long long p = p0 + pr; p < p1; p += R) {
                float f[8];
                unpack8(*reinterpret_cast<const u32x4_t*>(xs + p * g.Cp + cc * 8), f);
#pragma unroll
                for (int j = 0; j < 8; ++j) { s[j] += f[j]; q[j] += f[j] * f[j]; }
            }
        }
        if (pr < R && K > 0) {
            constexpr int KK = K > 0 ? K : 1;
            for (long long pc = (long long)blk * (KK * R); pc < g.npix; pc += (long long)g.nblk * (KK * R)) {
                u32x4_t v[KK];
#pragma unroll
                for (int k = 0; k < KK; ++k) {
                    const long long p = pc + k * R + pr;
                    v[k] = p < g.npix ? gn_ld_stream(xs + p * g.Cp + cc * 8) : u32x4_t{0u, 0u, 0u, 0u};
                }
#pragma unroll
                for (int k = 0; k < KK; ++k) {
                    float f[8];
                    unpack8(v[k], f);
#pragma unroll
                    for (int j = 0; j < 8; ++j) { s[j] += f[j]; q[j] += f[j] * f[j]; }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { red[tid * 16 + j] = s[j]; red[tid * 16 + 8 + j] = q[j]; }
        __syncthreads();
        if (tid < chb) {
            float ts[8], tq[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) ts[j] = tq[j] = 0.f;
            for (int r = 0; r < R; ++r) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { ts[j] += red[(r * chb + tid) * 16 + j]; tq[j] += red[(r * chb + tid) * 16 + 8 + j]; }
            }
            float* o = part + (((long long)n * g.nblk + blk) * g.Cp + (cc0 + tid) * 8) * 2;
#pragma unroll
            for (int j = 0; j < 8; ++j) { o[2 * j] = ts[j]; o[2 * j + 1] = tq[j]; }
        }
        __syncthreads();
    }
}

// ---- finalize: one 1024-thread block per (n, group) sums the per-block partials of the group's channels (fixed order, fp64)
//      and writes mean / rstd.  It is a pure latency chain between the two streaming passes (92 launches per training step), so
//      the loads are spread over 16 waves and issued 8 deep; the (block, channel) cursor advances without divisions ----
#define GN_FIN_THREADS 1024
__device__ __forceinline__ double block_sum_fin(double v, double* red) {     // red: GN_FIN_THREADS / 64 doubles
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();                                    // red may still be read from a previous call
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < GN_FIN_THREADS / 64; ++w) t += red[w];
    return t;
}

__global__ void __launch_bounds__(GN_FIN_THREADS) gn_finalize_kernel(const float* __restrict__ part, GnGeom g, float eps,
                                                                     float* __restrict__ mean, float* __restrict__ rstd) {
    __shared__ double red[GN_FIN_THREADS / 64];
    const int n = blockIdx.y, grp = blockIdx.x, tid = threadIdx.x;
    const int cg = g.C / g.G;
    // the group's partials: nblk rows of cg (sum, sumsq) pairs; thread t takes pairs t, t + 1024, ... of the flattened list
    double s = 0.0, q = 0.0;
    const float* base = part + ((long long)n * g.nblk * g.Cp + (long long)grp * cg) * 2;
    const int total = g.nblk * cg;
    const int dblk = GN_FIN_THREADS / cg, dci = GN_FIN_THREADS % cg;
    int blk = tid / cg, ci = tid % cg;
#pragma unroll 8
    for (int f = tid; f < total; f += GN_FIN_THREADS) {
        const float2 v = *reinterpret_cast<const float2*>(base + ((long long)blk * g.Cp + ci) * 2);
        s += (double)v.x;
        q += (double)v.y;
        blk += dblk; ci += dci;
        if (ci >= cg) { ci -= cg; ++blk; }
    }
    s = block_sum_fin(s, red);
    q = block_sum_fin(q, red);
    if (tid == 0) {
        const double cnt = (double)cg * (double)g.npix;
        const double m = s / cnt;
        double var = q / cnt - m * m;
        if (var < 0.0) var = 0.0;
        mean[n * g.G + grp] = (float)m;
        rstd[n * g.G + grp] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

// forward coefficients of one channel: z = x * a + b
__device__ __forceinline__ void gn_coef(int n, int c, const GnGeom& g, const float* gamma, const float* beta,
                                        const float* ada_s, const float* ada_b, const float* mean, const float* rstd,
                                        float& a, float& b, float& mu, float& rs, float& gam_eff) {
    if (c >= g.C) { a = b = mu = rs = gam_eff = 0.f; return; }
    const int grp = c / (g.C / g.G);
    mu = mean[n * g.G + grp];
    rs = rstd[n * g.G + grp];
    const float ga = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
    const float as = ada_s ? ada_s[(long long)n * g.C + c] : 1.f, ab = ada_b ? ada_b[(long long)n * g.C + c] : 0.f;
    gam_eff = ga * as;
    a = rs * gam_eff;
    b = (be - mu * rs * ga) * as + ab;
}

// the same for the 8 consecutive channels [c0, c0 + 8) of one 16-byte chunk.  Common case -- all 8 inside C and inside ONE group, no
// adaptive scale / shift: one group lookup, gamma / beta as two 16-byte loads each (c0 is a multiple of 8, cudaMalloc'ed parameters are
// 16-byte aligned when c0 * 4 is) instead of 8 divisions and 32 dependent scalar loads; this is the prologue of every block of every pass
__device__ __forceinline__ void gn_coef8(int n, int c0, const GnGeom& g, const float* gamma, const float* beta, const float* ada_s,
                                         const float* ada_b, const float* mean, const float* rstd, float* a, float* b, float* mu,
                                         float* rs, float* ge) {
    const int cg = g.C / g.G;
    const bool fast = c0 + 8 <= g.C && (cg & 7) == 0 && !ada_s && !ada_b && ((reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta)) & 15) == 0;
    if (fast) {
        const int grp = c0 / cg;
        const float m = mean[n * g.G + grp], r = rstd[n * g.G + grp];
        float ga[8], be[8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float4 gv = gamma ? *reinterpret_cast<const float4*>(gamma + c0 + 4 * h) : float4{1.f, 1.f, 1.f, 1.f};
            const float4 bv = beta ? *reinterpret_cast<const float4*>(beta + c0 + 4 * h) : float4{0.f, 0.f, 0.f, 0.f};
            ga[4 * h] = gv.x; ga[4 * h + 1] = gv.y; ga[4 * h + 2] = gv.z; ga[4 * h + 3] = gv.w;
            be[4 * h] = bv.x; be[4 * h + 1] = bv.y; be[4 * h + 2] = bv.z; be[4 * h + 3] = bv.w;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            mu[j] = m; rs[j] = r; ge[j] = ga[j];
            a[j] = r * ga[j];
            b[j] = be[j] - m * r * ga[j];
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) gn_coef(n, c0 + j, g, gamma, beta, ada_s, ada_b, mean, rstd, a[j], b[j], mu[j], rs[j], ge[j]);
}

template <int K, int ACT>
__global__ void __launch_bounds__(256) gn_apply_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, GnGeom g,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ ada_s, const float* __restrict__ ada_b,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd) {
    const int n = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x;
    const long long p0 = (long long)blk * g.pix_per_blk;
    long long p1 = p0 + g.pix_per_blk;
    if (p1 > g.npix) p1 = g.npix;
    const bf16_t* xs = x + (long long)n * g.npix * g.Cp;
    bf16_t* ys = y + (long long)n * g.npix * g.Cp;
    for (int cc0 = 0; cc0 < g.CH; cc0 += 256) {
        const int chb = g.CH - cc0 < 256 ? g.CH - cc0 : 256;
        const int R = 256 / chb;
        const int pr = tid / chb, cc = cc0 + tid % chb;
        if (pr >= R) continue;
        float a[8], b[8];
        {
            float mu[8], rs[8], ge[8];
            gn_coef8(n, cc * 8, g, gamma, beta, ada_s, ada_b, mean, rstd, a, b, mu, rs, ge);
        }
        if (K == 0) {
#pragma unroll 2
        for (long long p = p0 + pr; p < p1; p += R) {
            float f[8];
            unpack8(*reinterpret_cast<const u32x4_t*>(xs + p * g.Cp + cc * 8), f);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float z = f[j] * a[j] + b[j];
                f[j] = gn_act<ACT>(z);
            }
            *reinterpret_cast<u32x4_t*>(ys + p * g.Cp + cc * 8) = pack8(f);
        }
        } else {
            constexpr int KK = K > 0 ? K : 1;
            for (long long pc = (long long)blk * (KK * R); pc < g.npix; pc += (long long)g.nblk * (KK * R)) {
                u32x4_t v[KK];
#pragma unroll
                for (int k = 0; k < KK; ++k) {
                    const long long p = pc + k * R + pr;
                    v[k] = p < g.npix ? gn_ld_stream(xs + p * g.Cp + cc * 8) : u32x4_t{0u, 0u, 0u, 0u};
                }
#pragma unroll
                for (int k = 0; k < KK; ++k) {
                    const long long p = pc + k * R + pr;
                    float f[8];
                    unpack8(v[k], f);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        float z = f[j] * a[j] + b[j];
                        f[j] = gn_act<ACT>(z);
                    }
                    if (p < g.npix) *reinterpret_cast<u32x4_t*>(ys + p * g.Cp + cc * 8) = pack8(f);
                }
            }
        }
    }
}


// launch one of the streaming passes: sweep order and activation (0 none, 1 SiLU, 2 LeakyReLU 0.01) as template arguments
#define GN_UNPAREN(...) __VA_ARGS__
#define GN_LAUNCH_K(kernel, cfg, args)                                                         \
    do {                                                                                       \
        if (gn_sweep()) kernel<GN_K_STATS><<<GN_UNPAREN cfg>>>(GN_UNPAREN args);               \
        else kernel<0><<<GN_UNPAREN cfg>>>(GN_UNPAREN args);                                   \
    } while (0)
#define GN_LAUNCH_KA_(kernel, K, act, cfg, args)                                               \
    do {                                                                                       \
        if ((act) == 1) kernel<K, 1><<<GN_UNPAREN cfg>>>(GN_UNPAREN args);                     \
        else if ((act) == 2) kernel<K, 2><<<GN_UNPAREN cfg>>>(GN_UNPAREN args);                \
        else kernel<K, 0><<<GN_UNPAREN cfg>>>(GN_UNPAREN args);                                \
    } while (0)
#define GN_LAUNCH_KA(kernel, act, cfg, args)                                                   \
    do {                                                                                       \
        if (gn_sweep()) GN_LAUNCH_KA_(kernel, GN_K_PASS, act, cfg, args);                      \
        else GN_LAUNCH_KA_(kernel, 0, act, cfg, args);                                         \
    } while (0)

// Samples per launch group: all of them unless GENIE_GN_CHUNK_MB > 0 and the tensors streamed per sample (`streams` of them, each
// `bytes_per_sample`) exceed that many MiB in total; then as many whole samples as fit.
static int gn_chunk_samples(int N, long long bytes_per_sample, int streams) {
    static long long chunk = -1;
    if (chunk < 0) { const char* e = getenv("GENIE_GN_CHUNK_MB"); chunk = e ? atoll(e) * (1ll << 20) : GENIE_GN_CHUNK_DEFAULT_MB * (1ll << 20); }
    if (chunk <= 0 || bytes_per_sample * streams * N <= chunk) return N;
    long long n = chunk / (bytes_per_sample * streams);
    return (int)(n < 1 ? 1 : n);
}

static int gn_fused_fwd_try(const void* x, void* y, int N, long long npix, int C, int Cp, int G, const float* gamma, const float* beta,
                            const float* ada_s, const float* ada_b, float eps, int act, float* mean, float* rstd, float* ws, long long ws_floats,
                            hipStream_t s);      // below: the one-pass forward

extern "C" int genie_groupnorm_fwd(const void* x, void* y, int N, int64_t npix, int C, int cpitch, int G, const float* gamma,
                                   const float* beta, const float* ada_scale, const float* ada_shift, float eps, int act,
                                   float* mean, float* rstd, float* ws, void* stream) {
    GENIE_CHECK_ARG(x && y && mean && rstd && ws, "genie_groupnorm_fwd: null pointer");
    GENIE_CHECK_ARG(G >= 1 && C % G == 0, "genie_groupnorm_fwd: num_channels %d must be divisible by num_groups %d", C, G);
    GENIE_CHECK_ARG(cpitch % 8 == 0 && cpitch >= C, "genie_groupnorm_fwd: bad channel pitch %d for C=%d", cpitch, C);
    if (N == 0 || npix == 0) return GENIE_OK;
    hipStream_t s = (hipStream_t)stream;
    {
        const int r = gn_fused_fwd_try(x, y, N, npix, C, cpitch, G, gamma, beta, ada_scale, ada_shift, eps, act, mean, rstd, ws,
                                       genie_groupnorm_ws_floats(N, C, G), s);
        if (r <= 0) return r;
    }
    // Sample chunks: statistics and apply of a chunk run back to back, so the apply pass re-reads the chunk out of the
    // memory-side cache (256 MB Infinity Cache) instead of HBM -- one HBM read + one write per element instead of two reads.
    const int nsub = gn_chunk_samples(N, npix * cpitch * 2ll, 1);
    for (int n0 = 0; n0 < N; n0 += nsub) {
        const int nn = N - n0 < nsub ? N - n0 : nsub;
        const GnGeom g = gn_geom(nn, npix, C, cpitch, G);
        const bf16_t* xs = (const bf16_t*)x + (long long)n0 * npix * cpitch;
        bf16_t* ys = (bf16_t*)y + (long long)n0 * npix * cpitch;
        const float* as = ada_scale ? ada_scale + (long long)n0 * C : nullptr;
        const float* ab = ada_shift ? ada_shift + (long long)n0 * C : nullptr;
        GN_LAUNCH_K(gn_stats_kernel, (dim3(g.nblk, nn), 256, 0, s), (xs, g, ws));
        GENIE_CHECK_LAUNCH();
        gn_finalize_kernel<<<dim3(G, nn), GN_FIN_THREADS, 0, s>>>(ws, g, eps, mean + (long long)n0 * G, rstd + (long long)n0 * G);
        GENIE_CHECK_LAUNCH();
        GN_LAUNCH_KA(gn_apply_kernel, act, (dim3(gn_geom_apply(g).nblk, nn), 256, 0, s), (xs, ys, gn_geom_apply(g), gamma, beta, as, ab, mean + (long long)n0 * G, rstd + (long long)n0 * G));
        GENIE_CHECK_LAUNCH();
    }
    return GENIE_OK;
}

// ---- fused forward: statistics and apply in ONE pass over HBM (1 read + 1 write) -------------------------------------------------
// The two-pass forward above reads x twice (the 1-GB activations of a 64-clip step do not survive in any cache between the passes); it
// runs at what a device copy reaches on the bytes it moves, so only moving fewer bytes helps.  Here a clip's slice stays ON CHIP between
// the statistics and the apply -- in REGISTERS: a block of 512 threads loads NI 16-byte chunks per thread (NI x 8 KB contiguous bytes of
// the clip, all loads in flight at once), sums them, the `bpc` blocks of the clip exchange their per-group partial sums through global
// memory (the exchange is the barrier among the blocks of ONE clip, see gn_publish / gn_await), every block finishes mean / rstd from the partials in a
// fixed order (fp64), applies scale / shift / activation to the registers and stores.  The grid is sized to what is RESIDENT at once
// (hipOccupancy... x CUs): R = resident / bpc clips per round, ceil(N / R) rounds; the blocks of a clip therefore always run together and
// the spin can only wait for blocks that are executing (a bound on the spin turns a would-be hang into an error flag all the same).
// Two blocks per CU cover each other's barrier wait.  Needs: 8 channels of a chunk in one group ((C / G) % 8 == 0), chunks per pixel
// CH = Cp / 8 a power of two <= 64 (thread t owns chunk t % CH for its lifetime), bpc <= resident blocks.  Everything else takes the
// two-pass path.
static __device__ int g_gn_fused_err;

struct GnFusedArgs {
    const bf16_t* x; bf16_t* y;
    int N, C, Cp, G, CH;
    long long npix, clip_chunks;         // 16-byte chunks per clip = npix * CH
    int bpc, R, rounds;                  // blocks per clip, clips per round, rounds
    const float *gamma, *beta;
    float eps; int act;
    float *mean, *rstd;
    unsigned long long* part;            // [N][bpc][G] (sum, sum of squares) as one 64-bit word; all-ones = not there yet
};

// The exchange IS the barrier: a block publishes its (sum, sum of squares) of a group as ONE 64-bit device-scope store into its own entry of
// the clip's table (pre-set to all-ones = NaN bits by a memset in front of the launch); a reader polls the entries it needs until their
// upper half is no longer the NaN pattern.  No read-modify-write on a shared address (256 arrivals at one counter serialise at the memory
// side and cost ~60 us per round), no fences (data and flag are the same atomic word; device-scope accesses bypass the per-XCD L2s).
__device__ __forceinline__ void gn_publish(unsigned long long* entry, float s, float q) {
    const unsigned long long w = (unsigned long long)__float_as_uint(s) | ((unsigned long long)__float_as_uint(q) << 32);
    __hip_atomic_store(entry, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void gn_await(const unsigned long long* entry, float& s, float& q) {
    unsigned long long w;
    unsigned spins = 0;
    for (;;) {
        w = __hip_atomic_load(entry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(w >> 32) != 0xFFFFFFFFu) break;
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1u << 22)) { atomicExch(&g_gn_fused_err, 1); break; }             // never hang the GPU: flag and carry on
    }
    s = __uint_as_float((unsigned)w);
    q = __uint_as_float((unsigned)(w >> 32));
}

// SETS = 2: two register sets of NI chunks, exchange pipelined across rounds.  SETS = 1: one set of NI chunks (twice the slice per block, so half
// the blocks per clip and TWO clips in flight); the odd clip slots start half a round late, so that one slot's exchange / arithmetic falls into
// the other's transfer (0.71 ms; a strict hand-over -- clip n loads when clip n - 1 has published -- is a chain of its own: 0.76 ms).
template <int NI, int SETS>
__global__ void __launch_bounds__(512, 4) gn_fused_fwd_kernel(const GnFusedArgs a) {
    __shared__ float red[512 * 2];
    __shared__ double dred[16];
    __shared__ float stat[64 * 2];                       // mean, rstd per group (G <= 64)
    const int tid = threadIdx.x;
    const int slot = blockIdx.x / a.bpc, blk = blockIdx.x - slot * a.bpc;
    const int cc = tid % a.CH;                           // this thread's 16-byte chunk of every pixel it touches
    const int cpg = a.CH / a.G;                          // chunks per group
    const int grp = cc / cpg;
    const long long f0 = (long long)blk * (512 * NI) + tid;     // first chunk of the clip this thread owns; then + 512 per i

    float ga[8], be[8];                                  // this thread's 8 channels (pad channels: 0)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = cc * 8 + j;
        ga[j] = c < a.C ? (a.gamma ? a.gamma[c] : 1.f) : 0.f;
        be[j] = c < a.C ? (a.beta ? a.beta[c] : 0.f) : 0.f;
    }
    auto load = [&](u32x4_t (&v)[NI], int round) {
        const int n = round * a.R + slot;
        const bf16_t* xs = a.x + (long long)n * a.clip_chunks * 8;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const long long f = f0 + 512ll * i;
            v[i] = f < a.clip_chunks ? *reinterpret_cast<const u32x4_t*>(xs + f * 8) : u32x4_t{0u, 0u, 0u, 0u};
        }
    };
    // stage A of a round: sums of the registers, block partial, publish
    auto stage_a = [&](u32x4_t (&v)[NI], int round) {
        const int n = round * a.R + slot;
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            float f8[8];
            unpack8(v[i], f8);
#pragma unroll
            for (int j = 0; j < 8; ++j) { s += f8[j]; q += f8[j] * f8[j]; }
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) asm volatile("" : "+v"(v[i]));         // keep the PACKED chunks across the exchange, not 8 floats each
        // block partials per group, fixed order: threads t, t + 64, ... share a chunk index (CH divides 64), so 64 threads fold the
        // eight waves, then one thread per group folds its chunks
        red[tid * 2] = s; red[tid * 2 + 1] = q;
        __syncthreads();
        if (tid < 64) {
            float ts = 0.f, tq = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) { ts += red[(tid + 64 * k) * 2]; tq += red[(tid + 64 * k) * 2 + 1]; }
            red[tid * 2] = ts; red[tid * 2 + 1] = tq;    // (slot tid is only read by thread tid above)
        }
        __syncthreads();
        if (tid < a.G) {
            float ts = 0.f, tq = 0.f;
            for (int t = 0; t < 64; ++t)
                if ((t % a.CH) / cpg == tid) { ts += red[t * 2]; tq += red[t * 2 + 1]; }
            gn_publish(a.part + ((long long)n * a.bpc + blk) * a.G + tid, ts, tq);
        }
        __syncthreads();                                 // red is reused
    };
    // stage B: wait for the clip's other blocks, statistics, apply, store
    auto stage_b = [&](u32x4_t (&v)[NI], int round) {
        const int n = round * a.R + slot;
        bf16_t* ys = a.y + (long long)n * a.clip_chunks * 8;
        // group totals from the bpc partials: thread t takes blocks t, t + 512, ... (ONE device-scope round trip for all of them instead of a
        // chain of them in one wave), folded in a fixed order, fp64 across blocks as in gn_finalize_kernel; waiting for an entry = waiting
        // for its block
        for (int g = 0; g < a.G; ++g) {
            double ds = 0.0, dq = 0.0;
            for (int b = tid; b < a.bpc; b += 512) {
                float ps, pq;
                gn_await(a.part + ((long long)n * a.bpc + b) * a.G + g, ps, pq);
                ds += (double)ps;
                dq += (double)pq;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { ds += __shfl_xor(ds, o, 64); dq += __shfl_xor(dq, o, 64); }
            if ((tid & 63) == 0) { dred[(tid >> 6) * 2] = ds; dred[(tid >> 6) * 2 + 1] = dq; }
            __syncthreads();
            if (tid == 0) {
                double S = 0.0, Q = 0.0;
#pragma unroll
                for (int w = 0; w < 8; ++w) { S += dred[w * 2]; Q += dred[w * 2 + 1]; }
                const double cnt = (double)(a.C / a.G) * (double)a.npix;
                const double m = S / cnt;
                double var = Q / cnt - m * m;
                if (var < 0.0) var = 0.0;
                const float mf = (float)m, rf = (float)(1.0 / sqrt(var + (double)a.eps));
                stat[g * 2] = mf; stat[g * 2 + 1] = rf;
                if (blk == 0) { a.mean[n * a.G + g] = mf; a.rstd[n * a.G + g] = rf; }
            }
            __syncthreads();
        }
        // coefficients of this thread's 8 channels: z = x * ca + cb  (gamma / beta were loaded once, in front of the rounds: a global load
        // HERE waits behind the next round's prefetch and the previous round's stores -- eight such waits were 16 us per round)
        float ca[8], cb[8];
        {
            const float mu = stat[grp * 2], rs = stat[grp * 2 + 1];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                ca[j] = rs * ga[j];
                cb[j] = be[j] - mu * rs * ga[j];
            }
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const long long f = f0 + 512ll * i;
            if (f < a.clip_chunks) {
                float f8[8];
                unpack8(v[i], f8);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float z = f8[j] * ca[j] + cb[j];
                    f8[j] = a.act == 1 ? silu_f(z) : (a.act == 2 ? (z > 0.f ? z : 0.01f * z) : z);
                }
                *reinterpret_cast<u32x4_t*>(ys + f * 8) = pack8(f8);
            }
        }
        __syncthreads();                                 // red / stat are reused by the next round
    };
    // Two register sets, software-pipelined ACROSS the exchange: round r + 1 is loaded, summed and published BEFORE round r waits for its
    // clip's other blocks -- by the time a block asks for the partials of round r + 1 everybody published them an iteration ago, and the
    // stores of round r overlap the loads of round r + 2.  (`rounds_slot` is the same for every block of a slot: they leave together.)
    const int rs = (a.N - slot + a.R - 1) / a.R;
    if constexpr (SETS == 1) {
        u32x4_t va[NI];
        if (slot & 1) {                                  // de-phase the odd slots by about half a round (~6 us)
            __builtin_amdgcn_s_sleep(127);
            __builtin_amdgcn_s_sleep(127);
        }
        for (int r = 0; r < rs; ++r) {
            load(va, r);
            stage_a(va, r);
            stage_b(va, r);
        }
        return;
    }
    u32x4_t va[NI], vb[NI];
    if (rs > 0) {
        load(va, 0);
        if (rs > 1) load(vb, 1);
        stage_a(va, 0);
    }
    for (int r = 0; r < rs; r += 2) {                    // invariant: va = round r with stage A done, vb = round r + 1 loaded
        if (r + 1 < rs) stage_a(vb, r + 1);
        stage_b(va, r);
        if (r + 2 < rs) load(va, r + 2);
        if (r + 1 < rs) {
            if (r + 2 < rs) stage_a(va, r + 2);
            stage_b(vb, r + 1);
            if (r + 3 < rs) load(vb, r + 3);
        }
    }
}

extern "C" int genie_gn_fused_error(void) {              // tests: did any clip barrier give up?  (synchronises)
    int v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_gn_fused_err), sizeof(int)) != hipSuccess) return -1;
    return v;
}

// returns 1 when the fused path does not apply (caller runs the two-pass path), 0 on launch, < 0 on error
static int gn_fused_fwd_try(const void* x, void* y, int N, long long npix, int C, int Cp, int G, const float* gamma, const float* beta,
                            const float* ada_s, const float* ada_b, float eps, int act, float* mean, float* rstd, float* ws, long long ws_floats,
                            hipStream_t s) {
    // OFF by default (GENIE_GN_FUSED=1 / 2 enable the two forms): measured SLOWER than the two-pass forward -- 0.88 / 0.71 vs 0.64 ms on 64 clips of
    // 128 x 16x64x64 (0.116 vs 0.080 ms on 8), step 471 vs 463 ms.  One read + one write instead of two reads + one write, but every round
    // is a chain of latencies (load -> sum -> publish -> device-scope round trip -> statistics -> store) that 2 blocks x 32 KB per CU do
    // not cover: 14 us per 16.8-MB clip against 6.5 us of transfer.  History of the number: one shared counter per clip 2.5 ms (256
    // arrivals serialise at the memory side), publish / poll entries 0.86 ms, parameter loads hoisted out of the rounds 1.04 -> ...,
    // exchange pipelined across rounds 0.90 ms.  DESIGN.md section 8.
    static const int on = getenv("GENIE_GN_FUSED") ? atoi(getenv("GENIE_GN_FUSED")) : 0;
    if (!on) return 1;
    const int CH = Cp / 8;
    if (C != Cp || C % G != 0 || (C / G) % 8 != 0 || G > 64 || CH > 64 || (CH & (CH - 1)) != 0) return 1;
    static const int max_g = getenv("GENIE_GN_FUSED_MAXG") ? atoi(getenv("GENIE_GN_FUSED_MAXG")) : 1;      // the exchange walks the groups one after the other: G = 1 (the residual blocks) by default
    if (G > max_g || ada_s || ada_b) return 1;           // (per-sample adaptive scale / shift: two-pass path)
    static int resident[4] = {0, 0, 0, 0};               // blocks of 512 threads resident on the device, per instantiation
    static int ncu = 0;
    if (!ncu) {
        hipDeviceProp_t pr;
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return 1;
        ncu = pr.multiProcessorCount;
    }
    const long long chunks = npix * CH;
    int best_ni = -1, best_bpc = 0, best_R = 0;
    long long best_used = -1;
    // GENIE_GN_FUSED=1: two register sets of 4 / 2 chunks; =2: one set of 8 / 4 chunks, odd clip slots de-phased
    const int nis[4] = {4, 2, 8, 4};
    const void* fns[4] = {(const void*)gn_fused_fwd_kernel<4, 2>, (const void*)gn_fused_fwd_kernel<2, 2>, (const void*)gn_fused_fwd_kernel<8, 1>,
                          (const void*)gn_fused_fwd_kernel<4, 1>};
    for (int k = (on == 2 ? 2 : 0); k < (on == 2 ? 4 : 2); ++k) {
        if (!resident[k]) {
            int nb = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fns[k], 512, 0) != hipSuccess || nb < 1) return 1;
            resident[k] = nb * ncu;
        }
        const long long bpc = (chunks + 512ll * nis[k] - 1) / (512ll * nis[k]);
        if (bpc > resident[k]) continue;
        long long R = resident[k] / bpc;
        if (R > N) R = N;
        const long long used = R * bpc;
        if (used > best_used) { best_used = used; best_ni = k; best_bpc = (int)bpc; best_R = (int)R; }
    }
    if (best_ni < 0) return 1;
    const long long need = (long long)N * best_bpc * G * 2 + 64;
    if (need > ws_floats) return 1;
    GnFusedArgs a;
    a.x = (const bf16_t*)x; a.y = (bf16_t*)y; a.N = N; a.C = C; a.Cp = Cp; a.G = G; a.CH = CH; a.npix = npix; a.clip_chunks = chunks;
    a.bpc = best_bpc; a.R = best_R; a.rounds = (N + best_R - 1) / best_R;
    a.gamma = gamma; a.beta = beta; a.eps = eps; a.act = act; a.mean = mean; a.rstd = rstd;
    a.part = reinterpret_cast<unsigned long long*>(ws);
    if (hipMemsetAsync(a.part, 0xFF, sizeof(unsigned long long) * (size_t)N * best_bpc * G, s) != hipSuccess) return 1;
    const dim3 grid(best_R * best_bpc);
    switch (best_ni) {
        case 0: gn_fused_fwd_kernel<4, 2><<<grid, 512, 0, s>>>(a); break;
        case 1: gn_fused_fwd_kernel<2, 2><<<grid, 512, 0, s>>>(a); break;
        case 2: gn_fused_fwd_kernel<8, 1><<<grid, 512, 0, s>>>(a); break;
        default: gn_fused_fwd_kernel<4, 1><<<grid, 512, 0, s>>>(a); break;
    }
    GENIE_CHECK_LAUNCH();
    return 0;
}

// ---- backward -------------------------------------------------------------------------------------
// z = xhat * gam_eff + b_eff,  dz = dy * act'(z);  S1[n,c] = sum dz,  S2[n,c] = sum dz * xhat
template <int K, int ACT>
__global__ void __launch_bounds__(256) gn_bwd_reduce_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, GnGeom g,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ ada_s, const float* __restrict__ ada_b,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            float* __restrict__ part) {
    __shared__ float red[256 * 16];
    const int n = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x;
    const long long p0 = (long long)blk * g.pix_per_blk;
    long long p1 = p0 + g.pix_per_blk;
    if (p1 > g.npix) p1 = g.npix;
    const bf16_t* xs = x + (long long)n * g.npix * g.Cp;
    const bf16_t* ds = dy + (long long)n * g.npix * g.Cp;
    for (int cc0 = 0; cc0 < g.CH; cc0 += 256) {
        const int chb = g.CH - cc0 < 256 ? g.CH - cc0 : 256;
        const int R = 256 / chb;
        const int pr = tid / chb, cc = cc0 + tid % chb;
        float s1[8], s2[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
        if (pr < R) {
            float a[8], b[8], mu[8], rs[8];
            {
                float ge[8];
                gn_coef8(n, cc * 8, g, gamma, beta, ada_s, ada_b, mean, rstd, a, b, mu, rs, ge);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) mu[j] = -mu[j] * rs[j];
            if (K == 0) {
    #pragma unroll 2
        for (long long p = p0 + pr; p < p1; p += R) {
                float f[8], d[8];
                unpack8(*reinterpret_cast<const u32x4_t*>(xs + p * g.Cp + cc * 8), f);
                unpack8(*reinterpret_cast<const u32x4_t*>(ds + p * g.Cp + cc * 8), d);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float z = f[j] * a[j] + b[j];
                    const float dz = gn_act_bwd<ACT>(z, d[j]);
                    s1[j] += dz;
                    s2[j] += dz * (f[j] * rs[j] + mu[j]);      // mu[j] holds -mean * rstd
                }
            }
            } else {
                constexpr int KK = K > 0 ? K : 1;
                for (long long pc = (long long)blk * (KK * R); pc < g.npix; pc += (long long)g.nblk * (KK * R)) {
                    u32x4_t vx[KK], vd[KK];
#pragma unroll
                    for (int k = 0; k < KK; ++k) {
                        const long long p = pc + k * R + pr;
                        const bool ok = p < g.npix;
                        vx[k] = ok ? gn_ld_stream(xs + p * g.Cp + cc * 8) : u32x4_t{0u, 0u, 0u, 0u};
                        vd[k] = ok ? gn_ld_stream(ds + p * g.Cp + cc * 8) : u32x4_t{0u, 0u, 0u, 0u};      // dy = 0: the pixel adds nothing
                    }
#pragma unroll
                    for (int k = 0; k < KK; ++k) {
                        float f[8], d[8];
                        unpack8(vx[k], f);
                        unpack8(vd[k], d);
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const float z = f[j] * a[j] + b[j];
                            const float dz = gn_act_bwd<ACT>(z, d[j]);
                            s1[j] += dz;
                            s2[j] += dz * (f[j] * rs[j] + mu[j]);      // mu[j] holds -mean * rstd
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { red[tid * 16 + j] = s1[j]; red[tid * 16 + 8 + j] = s2[j]; }
        __syncthreads();
        if (tid < chb) {
            float t1[8], t2[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) t1[j] = t2[j] = 0.f;
            for (int r = 0; r < R; ++r) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { t1[j] += red[(r * chb + tid) * 16 + j]; t2[j] += red[(r * chb + tid) * 16 + 8 + j]; }
            }
            float* o = part + (((long long)n * g.nblk + blk) * g.Cp + (cc0 + tid) * 8) * 2;
#pragma unroll
            for (int j = 0; j < 8; ++j) { o[2 * j] = t1[j]; o[2 * j + 1] = t2[j]; }
        }
        __syncthreads();
    }
}

// backward finalize: one 1024-thread block per (n, group).  tpc threads share a channel (each sums a slice of the blocks, loads 8
// deep), combined in a fixed order through LDS: totals over the blocks (fp64) -> parameter gradients, then the group totals ->
// k2, k3 of  dx = k1 * dz + k2 * x + k3.
__global__ void __launch_bounds__(GN_FIN_THREADS) gn_bwd_finalize_kernel(const float* __restrict__ part, GnGeom g, const float* __restrict__ gamma,
                                                                         const float* __restrict__ beta, const float* __restrict__ ada_s,
                                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                         float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                         float* __restrict__ dada_s, float* __restrict__ dada_b,
                                                                         float* __restrict__ kcoef) {
    __shared__ double red[GN_FIN_THREADS / 64];
    __shared__ double part_s[GN_FIN_THREADS][2];
    const int n = blockIdx.y, grp = blockIdx.x, tid = threadIdx.x;
    const int cg = g.C / g.G;
    int tpc = 1;
    while (tpc < 32 && cg * tpc * 2 <= GN_FIN_THREADS) tpc *= 2;     // power of two, cg * tpc <= 1024
    const int cpp = GN_FIN_THREADS / tpc;                            // channels per pass
    double P1 = 0.0, P2 = 0.0;
    for (int c0 = 0; c0 < cg; c0 += cpp) {
        const int ci = c0 + tid / tpc, sub = tid % tpc;
        double s1 = 0.0, s2 = 0.0;
        if (ci < cg) {
            const float* o = part + ((long long)n * g.nblk * g.Cp + (grp * cg + ci)) * 2;
#pragma unroll 8
            for (int blk = sub; blk < g.nblk; blk += tpc) {
                const float2 v = *reinterpret_cast<const float2*>(o + (long long)blk * g.Cp * 2);
                s1 += (double)v.x;
                s2 += (double)v.y;
            }
        }
        __syncthreads();
        part_s[tid][0] = s1; part_s[tid][1] = s2;
        __syncthreads();
        if (sub == 0 && ci < cg) {
            for (int u = 1; u < tpc; ++u) { s1 += part_s[tid + u][0]; s2 += part_s[tid + u][1]; }
            const int c = grp * cg + ci;
            const float ga = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
            const float as = ada_s ? ada_s[(long long)n * g.C + c] : 1.f;
            if (dgamma) atomicAdd(dgamma + c, (float)(s2 * as));
            if (dbeta) atomicAdd(dbeta + c, (float)(s1 * as));
            if (dada_s) dada_s[(long long)n * g.C + c] = (float)(ga * s2 + be * s1);
            if (dada_b) dada_b[(long long)n * g.C + c] = (float)s1;
            // same roundings as the two-kernel version: per-channel weighted totals go through fp32
            P1 += (double)(float)((double)(ga * as) * s1);
            P2 += (double)(float)((double)(ga * as) * s2);
        }
    }
    P1 = block_sum_fin(P1, red);
    P2 = block_sum_fin(P2, red);
    if (tid == 0) {
        const double M = (double)cg * (double)g.npix;
        const double rs = (double)rstd[n * g.G + grp], mu = (double)mean[n * g.G + grp];
        kcoef[(n * g.G + grp) * 2 + 0] = (float)(-rs * rs * P2 / M);
        kcoef[(n * g.G + grp) * 2 + 1] = (float)(-rs * P1 / M + rs * rs * mu * P2 / M);
    }
}

template <int K, int ACT>
__global__ void __launch_bounds__(256) gn_bwd_apply_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                           bf16_t* __restrict__ dx, GnGeom g, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ ada_s,
                                                           const float* __restrict__ ada_b, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, const float* __restrict__ kcoef) {
    const int n = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x;
    const long long p0 = (long long)blk * g.pix_per_blk;
    long long p1 = p0 + g.pix_per_blk;
    if (p1 > g.npix) p1 = g.npix;
    const bf16_t* xs = x + (long long)n * g.npix * g.Cp;
    const bf16_t* ds = dy + (long long)n * g.npix * g.Cp;
    bf16_t* os = dx + (long long)n * g.npix * g.Cp;
    for (int cc0 = 0; cc0 < g.CH; cc0 += 256) {
        const int chb = g.CH - cc0 < 256 ? g.CH - cc0 : 256;
        const int R = 256 / chb;
        const int pr = tid / chb, cc = cc0 + tid % chb;
        if (pr >= R) continue;
        float a[8], b[8], k1[8], k2[8], k3[8];
        {
            float mu[8], rs[8], ge[8];
            gn_coef8(n, cc * 8, g, gamma, beta, ada_s, ada_b, mean, rstd, a, b, mu, rs, ge);
            const int cg = g.C / g.G;
            const bool one_group = cc * 8 + 8 <= g.C && (cg & 7) == 0;
            const float2 kk = one_group ? *reinterpret_cast<const float2*>(kcoef + (n * g.G + cc * 8 / cg) * 2) : float2{0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = cc * 8 + j;
                if (one_group) {
                    k1[j] = rs[j] * ge[j]; k2[j] = kk.x; k3[j] = kk.y;
                } else if (c < g.C) {
                    const int grp = c / cg;
                    k1[j] = rs[j] * ge[j];
                    k2[j] = kcoef[(n * g.G + grp) * 2 + 0];
                    k3[j] = kcoef[(n * g.G + grp) * 2 + 1];
                } else {
                    k1[j] = k2[j] = k3[j] = 0.f;
                }
            }
        }
        if (K == 0) {
#pragma unroll 2
        for (long long p = p0 + pr; p < p1; p += R) {
            float f[8], d[8];
            unpack8(*reinterpret_cast<const u32x4_t*>(xs + p * g.Cp + cc * 8), f);
            unpack8(*reinterpret_cast<const u32x4_t*>(ds + p * g.Cp + cc * 8), d);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float z = f[j] * a[j] + b[j];
                const float dz = gn_act_bwd<ACT>(z, d[j]);
                d[j] = k1[j] * dz + k2[j] * f[j] + k3[j];
            }
            *reinterpret_cast<u32x4_t*>(os + p * g.Cp + cc * 8) = pack8(d);
        }
        } else {
            constexpr int KK = K > 0 ? K : 1;
            for (long long pc = (long long)blk * (KK * R); pc < g.npix; pc += (long long)g.nblk * (KK * R)) {
                u32x4_t vx[KK], vd[KK];
#pragma unroll
                for (int k = 0; k < KK; ++k) {
                    const long long p = pc + k * R + pr;
                    const bool ok = p < g.npix;
                    vx[k] = ok ? gn_ld_stream(xs + p * g.Cp + cc * 8) : u32x4_t{0u, 0u, 0u, 0u};
                    vd[k] = ok ? gn_ld_stream(ds + p * g.Cp + cc * 8) : u32x4_t{0u, 0u, 0u, 0u};
                }
#pragma unroll
                for (int k = 0; k < KK; ++k) {
                    const long long p = pc + k * R + pr;
                    float f[8], d[8];
                    unpack8(vx[k], f);
                    unpack8(vd[k], d);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float z = f[j] * a[j] + b[j];
                        const float dz = gn_act_bwd<ACT>(z, d[j]);
                        d[j] = k1[j] * dz + k2[j] * f[j] + k3[j];
                    }
                    if (p < g.npix) *reinterpret_cast<u32x4_t*>(os + p * g.Cp + cc * 8) = pack8(d);
                }
            }
        }
    }
}

extern "C" int genie_groupnorm_bwd(const void* x, const void* dy, void* dx, int N, int64_t npix, int C, int cpitch, int G,
                                   const float* gamma, const float* beta, const float* ada_scale, const float* ada_shift, int act,
                                   const float* mean, const float* rstd, float* dgamma, float* dbeta, float* dada_scale,
                                   float* dada_shift, float* ws, void* stream) {
    GENIE_CHECK_ARG(x && dy && dx && mean && rstd && ws, "genie_groupnorm_bwd: null pointer");
    GENIE_CHECK_ARG(G >= 1 && C % G == 0, "genie_groupnorm_bwd: num_channels %d must be divisible by num_groups %d", C, G);
    GENIE_CHECK_ARG(cpitch % 8 == 0 && cpitch >= C, "genie_groupnorm_bwd: bad channel pitch %d for C=%d", cpitch, C);
    if (N == 0 || npix == 0) return GENIE_OK;
    hipStream_t s = (hipStream_t)stream;
    float* chan = ws + (long long)N * GN_MAX_BLK * cpitch * 2;
    float* kcoef = chan + (long long)N * cpitch * 4;
    const int nsub = gn_chunk_samples(N, npix * cpitch * 2ll, 2);          // x and dy of a chunk stay in the memory-side cache
    for (int n0 = 0; n0 < N; n0 += nsub) {
        const int nn = N - n0 < nsub ? N - n0 : nsub;
        const GnGeom g = gn_geom(nn, npix, C, cpitch, G);
        const long long eo = (long long)n0 * npix * cpitch;
        const bf16_t* xs = (const bf16_t*)x + eo;
        const bf16_t* dys = (const bf16_t*)dy + eo;
        const float* as = ada_scale ? ada_scale + (long long)n0 * C : nullptr;
        const float* ab = ada_shift ? ada_shift + (long long)n0 * C : nullptr;
        const float* mu = mean + (long long)n0 * G;
        const float* rs = rstd + (long long)n0 * G;
        GN_LAUNCH_KA(gn_bwd_reduce_kernel, act, (dim3(g.nblk, nn), 256, 0, s), (xs, dys, g, gamma, beta, as, ab, mu, rs, ws));
        GENIE_CHECK_LAUNCH();
        gn_bwd_finalize_kernel<<<dim3(G, nn), GN_FIN_THREADS, 0, s>>>(ws, g, gamma, beta, as, mu, rs, dgamma, dbeta, dada_scale ? dada_scale + (long long)n0 * C : nullptr,
                                                                      dada_shift ? dada_shift + (long long)n0 * C : nullptr, kcoef);
        GENIE_CHECK_LAUNCH();
        GN_LAUNCH_KA(gn_bwd_apply_kernel, act, (dim3(gn_geom_apply(g).nblk, nn), 256, 0, s), (xs, dys, (bf16_t*)dx + eo, gn_geom_apply(g), gamma, beta, as, ab, mu, rs, kcoef));
        GENIE_CHECK_LAUNCH();
    }
    return GENIE_OK;
}


// ---- GroupNorm around convolutions that did part of the work in their epilogue (GenieConvDesc.gn_sums / gnb_part) --------------------
// One group, no adaptive scale / shift (the residual blocks of reference video.py:539-656).

// mean / rstd of sample n from the fp64 (sum, sum of squares) a conv epilogue accumulated over its output
__global__ void gn_finalize_sums_kernel(const double* __restrict__ sums, int N, double cnt, float eps, float* __restrict__ mean, float* __restrict__ rstd) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const double m = sums[2 * n] / cnt;
    double var = sums[2 * n + 1] / cnt - m * m;
    if (var < 0.0) var = 0.0;
    mean[n] = (float)m;
    rstd[n] = (float)(1.0 / sqrt(var + (double)eps));
}

extern "C" int genie_groupnorm_fwd_from_sums(const void* x, void* y, int N, int64_t npix, int C, int cpitch, const float* gamma, const float* beta,
                                             float eps, int act, float* mean, float* rstd, const double* sums, void* stream) {
    GENIE_CHECK_ARG(x && y && mean && rstd && sums, "genie_groupnorm_fwd_from_sums: null pointer");
    GENIE_CHECK_ARG(cpitch % 8 == 0 && cpitch >= C && C >= 1, "genie_groupnorm_fwd_from_sums: bad channel pitch %d for C=%d", cpitch, C);
    if (N == 0 || npix == 0) return GENIE_OK;
    hipStream_t s = (hipStream_t)stream;
    const GnGeom g = gn_geom(N, npix, C, cpitch, 1);
    gn_finalize_sums_kernel<<<(N + 255) / 256, 256, 0, s>>>(sums, N, (double)C * (double)npix, eps, mean, rstd);
    GENIE_CHECK_LAUNCH();
    GN_LAUNCH_KA(gn_apply_kernel, act, (dim3(gn_geom_apply(g).nblk, N), 256, 0, s), ((const bf16_t*)x, (bf16_t*)y, gn_geom_apply(g), gamma, beta, nullptr, nullptr, mean, rstd));
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}

extern "C" int64_t genie_groupnorm_bwd_from_part_ws_floats(int N) { return (int64_t)N * 2 + 64; }

extern "C" int genie_groupnorm_bwd_from_part(const void* x, const void* dy, void* dx, int N, int64_t npix, int C, int cpitch, const float* gamma,
                                             const float* beta, int act, const float* mean, const float* rstd, float* dgamma, float* dbeta,
                                             const float* part, int nblk, float* ws, void* stream) {
    GENIE_CHECK_ARG(x && dy && dx && mean && rstd && part && ws, "genie_groupnorm_bwd_from_part: null pointer");
    GENIE_CHECK_ARG(cpitch % 8 == 0 && cpitch >= C && C >= 1, "genie_groupnorm_bwd_from_part: bad channel pitch %d for C=%d", cpitch, C);
    GENIE_CHECK_ARG(nblk >= 1, "genie_groupnorm_bwd_from_part: nblk %d", nblk);
    if (N == 0 || npix == 0) return GENIE_OK;
    hipStream_t s = (hipStream_t)stream;
    const GnGeom g = gn_geom(N, npix, C, cpitch, 1);
    GnGeom gf = g;
    gf.nblk = nblk;                                      // the partials come one per 256-row conv tile, not one per block of the reduce pass
    gn_bwd_finalize_kernel<<<dim3(1, N), GN_FIN_THREADS, 0, s>>>(part, gf, gamma, beta, nullptr, mean, rstd, dgamma, dbeta, nullptr, nullptr, ws);
    GENIE_CHECK_LAUNCH();
    GN_LAUNCH_KA(gn_bwd_apply_kernel, act, (dim3(gn_geom_apply(g).nblk, N), 256, 0, s), ((const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx, gn_geom_apply(g), gamma, beta, nullptr, nullptr, mean, rstd, ws));
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}
