// Space-time transformer attention on gfx950 MFMA (reference genie/module/attention.py:154-371).
//
//   prologue : u = LayerNorm(rotary(x))              one pass, fp32 angles from a host table (attention.py:219-220)
//   core     : out = softmax(scale * Q K^T [causal]) V + resid,  Q = K = V = u unless a condition supplies K, V
//
// One flash-style kernel serves the spatial case (sequence = the H*W pixels of one frame, non-causal) and the
// temporal case (sequence = the T frames of one pixel, causal): a token's address is
//     base(seq) + pos * pos_stride + head * DH,   base(seq) = (seq / n_inner) * stride_outer + (seq % n_inner) * stride_inner
// so the reference's rearranges/pack/unpack (attention.py:289-306, 357-371) are pure address arithmetic.
//
// MFMA formulation ("swapped" operands, so that softmax is lane-local):
//   S^T[key][query] = mfma(A = K rows (ds_read_b128), B = Q rows (registers))       lane & 31 = query
//   P^T is already laid out as the B operand of the second product when V^T is read with the matching key
//   permutation, which is exactly what two ds_read_b64_tr_b16 per fragment deliver:
//   O^T[d][query]   = mfma(A = V^T (transposing LDS reads), B = P^T (registers))
// One wave = 32 queries; a workgroup of 1..4 waves shares the K/V tiles (64 keys) in LDS.
#include <type_traits>
#include "common.h"
#include "genie_hip.h"
#include "attn_args.h"
#include "attn_common.h"

static __device__ __attribute__((aligned(256))) uint32_t g_zero_page_a[64];

// ------------------------------------------------------------------------------------------------
// rotary + LayerNorm prologue: one wave per token, C <= 2048
// x, u: [ntok][C] bf16 rows (row pitch = pitch elements); pos(token) = (token / pos_div) % pos_mod
// cs: fp32 table [npos][C] holding cos in even slots and sin in odd slots of each feature PAIR:
//     cs[p][2i] = cos(p * freq_i), cs[p][2i+1] = sin(p * freq_i)
// NIT = 16-B chunks per lane (C <= 512 NIT); lanes past the row read chunk 0 and contribute zeros, so the streaming part has
// no divergent branches (the first version's per-element `gamma ? gamma[c] : 1` compiled to 64 branches with dword loads, and
// its 64-bit (token / pos_div) % pos_mod to four division expansions: 1 TB/s).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void rot_fwd8(float (&f)[8], const f32x4_t c0, const f32x4_t c1) {
    const float cc[4] = {c0[0], c0[2], c1[0], c1[2]}, sn[4] = {c0[1], c0[3], c1[1], c1[3]};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a = f[2 * j], b = f[2 * j + 1];
        f[2 * j] = a * cc[j] - b * sn[j];          // x1 cos - x2 sin
        f[2 * j + 1] = b * cc[j] + a * sn[j];      // x2 cos + x1 sin
    }
}
__device__ __forceinline__ void rot_bwd8(float (&f)[8], const f32x4_t c0, const f32x4_t c1) {      // transpose of rot_fwd8
    const float cc[4] = {c0[0], c0[2], c1[0], c1[2]}, sn[4] = {c0[1], c0[3], c1[1], c1[3]};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a = f[2 * j], b = f[2 * j + 1];
        f[2 * j] = a * cc[j] + b * sn[j];
        f[2 * j + 1] = b * cc[j] - a * sn[j];
    }
}
__device__ __forceinline__ void load8f(const float* p, float (&f)[8]) {
    const f32x4_t a = *reinterpret_cast<const f32x4_t*>(p), b = *reinterpret_cast<const f32x4_t*>(p + 4);
    f[0] = a[0]; f[1] = a[1]; f[2] = a[2]; f[3] = a[3]; f[4] = b[0]; f[5] = b[1]; f[6] = b[2]; f[7] = b[3];
}

// Both kernels are persistent: a wave keeps gamma / beta (and the dgamma / dbeta partial sums) in registers and walks the tokens
// in groups of TG, all loads of a group issued before its arithmetic.  Per-token waves re-read gamma and beta every time: every
// wave of the chip then hits the same 32 L2 lines, and that hot spot alone cost 100 us of a 128 us call (65536 x 512; 25 us
// without gamma / beta), far more than the 2 KB / token of the cos / sin table, which is spread over pos_mod rows.
template <int NIT, int TG, bool ROT>
__global__ void __launch_bounds__(256) rotary_ln_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ u, int ntok, int C, long long pitch,
                                                            const float* __restrict__ cs, unsigned pos_div, unsigned pos_mod,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                            float* __restrict__ stats /* [ntok][2] mean, rstd of the rotated row */) {
    const int lane = threadIdx.x & 63;
    const int wg = (int)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = (int)gridDim.x * 4;
    const int nch = C >> 3;
    const float invC = 1.f / (float)C;
    int chc[NIT];
    float ga[NIT][8], be[NIT][8];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int ch = lane + it * 64;
        chc[it] = ch < nch ? ch : 0;
        if (gamma) load8f(gamma + chc[it] * 8, ga[it]);
        if (beta) load8f(beta + chc[it] * 8, be[it]);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (!gamma) ga[it][j] = 1.f;
            if (!beta) be[it][j] = 0.f;
        }
    }
    for (int base = wg * TG; base < ntok; base += nwaves * TG) {
        u32x4_t raw[TG][NIT];
        f32x4_t c0[TG][NIT], c1[TG][NIT];
#pragma unroll
        for (int t = 0; t < TG; ++t) {
            const int tok = base + t < ntok ? base + t : ntok - 1;
            const float* csr = cs + (long long)(((unsigned)tok / pos_div) % pos_mod) * C;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                raw[t][it] = *reinterpret_cast<const u32x4_t*>(x + (long long)tok * pitch + chc[it] * 8);
                if (ROT) {
                    c0[t][it] = *reinterpret_cast<const f32x4_t*>(csr + chc[it] * 8);
                    c1[t][it] = *reinterpret_cast<const f32x4_t*>(csr + chc[it] * 8 + 4);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);                           // every load of the group is in flight before its arithmetic starts
#pragma unroll
        for (int t = 0; t < TG; ++t) {
            float v[NIT * 8];
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                float f[8];
                unpack8(raw[t][it], f);
                if (ROT) rot_fwd8(f, c0[t][it], c1[t][it]);
                const bool ok = lane + it * 64 < nch;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float e = ok ? f[j] : 0.f;
                    v[it * 8 + j] = e; s += e; q += e * e;
                }
            }
            s = wave_sum(s);
            q = wave_sum(q);
            const float mean = s * invC;
            float var = q * invC - mean * mean;
            var = var < 0.f ? 0.f : var;
            const float rstd = rsqrtf(var + eps);
            const int tok = base + t;
            if (tok < ntok) {
                if (lane == 0 && stats) { stats[(long long)tok * 2] = mean; stats[(long long)tok * 2 + 1] = rstd; }
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    float f[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) f[j] = (v[it * 8 + j] - mean) * rstd * ga[it][j] + be[it][j];
                    if (lane + it * 64 < nch) *reinterpret_cast<u32x4_t*>(u + (long long)tok * pitch + chc[it] * 8) = pack8(f);
                }
            }
        }
    }
}

// backward of u = LN(rot(x)):  dx = rot^T( LN'(du) ) (+ dres), dgamma += sum du * xhat, dbeta += sum du
// A lane owns the same channels for every token, so the dgamma / dbeta partial sums live in registers (the first version did
// two LDS atomics per element: 264 us per call at 65536 x 256).  The waves of a block combine through LDS atomics and the
// block issues ONE global atomic per channel: with a block per 4 tokens the global atomics (2 C per block onto 2 C addresses)
// took longer than the streaming (57 us at 8192 x 512).
template <int NIT, int TG, int NWV, bool ROT, bool RES>
__global__ void __launch_bounds__(64 * NWV) rotary_ln_bwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ du,
                                                                 const bf16_t* __restrict__ dres, bf16_t* __restrict__ dx, int ntok, int C,
                                                                 long long pitch, const float* __restrict__ cs, unsigned pos_div,
                                                                 unsigned pos_mod, const float* __restrict__ gamma,
                                                                 const float* __restrict__ stats, float* __restrict__ dgamma,
                                                                 float* __restrict__ dbeta) {
    extern __shared__ float sm[];     // [2][C] block sums of dgamma / dbeta
    const int lane = threadIdx.x & 63;
    const int wg = (int)blockIdx.x * NWV + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = (int)gridDim.x * NWV;
    const int nch = C >> 3;
    for (int i = threadIdx.x; i < 2 * C; i += 64 * NWV) sm[i] = 0.f;
    float gam[NIT * 8], ag[NIT * 8], ab[NIT * 8];
    int chc[NIT];
    float live[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int ch = lane + it * 64;
        chc[it] = ch < nch ? ch : 0;
        live[it] = ch < nch ? 1.f : 0.f;
        float ga[8];
        if (gamma) load8f(gamma + chc[it] * 8, ga);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            gam[it * 8 + j] = gamma ? ga[j] : 1.f;
            ag[it * 8 + j] = 0.f;
            ab[it * 8 + j] = 0.f;
        }
    }
    const float invC = 1.f / (float)C;
    for (int base = wg * TG; base < ntok; base += nwaves * TG) {
        u32x4_t xr[TG][NIT], dr[TG][NIT], rr[TG][NIT];
        f32x4_t c0[TG][NIT], c1[TG][NIT];
        float mean[TG], rstd[TG];
#pragma unroll
        for (int t = 0; t < TG; ++t) {
            const int tok = base + t < ntok ? base + t : ntok - 1;
            const float* csr = cs + (long long)(((unsigned)tok / pos_div) % pos_mod) * C;
            mean[t] = stats[(long long)tok * 2];
            rstd[t] = stats[(long long)tok * 2 + 1];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const long long o = (long long)tok * pitch + chc[it] * 8;
                xr[t][it] = *reinterpret_cast<const u32x4_t*>(x + o);
                dr[t][it] = *reinterpret_cast<const u32x4_t*>(du + o);
                if (RES) rr[t][it] = *reinterpret_cast<const u32x4_t*>(dres + o);
                if (ROT) {
                    c0[t][it] = *reinterpret_cast<const f32x4_t*>(csr + chc[it] * 8);
                    c1[t][it] = *reinterpret_cast<const f32x4_t*>(csr + chc[it] * 8 + 4);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);                           // every load of the group is in flight before its arithmetic starts
#pragma unroll
        for (int t = 0; t < TG; ++t) {
            const float tl = base + t < ntok ? 1.f : 0.f;            // the clamped tail tokens contribute nothing
            float xh[NIT * 8], g[NIT * 8];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                float f[8], d[8];
                unpack8(xr[t][it], f);
                unpack8(dr[t][it], d);
                if (ROT) rot_fwd8(f, c0[t][it], c1[t][it]);
                const float lv = live[it] * tl;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float dj = d[j] * lv;
                    const float h = (f[j] - mean[t]) * rstd[t];
                    const float gg = dj * gam[it * 8 + j];
                    xh[it * 8 + j] = h; g[it * 8 + j] = gg;
                    s1 += gg; s2 += gg * h;
                    ag[it * 8 + j] += dj * h;
                    ab[it * 8 + j] += dj;
                }
            }
            s1 = wave_sum(s1) * invC;
            s2 = wave_sum(s2) * invC;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                float f[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] = rstd[t] * (g[it * 8 + j] - s1 - xh[it * 8 + j] * s2);      // d/d(rotated x)
                if (ROT) rot_bwd8(f, c0[t][it], c1[t][it]);
                if (RES) {
                    float r[8];
                    unpack8(rr[t][it], r);
#pragma unroll
                    for (int j = 0; j < 8; ++j) f[j] += r[j];
                }
                if (lane + it * 64 < nch && base + t < ntok) *reinterpret_cast<u32x4_t*>(dx + (long long)(base + t) * pitch + chc[it] * 8) = pack8(f);
            }
        }
    }
    __syncthreads();                  // sm is zeroed
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        if (lane + it * 64 < nch) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                atomicAdd(&sm[chc[it] * 8 + j], ag[it * 8 + j]);
                atomicAdd(&sm[C + chc[it] * 8 + j], ab[it * 8 + j]);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C; i += 64 * NWV) {
        if (dgamma) atomicAdd(dgamma + i, sm[i]);
        if (dbeta) atomicAdd(dbeta + i, sm[C + i]);
    }
}

extern "C" int genie_rotary_layernorm_fwd(const void* x, void* u, int64_t ntok, int C, int64_t pitch, const float* cos_sin, int64_t pos_div,
                                          int pos_mod, const float* gamma, const float* beta, float eps, float* stats, void* stream) {
    GENIE_CHECK_ARG(x && u, "genie_rotary_layernorm_fwd: null pointer");
    GENIE_CHECK_ARG(C % 8 == 0 && C <= 2048 && pitch >= C && pitch % 8 == 0, "genie_rotary_layernorm_fwd: C=%d must be a multiple of 8, <= 2048, pitch %lld", C, (long long)pitch);
    GENIE_CHECK_ARG(pos_div >= 1 && pos_mod >= 1, "genie_rotary_layernorm_fwd: bad position spec");
    GENIE_CHECK_ARG(ntok < (1ll << 31) - 8 && pos_div < (1ll << 31), "genie_rotary_layernorm_fwd: more than 2^31 tokens");
    if (ntok == 0) return GENIE_OK;
    hipStream_t s = (hipStream_t)stream;
#define GENIE_LN_FWD(NITv, TGv)                                                                                       \
    do {                                                                                                              \
        long long blocks = (ntok + 4 * TGv - 1) / (4 * TGv);                                                          \
        if (blocks > 2048) blocks = 2048;                                                                             \
        if (cos_sin) rotary_ln_fwd_kernel<NITv, TGv, true><<<(unsigned)blocks, 256, 0, s>>>((const bf16_t*)x, (bf16_t*)u, (int)ntok, C, pitch, cos_sin, (unsigned)pos_div, (unsigned)pos_mod, gamma, beta, eps, stats); \
        else rotary_ln_fwd_kernel<NITv, TGv, false><<<(unsigned)blocks, 256, 0, s>>>((const bf16_t*)x, (bf16_t*)u, (int)ntok, C, pitch, cos_sin, (unsigned)pos_div, (unsigned)pos_mod, gamma, beta, eps, stats); \
    } while (0)
    if (C <= 512) GENIE_LN_FWD(1, 4);
    else if (C <= 1024) GENIE_LN_FWD(2, 2);
    else GENIE_LN_FWD(4, 1);
#undef GENIE_LN_FWD
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}

extern "C" int genie_rotary_layernorm_bwd(const void* x, const void* du, const void* dres, void* dx, int64_t ntok, int C, int64_t pitch,
                                          const float* cos_sin, int64_t pos_div, int pos_mod, const float* gamma, const float* stats,
                                          float* dgamma, float* dbeta, void* stream) {
    GENIE_CHECK_ARG(x && du && dx && stats, "genie_rotary_layernorm_bwd: null pointer");
    GENIE_CHECK_ARG(C % 8 == 0 && C <= 2048 && pitch >= C && pitch % 8 == 0, "genie_rotary_layernorm_bwd: bad C=%d / pitch", C);
    GENIE_CHECK_ARG(pos_div >= 1 && pos_mod >= 1, "genie_rotary_layernorm_bwd: bad position spec");
    GENIE_CHECK_ARG(ntok < (1ll << 31) - (1 << 20) && pos_div < (1ll << 31), "genie_rotary_layernorm_bwd: more than 2^31 tokens");
    if (ntok == 0) return GENIE_OK;
    hipStream_t s = (hipStream_t)stream;
#define GENIE_LN_BWD2(NITv, NWVv, ROTv, RESv)                                                                         \
    rotary_ln_bwd_kernel<NITv, 4 / NITv, NWVv, ROTv, RESv><<<(unsigned)blocks, 64 * NWVv, 2 * C * sizeof(float), s>>>(          \
        (const bf16_t*)x, (const bf16_t*)du, (const bf16_t*)dres, (bf16_t*)dx, (int)ntok, C, pitch, cos_sin, (unsigned)pos_div, (unsigned)pos_mod, gamma, stats, dgamma, dbeta)
#define GENIE_LN_BWD(NITv, NWVv)                                                                                      \
    do {                                                                                                              \
        long long blocks = (ntok + NWVv * (4 / NITv) - 1) / (NWVv * (4 / NITv));                                      \
        const long long cap = 512;                                    /* about one resident round of blocks (VGPR-bound occupancy) */ \
        if (blocks > cap) blocks = cap;                                                                               \
        if (cos_sin && dres) GENIE_LN_BWD2(NITv, NWVv, true, true);                                                   \
        else if (cos_sin) GENIE_LN_BWD2(NITv, NWVv, true, false);                                                     \
        else if (dres) GENIE_LN_BWD2(NITv, NWVv, false, true);                                                        \
        else GENIE_LN_BWD2(NITv, NWVv, false, false);                                                                 \
    } while (0)
    if (C <= 512) GENIE_LN_BWD(1, 8);
    else if (C <= 1024) GENIE_LN_BWD(2, 8);
    else GENIE_LN_BWD(4, 4);
#undef GENIE_LN_BWD
#undef GENIE_LN_BWD2
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}

// ------------------------------------------------------------------------------------------------
// attention core
// ------------------------------------------------------------------------------------------------
// Forward.  NW waves of 32 queries share 64-key K/V tiles that live in a ring of THREE LDS stages, DMA'd two tiles ahead
// (counted vmcnt + raw barrier: a whole tile stays in flight across the barrier).  Softmax runs in the exp2 domain with the
// scale folded into one FMA per score; masks are only evaluated on tiles that touch the end of the key range or the causal
// diagonal; the running-max rescale of O is skipped (wave-uniform branch) while no lane's maximum moves; P is rounded to bf16 by
// v_cvt_pk_bf16_f32.
template <int DH, int NW, bool KVSAME>
__global__ void __launch_bounds__(64 * NW) attn_fwd_kernel(const AttnArgs a) {
    constexpr int KT = 64;                       // keys per tile
    constexpr int ROWB = DH * 2;                 // bytes per LDS row
    constexpr int CPR = DH / 8;                  // 16-B chunks per row
    constexpr int TILE = KT * ROWB;
    constexpr int KS = DH / 16;                  // MFMA k-steps of the QK^T product
    constexpr int DT = DH / 32;                  // 32-row tiles of O^T
    constexpr int SLABS = TILE / 1024;           // 1-KiB DMA pieces per tile
    constexpr int LPW = SLABS / NW;              // pieces per wave per tile
    constexpr int LPT = KVSAME ? LPW : 2 * LPW;  // glds per wave per stage
    constexpr int STAGE = KVSAME ? TILE : 2 * TILE;
    constexpr int NSTAGE = 3;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qtiles = (a.Sq + 32 * NW - 1) / (32 * NW);
    const int seq = blockIdx.x / qtiles, qtile = blockIdx.x % qtiles, head = blockIdx.y;
    const int q0 = qtile * (32 * NW) + wave * 32;
    const int qi = q0 + (lane & 31);             // this lane's query
    const int h = lane >> 5;
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page_a);

    auto swz = [&](int row, int chunk) -> int { return attn_swz<CPR>(row, chunk); };

    // Q fragments (B operand): Q[query][16 ks + 8 h .. + 7]
    bf16x8_t qf[KS];
    {
        const bf16_t* qrow = a.q + seq_base(a.qm, seq) + (long long)(qi < a.Sq ? qi : 0) * a.qm.pos_stride + head * DH;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (qi < a.Sq) qf[ks] = *reinterpret_cast<const bf16x8_t*>(qrow + ks * 16 + h * 8);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) qf[ks][e] = 0;
            }
        }
    }

    f32x16_t oacc[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
    float m_run = -1e30f, l_run = 0.f;           // running max of the RAW scores, running sum of exp2(c * (s - m))
    const float c2 = a.scale * 1.4426950408889634f;
    const unsigned drop_key = attn_drop_seqkey(a.drop_key, seq, a.nhead, head);

    const long long kbase = seq_base(a.km, seq) + head * DH;
    const int blk_q_max = qtile * (32 * NW) + 32 * NW - 1;   // causal: no key beyond the block's last query contributes
    int k_end = a.Sk;
    if (a.causal && blk_q_max + 1 < k_end) k_end = blk_q_max + 1;
    const int ntile = (k_end + KT - 1) / KT;

    // per-lane constants of the staging: this wave's pieces of a tile
    int st_row[LPW], st_lc[LPW];
#pragma unroll
    for (int i = 0; i < LPW; ++i) {
        const int idx = (wave + i * NW) * 64 + lane;         // 16-B unit index inside the tile
        st_row[i] = idx / CPR;
        st_lc[i] = swz(st_row[i], idx % CPR);
    }
    auto stage = [&](int t, int buf) {
        char* kt_ = smem + buf * STAGE;
        const int k0 = t * KT;
#pragma unroll
        for (int i = 0; i < LPW; ++i) {
            const int slab = wave + i * NW;
            const int key = k0 + st_row[i];
            const bf16_t* pk = zero;
            const bf16_t* pv = zero;
            if (t < ntile && key < a.Sk) {                     // past the last tile: zero page, so every iteration issues LPT glds
                const long long off = kbase + (long long)key * a.km.pos_stride + st_lc[i] * 8;
                pk = a.k + off;
                pv = a.v + off;
            }
            __builtin_amdgcn_global_load_lds(GLB_PTR(pk), LDS_PTR(kt_ + slab * 1024), 16, 0, 0);
            if (!KVSAME) __builtin_amdgcn_global_load_lds(GLB_PTR(pv), LDS_PTR(kt_ + TILE + slab * 1024), 16, 0, 0);
        }
    };
    static_assert(SLABS % NW == 0, "every wave stages the same number of pieces (counted vmcnt)");

    const int g16 = lane >> 4, rr = (lane >> 2) & 3, qq = lane & 3;
    if (ntile > 0) {
        stage(0, 0);
        stage(1, 1);
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LPT) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    // loop-invariant LDS offsets (inside a stage) of every fragment read: the tile loop is unrolled over the three ring slots so
    // that the slot base is an immediate -- per tile the wave issues no address arithmetic at all (the first version spent
    // 20 VALU instructions per MFMA, most of them here; SQ_INSTS_VALU / SQ_INSTS_MFMA)
    uint32_t k_off[2][KS], v_off[2][2][DT][2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        const int row = kt * 32 + (lane & 31);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) k_off[kt][ks] = (uint32_t)(row * ROWB + (swz(row, ks * 2 + h) << 4));
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int r0 = kt * 32 + 16 * s2 + 4 * (g16 >> 1) + rr, r1 = r0 + 8;
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                const int col = d * 32 + 16 * (g16 & 1) + 4 * qq;
                v_off[kt][s2][d][0] = (uint32_t)(r0 * ROWB + (swz(r0, col >> 3) << 4) + (col & 7) * 2);
                v_off[kt][s2][d][1] = (uint32_t)(r1 * ROWB + (swz(r1, col >> 3) << 4) + (col & 7) * 2);
            }
        }
    }
    const uint32_t smem_off = attn_lds_offset(smem);
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int d = 0; d < DT; ++d) { v_off[kt][s2][d][0] += smem_off; v_off[kt][s2][d][1] += smem_off; }

    auto tile_body = [&](auto slot_c, int t) {
        constexpr int SLOT = decltype(slot_c)::value;
        constexpr int KBASE = SLOT * STAGE, VBASE = KVSAME ? KBASE : KBASE + TILE;
        const int k0 = t * KT;
        stage(t + 2, SLOT == 0 ? 2 : SLOT - 1);               // slot of tile t - 1 (every wave is past the barrier behind it)

        // ---- S^T = K Q^T for two 32-key tiles ----
        f32x16_t sacc[2];
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[kt][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(smem + KBASE + k_off[kt][ks]);
                sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sacc[kt], 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        // ---- V^T fragments: requested now, they land under the softmax arithmetic ----
        bf16x4_t vlo[2][2][DT], vhi[2][2][DT];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int d = 0; d < DT; ++d) {
                    vlo[kt][s2][d] = attn_tr16i<VBASE>(v_off[kt][s2][d][0]);
                    vhi[kt][s2][d] = attn_tr16i<VBASE>(v_off[kt][s2][d][1]);
                }
        // ---- masks: only where the tile crosses the end of the keys or (causal) the diagonal of this wave's queries ----
        const bool edge = (k0 + KT > a.Sk) || (a.causal && k0 + KT - 1 > q0);
        if (edge) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = k0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (key >= a.Sk || (a.causal && key > qi)) sacc[kt][r] = -INFINITY;
                }
        }
        // ---- online softmax in the exp2 domain (lane-local + one cross-half exchange) ----
        float tmax;
        {   // 32 -> 1 with three-input maxima (v_max3_f32): 16 instructions instead of 32
            float m3[11];
#pragma unroll
            for (int g = 0; g < 10; ++g) {
                const int e = 3 * g;
                m3[g] = fmaxf(fmaxf(sacc[e >> 4][e & 15], sacc[(e + 1) >> 4][(e + 1) & 15]), sacc[(e + 2) >> 4][(e + 2) & 15]);
            }
            m3[10] = fmaxf(sacc[1][14], sacc[1][15]);
            const float a0 = fmaxf(fmaxf(m3[0], m3[1]), m3[2]), a1 = fmaxf(fmaxf(m3[3], m3[4]), m3[5]);
            const float a2 = fmaxf(fmaxf(m3[6], m3[7]), m3[8]), a3 = fmaxf(m3[9], m3[10]);
            tmax = fmaxf(fmaxf(fmaxf(a0, a1), a2), a3);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);
        if (__builtin_amdgcn_ballot_w64(m_new > m_run) != 0) {          // some lane's maximum moved: rescale (exact, alpha = 1 elsewhere)
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c2);
            l_run *= alpha;
#pragma unroll
            for (int d = 0; d < DT; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
            m_run = m_new;
        }
        const float mc = m_run * c2;
        float psum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[kt][r], c2, -mc));
                sacc[kt][r] = p;
                psum += p;
            }
        psum += __shfl_xor(psum, 32, 64);
        l_run += psum;
        if (a.drop_thr) {                                  // dropout acts on softmax(S): the row sum above is of the undropped weights
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (!attn_drop_keep(drop_key, a.drop_thr, qi, k0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, a.Sk)) sacc[kt][r] = 0.f;
        }

        // ---- O^T += V^T P^T ----
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                u32x4_t pw;
#pragma unroll
                for (int e = 0; e < 4; ++e) pw[e] = pack_bf16x2(sacc[kt][8 * s2 + 2 * e], sacc[kt][8 * s2 + 2 * e + 1]);
                const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, pw);
#pragma unroll
                for (int d = 0; d < DT; ++d) {
                    asm volatile("" : "+v"(vlo[kt][s2][d]), "+v"(vhi[kt][s2][d]));
                    const bf16x8_t vf = __builtin_shufflevector(vlo[kt][s2][d], vhi[kt][s2][d], 0, 1, 2, 3, 4, 5, 6, 7);
                    oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oacc[d], 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_s_setprio(0);
        // tile t + 1 (issued one iteration ago) must have landed; the stage issued above stays in flight
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LPT) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    for (int t = 0; t < ntile; t += 3) {
        tile_body(std::integral_constant<int, 0>{}, t);
        if (t + 1 < ntile) tile_body(std::integral_constant<int, 1>{}, t + 1);
        if (t + 2 < ntile) tile_body(std::integral_constant<int, 2>{}, t + 2);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- epilogue: O / l (+ resid) through LDS (rows_put_f32); lane holds, for its query, d = 32 dt + (r & 3) + 8 (r >> 2) + 4 h ----
    __syncthreads();                                 // every wave is done with the ring: its memory now stages the output rows
    {
        float* fl = reinterpret_cast<float*>(smem) + wave * 32 * DH;
        const float inv = l_run > 0.f ? a.drop_scale / l_run : 0.f;     // (drop_scale = 1 without dropout)
        const int lr = lane & 31;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4_t f;
#pragma unroll
                for (int e = 0; e < 4; ++e) f[e] = oacc[d][4 * g + e] * inv;
                rows_put_f32<DH>(fl, lr, h, d * 4 + g, f);
            }
        const long long obase_s = seq_base(a.om, seq) + head * DH;
        // natural-log LSE of scale * s:  m * scale + ln(l)
        if (a.lse && h == 0 && qi < a.Sq) a.lse[((obase_s - head * DH + (long long)qi * a.om.pos_stride) / a.C) * a.nhead + head] = m_run * a.scale + __logf(l_run);
#pragma unroll
        for (int i = 0; i < CPR / 2; ++i) {
            const int idx = i * 64 + lane, row = idx / CPR, c = idx % CPR;
            if (q0 + row >= a.Sq) continue;
            float f[8];
            rows_get_f32<DH>(fl, row, c, f);
            const long long o = obase_s + (long long)(q0 + row) * a.om.pos_stride + c * 8;
            if (a.oattn) *reinterpret_cast<u32x4_t*>(a.oattn + o) = pack8(f);      // un-residualed output, kept for backward (D = rowsum(dO * O))
            if (a.resid) {
                float r[8];
                unpack8(*reinterpret_cast<const u32x4_t*>(a.resid + o), r);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] += r[e];
            }
            *reinterpret_cast<u32x4_t*>(a.out + o) = pack8(f);
        }
    }
}

// Short self-attention sequences (temporal attention: S = T <= 32 frames of one pixel, attention.py:300-306).  The general
// kernel gives such a sequence a whole 64-key tile and a 32-query wave: at T = 16 one eighth of its MFMA work is useful and
// every block pays the ring prologue for a single tile.  Here 32 / TP sequences are packed into the 32 rows a wave owns
// (TP = 8, 16 or 32 rows per sequence), so a wave's keys are its own rows: S^T = U U^T of those rows -- the A and the B
// operand are the same registers -- masked block-diagonally.  Waves are independent (private LDS rows, no block barrier);
// consecutive waves take the heads of the same rows so that a block reads whole token rows.
template <int TP>
__device__ __forceinline__ bool small_row_valid(int row, long long seq0, int nseq, int S) {
    return (seq0 + row / TP < nseq) && ((row & (TP - 1)) < S);
}

// CROSS (round 6): the keys / values are a CONDITION shared by every sequence of a clip -- the quantised action of the LatentAction decoder
// (action.py:136-160: temporal attention over the pixels of a clip with K, V = Linear(8 -> C) of the (B, T, 8) action codes, attention.py:222-223) --
// addressed through a kv map whose inner stride is 0.  Row r of the wave's K / V tiles is the condition row of (clip of sequence r / TP, frame r % TP);
// S^T = K U^T instead of U U^T, everything else as the packed self-attention form.  Rounds 1-5 sent these calls through the general one-wave kernel
// (1.23 ms per call in the LatentAction step against 0.27 ms for the self-attention layers of the same size).
template <int DH, int TP, bool CROSS>
__global__ void __launch_bounds__(256) attn_small_fwd_kernel(const AttnArgs a) {
    constexpr int ROWB = DH * 2, CPR = DH / 8, WTILE = 32 * ROWB, KS = DH / 16, DT = DH / 32, PCS = WTILE / 1024, SPW = 32 / TP;
    constexpr int WLDS = (CROSS ? 4 : 2) * WTILE;                            // per wave: 32 rows of bf16 in, 32 rows of fp32 out (+ K rows, V rows)
    constexpr int VOFF = CROSS ? 3 * WTILE : 0;
    __shared__ __attribute__((aligned(1024))) char smem[4 * WLDS];
    const int lane = threadIdx.x & 63, h = lane >> 5, lr = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long wg = (long long)blockIdx.x * 4 + wave;
    const int head = (int)(wg % a.nhead);
    const long long seq0 = (wg / a.nhead) * SPW;
    if (seq0 >= a.nseq) return;
    char* lds = smem + wave * WLDS;
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page_a);
#pragma unroll
    for (int i = 0; i < PCS; ++i) {
        const int idx = i * 64 + lane, row = idx / CPR;
        const bf16_t* src = zero;
        const bf16_t* sk = zero;
        const bf16_t* sv = zero;
        if (small_row_valid<TP>(row, seq0, a.nseq, a.Sq)) {
            const int seq = (int)(seq0 + row / TP), pos = row & (TP - 1), c = attn_swz<CPR>(row, idx % CPR) * 8;
            src = (CROSS ? a.q + seq_base(a.qm, seq) + (long long)pos * a.qm.pos_stride : a.k + seq_base(a.km, seq) + (long long)pos * a.km.pos_stride) + head * DH + c;
            if (CROSS) {
                sk = a.k + seq_base(a.km, seq) + (long long)pos * a.km.pos_stride + head * DH + c;
                sv = a.v + seq_base(a.km, seq) + (long long)pos * a.km.pos_stride + head * DH + c;
            }
        }
        __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(lds + i * 1024), 16, 0, 0);
        if (CROSS) {
            __builtin_amdgcn_global_load_lds(GLB_PTR(sk), LDS_PTR(lds + 2 * WTILE + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(GLB_PTR(sv), LDS_PTR(lds + 3 * WTILE + i * 1024), 16, 0, 0);
        }
    }
    const int g16 = lane >> 4, rr = (lane >> 2) & 3, qq = lane & 3;
    const uint32_t lds_off = attn_lds_offset(lds);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    bf16x8_t uf[KS], kf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        uf[ks] = *reinterpret_cast<const bf16x8_t*>(lds + lr * ROWB + (attn_swz<CPR>(lr, ks * 2 + h) << 4));
        kf[ks] = CROSS ? *reinterpret_cast<const bf16x8_t*>(lds + 2 * WTILE + lr * ROWB + (attn_swz<CPR>(lr, ks * 2 + h) << 4)) : uf[ks];
    }
    bf16x4_t vlo[2][DT], vhi[2][DT];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        const int r0 = 16 * s2 + 4 * (g16 >> 1) + rr, r1 = r0 + 8;
#pragma unroll
        for (int d = 0; d < DT; ++d) {
            const int col = d * 32 + 16 * (g16 & 1) + 4 * qq;
            vlo[s2][d] = attn_tr16i<VOFF>(lds_off + (uint32_t)(r0 * ROWB + (attn_swz<CPR>(r0, col >> 3) << 4) + (col & 7) * 2));
            vhi[s2][d] = attn_tr16i<VOFF>(lds_off + (uint32_t)(r1 * ROWB + (attn_swz<CPR>(r1, col >> 3) << 4) + (col & 7) * 2));
        }
    }
    f32x16_t sacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], uf[ks], sacc, 0, 0, 0);      // S^T[key][query]

    const bool q_ok = small_row_valid<TP>(lr, seq0, a.nseq, a.Sq);
    const int qpos = lr & (TP - 1);
    const float c2 = a.scale * 1.4426950408889634f;
    float tmax = -1e30f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int key = (r & 3) + 8 * (r >> 2) + 4 * h, kpos = key & (TP - 1);
        const bool ok = q_ok && ((key ^ lr) & ~(TP - 1)) == 0 && kpos < a.Sq && (!a.causal || kpos <= qpos);
        sacc[r] = ok ? sacc[r] : -INFINITY;
        tmax = fmaxf(tmax, sacc[r]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float mc = tmax * c2;
    float l = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[r], c2, -mc));
        sacc[r] = p;
        l += p;
    }
    l += __shfl_xor(l, 32, 64);

    f32x16_t oacc[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        u32x4_t pw;
#pragma unroll
        for (int e = 0; e < 4; ++e) pw[e] = pack_bf16x2(sacc[8 * s2 + 2 * e], sacc[8 * s2 + 2 * e + 1]);
        const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, pw);
#pragma unroll
        for (int d = 0; d < DT; ++d) {
            asm volatile("" : "+v"(vlo[s2][d]), "+v"(vhi[s2][d]));
            const bf16x8_t vf = __builtin_shufflevector(vlo[s2][d], vhi[s2][d], 0, 1, 2, 3, 4, 5, 6, 7);
            oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oacc[d], 0, 0, 0);
        }
    }
    // O / l goes back through the wave's LDS rows as fp32 so that the global traffic of out / o_attn / resid is whole 16-B chunks,
    // 8 lanes per token row (the MFMA layout would store 8 B per lane into 32 different rows per instruction: measured 2x slower)
    {
        const float inv = l > 0.f ? 1.f / l : 0.f;
        float* fl = reinterpret_cast<float*>(lds);
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4_t f;
#pragma unroll
                for (int e = 0; e < 4; ++e) f[e] = oacc[d][4 * g + e] * inv;
                *reinterpret_cast<f32x4_t*>(fl + lr * DH + (((d * 4 + g) ^ (lr & (CPR - 1))) << 3) + 4 * h) = f;
            }
        if (a.lse && h == 0 && q_ok) {
            const long long tok = (seq_base(a.om, (int)(seq0 + lr / TP)) + (long long)qpos * a.om.pos_stride) / a.C;
            a.lse[tok * a.nhead + head] = tmax * a.scale + __logf(l);
        }
#pragma unroll
        for (int i = 0; i < PCS; ++i) {
            const int idx = i * 64 + lane, row = idx / CPR, c = idx % CPR;
            if (!small_row_valid<TP>(row, seq0, a.nseq, a.Sq)) continue;
            const float* src = fl + row * DH + ((c ^ (row & (CPR - 1))) << 3);
            const f32x4_t f0 = *reinterpret_cast<const f32x4_t*>(src), f1 = *reinterpret_cast<const f32x4_t*>(src + 4);
            float f[8] = {f0[0], f0[1], f0[2], f0[3], f1[0], f1[1], f1[2], f1[3]};
            const long long o = seq_base(a.om, (int)(seq0 + row / TP)) + (long long)(row & (TP - 1)) * a.om.pos_stride + head * DH + c * 8;
            if (a.oattn) *reinterpret_cast<u32x4_t*>(a.oattn + o) = pack8(f);
            if (a.resid) {
                float r[8];
                unpack8(*reinterpret_cast<const u32x4_t*>(a.resid + o), r);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] += r[e];
            }
            *reinterpret_cast<u32x4_t*>(a.out + o) = pack8(f);
        }
    }
}

static SeqMap mk_map(const int64_t* m) {
    SeqMap s;
    s.n_inner = (int)m[0]; s.stride_outer = m[1]; s.stride_inner = m[2]; s.pos_stride = m[3];
    return s;
}

static bool same_map(const SeqMap& x, const SeqMap& y) {
    return x.n_inner == y.n_inner && x.stride_outer == y.stride_outer && x.stride_inner == y.stride_inner && x.pos_stride == y.pos_stride;
}
static int small_attn_mode();
// conditioned short sequences (attn_small_fwd_kernel<.., CROSS> / attn_smallx_bwd_kernel): Sq == Sk <= 32, d_head 32 / 64, K / V rows shared by the
// n_inner sequences of a clip (kv map: inner stride 0), whole packed groups per clip
template <class Args>
static bool small_cond_ok(const Args& a, const void* q, const void* k, const void* v, int d_head) {
    if (q == k || q == v || !small_attn_mode() || a.Sq != a.Sk || a.Sq > 32 || !(d_head == 32 || d_head == 64)) return false;
    const int tp = a.Sq <= 8 ? 8 : (a.Sq <= 16 ? 16 : 32);
    return a.km.stride_inner == 0 && a.km.n_inner >= 1 && a.km.n_inner % (32 / tp) == 0 && a.nseq % a.km.n_inner == 0;
}
static int small_attn_mode() {          // GENIE_ATTN_SMALL=0 sends short sequences through the general kernels (A/B timing, tests)
    static int mode = -1;
    if (mode < 0) { const char* e = getenv("GENIE_ATTN_SMALL"); mode = e ? atoi(e) : 1; }
    return mode;
}

// Host side of attention dropout: probability -> 32-bit threshold, 64-bit seed -> 32-bit call key
struct AttnDrop { unsigned thr, key; float scale; };
static AttnDrop attn_drop_of(float p, uint64_t seed) {
    AttnDrop d{0u, 0u, 1.f};
    if (p > 0.f) {
        const double t = (double)p * 4294967296.0;
        d.thr = t >= 4294967295.0 ? 4294967295u : (t < 1.0 ? 1u : (unsigned)t);
        uint64_t z = seed + 0x9E3779B97F4A7C15ull;                      // splitmix64 finaliser: every seed bit reaches the key
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        d.key = (unsigned)z ^ (unsigned)(z >> 32);
        d.scale = 1.f / (1.f - p);
    }
    return d;
}

static int attention_fwd_impl(const void* q, const void* k, const void* v, const void* resid, void* out, void* o_attn, float* lse, int nseq, int nhead,
                              int d_head, int Sq, int Sk, const int64_t* q_map, const int64_t* kv_map, const int64_t* out_map, float scale,
                              int causal, int out_channels, float dropout_p, uint64_t seed, void* stream) {
    GENIE_CHECK_ARG(q && k && v && out && q_map && kv_map && out_map, "genie_attention_fwd: null pointer");
    GENIE_CHECK_ARG(d_head == 8 || d_head == 16 || d_head == 32 || d_head == 64 || d_head == 128, "genie_attention_fwd: d_head %d not in {8, 16, 32, 64, 128}", d_head);
    GENIE_CHECK_ARG(nseq >= 1 && nhead >= 1 && Sq >= 1 && Sk >= 1, "genie_attention_fwd: empty problem");
    AttnArgs a;
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.resid = (const bf16_t*)resid; a.out = (bf16_t*)out; a.oattn = (bf16_t*)o_attn; a.lse = lse;
    a.qm = mk_map(q_map); a.km = mk_map(kv_map); a.om = mk_map(out_map);
    GENIE_CHECK_ARG(a.qm.n_inner >= 1 && a.km.n_inner >= 1 && a.om.n_inner >= 1, "genie_attention_fwd: bad sequence map");
    a.nseq = nseq; a.nhead = nhead; a.Sq = Sq; a.Sk = Sk; a.scale = scale; a.causal = causal; a.kv_same = (k == v) ? 1 : 0; a.xcd_swizzle = 0;
    a.C = out_channels;
    GENIE_CHECK_ARG(out_channels >= nhead * d_head, "genie_attention_fwd: out_channels %d < nhead * d_head", out_channels);
    GENIE_CHECK_ARG(scale > 0.f, "genie_attention_fwd: scale must be positive (got %g)", (double)scale);
    hipStream_t s = (hipStream_t)stream;
    const bool drop = dropout_p > 0.f;
    if (drop) {                                             // the general kernels only (the packed, conditioned and lean families take no mask)
        GENIE_CHECK_ARG(dropout_p < 1.f, "genie_attention_fwd_dropout: dropout_p %g not in [0, 1)", (double)dropout_p);
        GENIE_CHECK_ARG((long long)Sq * Sk < (1ll << 32), "genie_attention_fwd_dropout: Sq * Sk must fit 32 bits");
        const AttnDrop d = attn_drop_of(dropout_p, seed);
        a.drop_thr = d.thr; a.drop_key = d.key; a.drop_scale = d.scale;
    }
    if (d_head < 32) return genie_attn_narrow_fwd(a, d_head, s);                                     // fp32 VALU kernels (attention_narrow.hip)
    if (!drop && q == k && k == v && Sq == Sk && Sq <= 32 && same_map(a.qm, a.km) && small_attn_mode()) {   // packed short sequences (temporal attention)
        const int tp = Sq <= 8 ? 8 : (Sq <= 16 ? 16 : 32);
        const long long waves = ((long long)nseq + 32 / tp - 1) / (32 / tp) * nhead;
        const unsigned blocks = (unsigned)((waves + 3) / 4);
#define GENIE_ATTN_SMALL(DHv)                                                                            \
    do {                                                                                                 \
        if (tp == 8) attn_small_fwd_kernel<DHv, 8, false><<<blocks, 256, 0, s>>>(a);                     \
        else if (tp == 16) attn_small_fwd_kernel<DHv, 16, false><<<blocks, 256, 0, s>>>(a);              \
        else attn_small_fwd_kernel<DHv, 32, false><<<blocks, 256, 0, s>>>(a);                            \
    } while (0)
        if (d_head == 32) GENIE_ATTN_SMALL(32); else if (d_head == 64) GENIE_ATTN_SMALL(64); else GENIE_ATTN_SMALL(128);
#undef GENIE_ATTN_SMALL
        GENIE_CHECK_LAUNCH();
        return GENIE_OK;
    }
    if (!drop && small_cond_ok(a, q, k, v, d_head)) {      // packed short sequences against a per-clip condition (kv map with inner stride 0)
        const int tp = Sq <= 8 ? 8 : (Sq <= 16 ? 16 : 32);
        const long long waves = ((long long)nseq / (32 / tp)) * nhead;
        const unsigned blocks = (unsigned)((waves + 3) / 4);
#define GENIE_ATTN_SMALLX(DHv)                                                                           \
    do {                                                                                                 \
        if (tp == 8) attn_small_fwd_kernel<DHv, 8, true><<<blocks, 256, 0, s>>>(a);                      \
        else if (tp == 16) attn_small_fwd_kernel<DHv, 16, true><<<blocks, 256, 0, s>>>(a);               \
        else attn_small_fwd_kernel<DHv, 32, true><<<blocks, 256, 0, s>>>(a);                             \
    } while (0)
        if (d_head == 32) GENIE_ATTN_SMALLX(32); else GENIE_ATTN_SMALLX(64);
#undef GENIE_ATTN_SMALLX
        GENIE_CHECK_LAUNCH();
        return GENIE_OK;
    }
    if (!drop && genie_attn_lean_fwd_ok(a, d_head)) return genie_attn_lean_fwd(a, s);        // d_head 64, four waves per SIMD (attention_lean.hip)
    int nw = (Sq + 31) / 32;
    nw = nw >= 3 ? 4 : nw;                                        // 1, 2 or 4 waves (every wave stages the same number of pieces)
    const int qtiles = (Sq + 32 * nw - 1) / (32 * nw);
    GENIE_CHECK_ARG((long long)nseq * qtiles < (1ll << 31) && nhead <= 65535, "genie_attention_fwd: grid too large");
    const int tile = 64 * d_head * 2;
    int lds = 3 * (a.kv_same ? tile : 2 * tile);
    if (lds < nw * 32 * d_head * 4) lds = nw * 32 * d_head * 4;      // the epilogue stages NW x 32 fp32 rows in the ring's memory
    dim3 grid((unsigned)(nseq * qtiles), nhead, 1);
#define GENIE_ATTN_FWD(DHv, NWv)                                                                         \
    do {                                                                                                 \
        auto kf_ = a.kv_same ? attn_fwd_kernel<DHv, NWv, true> : attn_fwd_kernel<DHv, NWv, false>;        \
        if (lds > 65536) GENIE_CHECK_ARG(hipFuncSetAttribute((const void*)kf_, hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess, "hipFuncSetAttribute failed"); \
        kf_<<<grid, 64 * NWv, lds, s>>>(a);                                                              \
    } while (0)
#define GENIE_ATTN_FWD_NW(DHv)                                                                           \
    do {                                                                                                 \
        if (nw == 1) GENIE_ATTN_FWD(DHv, 1); else if (nw == 2) GENIE_ATTN_FWD(DHv, 2); else GENIE_ATTN_FWD(DHv, 4); \
    } while (0)
    if (d_head == 32) GENIE_ATTN_FWD_NW(32);
    else if (d_head == 64) GENIE_ATTN_FWD_NW(64);
    else GENIE_ATTN_FWD_NW(128);
#undef GENIE_ATTN_FWD_NW
#undef GENIE_ATTN_FWD
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}

extern "C" int genie_attention_fwd(const void* q, const void* k, const void* v, const void* resid, void* out, void* o_attn, float* lse, int nseq, int nhead,
                                   int d_head, int Sq, int Sk, const int64_t* q_map, const int64_t* kv_map, const int64_t* out_map, float scale,
                                   int causal, int out_channels, void* stream) {
    return attention_fwd_impl(q, k, v, resid, out, o_attn, lse, nseq, nhead, d_head, Sq, Sk, q_map, kv_map, out_map, scale, causal, out_channels, 0.f, 0, stream);
}

// genie_attention_fwd with dropout on the attention weights (reference attention.py:225-230: `dropout_p=self.dropout`): out = (softmax(S) o M / (1 - p)) V
// with M a pure function of (seed, sequence, head, query, key) -- attn_args.h.  lse is the softmax's (no dropout in it); o_attn is the dropped output.
extern "C" int genie_attention_fwd_dropout(const void* q, const void* k, const void* v, const void* resid, void* out, void* o_attn, float* lse, int nseq,
                                           int nhead, int d_head, int Sq, int Sk, const int64_t* q_map, const int64_t* kv_map, const int64_t* out_map,
                                           float scale, int causal, int out_channels, float dropout_p, uint64_t seed, void* stream) {
    GENIE_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, "genie_attention_fwd_dropout: dropout_p %g not in [0, 1)", (double)dropout_p);
    return attention_fwd_impl(q, k, v, resid, out, o_attn, lse, nseq, nhead, d_head, Sq, Sk, q_map, kv_map, out_map, scale, causal, out_channels, dropout_p, seed, stream);
}

// The keep decisions of the two dropout entry points, written out: keep[((seq * nhead + head) * Sq + q) * Sk + k] = 1 | 0.  For tests and debugging (the
// kernels never store a mask).
__global__ void __launch_bounds__(256) attn_dropout_mask_kernel(uint8_t* __restrict__ keep, int nseq, int nhead, int Sq, int Sk, unsigned thr, unsigned key) {
    const long long n = (long long)nseq * nhead * Sq * Sk;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int k = (int)(i % Sk), q = (int)((i / Sk) % Sq);
        const long long sh = i / ((long long)Sq * Sk);
        keep[i] = attn_drop_keep(attn_drop_seqkey(key, (int)(sh / nhead), nhead, (int)(sh % nhead)), thr, q, k, Sk) ? 1 : 0;
    }
}

extern "C" int genie_attention_dropout_mask(uint8_t* keep, int nseq, int nhead, int Sq, int Sk, float dropout_p, uint64_t seed, void* stream) {
    GENIE_CHECK_ARG(keep && nseq >= 1 && nhead >= 1 && Sq >= 1 && Sk >= 1, "genie_attention_dropout_mask: null pointer / empty problem");
    GENIE_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, "genie_attention_dropout_mask: dropout_p %g not in [0, 1)", (double)dropout_p);
    GENIE_CHECK_ARG((long long)Sq * Sk < (1ll << 32), "genie_attention_dropout_mask: Sq * Sk must fit 32 bits");
    const AttnDrop d = attn_drop_of(dropout_p, seed);
    const long long n = (long long)nseq * nhead * Sq * Sk;
    const unsigned blocks = (unsigned)((n + 255) / 256 > 65536 ? 65536 : (n + 255) / 256);
    attn_dropout_mask_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(keep, nseq, nhead, Sq, Sk, d.thr, d.key);
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}


// ------------------------------------------------------------------------------------------------
// backward
//   D[token][head] = sum_d dO * (out - resid)                                   (preprocess)
//   dQ kernel  : per query tile, loop over key tiles   (lane = query, as forward)
//   dKV kernel : per key tile, loop over query tiles   (lane = key)
// ------------------------------------------------------------------------------------------------
template <int DH>
__global__ void __launch_bounds__(256) attn_bwd_prep_kernel(const bf16_t* __restrict__ dO, const bf16_t* __restrict__ out,
                                                            const bf16_t* __restrict__ resid, float* __restrict__ D, long long ntok, int C,
                                                            int nhead, const float* __restrict__ lse, float* __restrict__ lse2, float* __restrict__ negD) {
    const int lane = threadIdx.x & 63;
    const long long tok = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= ntok) return;
    constexpr int LPH = DH / 8;                   // lanes per head
    const int nch = (nhead * DH) >> 3;
    for (int ch = lane; ch < ((nch + 63) / 64) * 64; ch += 64) {
        float acc = 0.f;
        if (ch < nch) {
            float a[8], b[8];
            unpack8(*reinterpret_cast<const u32x4_t*>(dO + tok * C + ch * 8), a);
            unpack8(*reinterpret_cast<const u32x4_t*>(out + tok * C + ch * 8), b);
            if (resid) {
                float r[8];
                unpack8(*reinterpret_cast<const u32x4_t*>(resid + tok * C + ch * 8), r);
#pragma unroll
                for (int j = 0; j < 8; ++j) b[j] -= r[j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += a[j] * b[j];
        }
#pragma unroll
        for (int o = 1; o < LPH; o <<= 1) acc += __shfl_xor(acc, o, 64);
        if (ch < nch && (ch % LPH) == 0) {
            D[tok * nhead + ch / LPH] = acc;
            lse2[tok * nhead + ch / LPH] = lse[tok * nhead + ch / LPH] * 1.4426950408889634f;      // the exp2-domain form the backward kernels subtract
            negD[tok * nhead + ch / LPH] = -acc;                                                  // accumulator input of the lean kernels' dP products
        }
    }
}

// Both backward kernels share the forward's structure: 64-row tiles in a ring of three LDS stages DMA'd two tiles ahead with
// counted waits, exp2-domain probabilities p = exp2(c s - lse log2 e), masks only on edge tiles, transposing LDS reads (asm)
// requested before the element-wise phase.  The softmax scale of dS is applied ONCE to the dQ / dK accumulators at the end.
template <int DH, int NW, bool KVSAME>
__global__ void __launch_bounds__(64 * NW) attn_bwd_dq_kernel(const AttnBwdArgs a) {
    constexpr int KT = 64, ROWB = DH * 2, CPR = DH / 8, TILE = KT * ROWB, KS = DH / 16, DT = DH / 32;
    constexpr int SLABS = TILE / 1024, LPW = SLABS / NW, LPT = KVSAME ? LPW : 2 * LPW;
    constexpr int STAGE = KVSAME ? TILE : 2 * TILE;
    static_assert(SLABS % NW == 0, "every wave stages the same number of pieces (counted vmcnt)");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qtiles = (a.Sq + 32 * NW - 1) / (32 * NW);
    const int seq = blockIdx.x / qtiles, qtile = blockIdx.x % qtiles, head = blockIdx.y;
    const int q0 = qtile * (32 * NW) + wave * 32;
    const int qi = q0 + (lane & 31), h = lane >> 5;
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page_a);
    auto swz = [&](int row, int chunk) -> int { return attn_swz<CPR>(row, chunk); };
    bf16x8_t qf[KS], dof[KS];
    float lse2 = 0.f, D_q = 0.f;                  // lse * log2(e)
    {
        const bool ok = qi < a.Sq;
        const long long qoff = seq_base(a.qm, seq) + (long long)(ok ? qi : 0) * a.qm.pos_stride;
        const long long ooff = seq_base(a.om, seq) + (long long)(ok ? qi : 0) * a.om.pos_stride;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ok) {
                qf[ks] = *reinterpret_cast<const bf16x8_t*>(a.q + qoff + head * DH + ks * 16 + h * 8);
                dof[ks] = *reinterpret_cast<const bf16x8_t*>(a.dO + ooff + head * DH + ks * 16 + h * 8);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) { qf[ks][e] = 0; dof[ks][e] = 0; }
            }
        }
        if (ok) {
            const long long tok = ooff / a.C;
            lse2 = a.lse[tok * a.nhead + head] * 1.4426950408889634f;
            D_q = a.D[tok * a.nhead + head];
        }
    }
    f32x16_t dq[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[d][r] = 0.f;
    const float c2 = a.scale * 1.4426950408889634f;
    const unsigned drop_key = attn_drop_seqkey(a.drop_key, seq, a.nhead, head);
    const long long kbase = seq_base(a.km, seq) + head * DH;
    const int blk_q_max = qtile * (32 * NW) + 32 * NW - 1;
    int k_end = a.Sk;
    if (a.causal && blk_q_max + 1 < k_end) k_end = blk_q_max + 1;
    const int ntile = (k_end + KT - 1) / KT;
    const int g16 = lane >> 4, rr = (lane >> 2) & 3, qq = lane & 3;

    int st_row[LPW], st_lc[LPW];
#pragma unroll
    for (int i = 0; i < LPW; ++i) {
        const int idx = (wave + i * NW) * 64 + lane;
        st_row[i] = idx / CPR;
        st_lc[i] = swz(st_row[i], idx % CPR);
    }
    auto stage = [&](int t, int buf) {
        char* kt_ = smem + buf * STAGE;
        const int k0 = t * KT;
#pragma unroll
        for (int i = 0; i < LPW; ++i) {
            const int slab = wave + i * NW;
            const int key = k0 + st_row[i];
            const bf16_t* pk = zero;
            const bf16_t* pv = zero;
            if (t < ntile && key < a.Sk) {
                const long long off = kbase + (long long)key * a.km.pos_stride + st_lc[i] * 8;
                pk = a.k + off;
                pv = a.v + off;
            }
            __builtin_amdgcn_global_load_lds(GLB_PTR(pk), LDS_PTR(kt_ + slab * 1024), 16, 0, 0);
            if (!KVSAME) __builtin_amdgcn_global_load_lds(GLB_PTR(pv), LDS_PTR(kt_ + TILE + slab * 1024), 16, 0, 0);
        }
    };

    if (ntile > 0) {
        stage(0, 0);
        stage(1, 1);
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LPT) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    // loop-invariant fragment offsets; the tile loop is unrolled over the three ring slots so that the slot base is an immediate
    uint32_t k_off[2][KS], t_off[2][2][DT][2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        const int row = kt * 32 + (lane & 31);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) k_off[kt][ks] = (uint32_t)(row * ROWB + (swz(row, ks * 2 + h) << 4));
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int r0 = kt * 32 + 16 * s2 + 4 * (g16 >> 1) + rr, r1 = r0 + 8;
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                const int col = d * 32 + 16 * (g16 & 1) + 4 * qq;
                t_off[kt][s2][d][0] = attn_lds_offset(smem) + (uint32_t)(r0 * ROWB + (swz(r0, col >> 3) << 4) + (col & 7) * 2);
                t_off[kt][s2][d][1] = attn_lds_offset(smem) + (uint32_t)(r1 * ROWB + (swz(r1, col >> 3) << 4) + (col & 7) * 2);
            }
        }
    }
    auto tile_body = [&](auto slot_c, int t) {
        constexpr int SLOT = decltype(slot_c)::value;
        constexpr int KBASE = SLOT * STAGE, VBASE = KVSAME ? KBASE : KBASE + TILE;
        const int k0 = t * KT;
        stage(t + 2, SLOT == 0 ? 2 : SLOT - 1);
        f32x16_t sacc[2], pacc[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { sacc[kt][r] = 0.f; pacc[kt][r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(smem + KBASE + k_off[kt][ks]);
                sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sacc[kt], 0, 0, 0);      // S^T
                if (KVSAME) {
                    pacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, dof[ks], pacc[kt], 0, 0, 0); // dP^T (V == K)
                } else {
                    const bf16x8_t vf = *reinterpret_cast<const bf16x8_t*>(smem + VBASE + k_off[kt][ks]);
                    pacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[ks], pacc[kt], 0, 0, 0);
                }
            }
        }
        // K^T fragments for dQ^T += K^T dS^T, requested before the element-wise phase
        bf16x4_t klo[2][2][DT], khi[2][2][DT];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int d = 0; d < DT; ++d) {
                    klo[kt][s2][d] = attn_tr16i<KBASE>(t_off[kt][s2][d][0]);
                    khi[kt][s2][d] = attn_tr16i<KBASE>(t_off[kt][s2][d][1]);
                }
        const bool edge = (k0 + KT > a.Sk) || (a.causal && k0 + KT - 1 > q0) || (q0 + 32 > a.Sq);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[kt][r], c2, -lse2));
                if (edge) {
                    const int key = k0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (key >= a.Sk || (a.causal && key > qi) || qi >= a.Sq) p = 0.f;
                }
                float dp = pacc[kt][r];
                if (a.drop_thr)                             // dP reaches the softmax through the same mask and 1 / (1 - p) as the forward's weights
                    dp = attn_drop_keep(drop_key, a.drop_thr, qi, k0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, a.Sk) ? dp * a.drop_scale : 0.f;
                sacc[kt][r] = p * (dp - D_q);                                                           // dS^T / scale
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                u32x4_t dw;
#pragma unroll
                for (int e = 0; e < 4; ++e) dw[e] = pack_bf16x2(sacc[kt][8 * s2 + 2 * e], sacc[kt][8 * s2 + 2 * e + 1]);
                const bf16x8_t df = __builtin_bit_cast(bf16x8_t, dw);
#pragma unroll
                for (int d = 0; d < DT; ++d) {
                    asm volatile("" : "+v"(klo[kt][s2][d]), "+v"(khi[kt][s2][d]));
                    const bf16x8_t kT = __builtin_shufflevector(klo[kt][s2][d], khi[kt][s2][d], 0, 1, 2, 3, 4, 5, 6, 7);
                    dq[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kT, df, dq[d], 0, 0, 0);           // dQ^T += K^T dS^T
                }
            }
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LPT) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    for (int t = 0; t < ntile; t += 3) {
        tile_body(std::integral_constant<int, 0>{}, t);
        if (t + 1 < ntile) tile_body(std::integral_constant<int, 1>{}, t + 1);
        if (t + 2 < ntile) tile_body(std::integral_constant<int, 2>{}, t + 2);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                 // ring memory becomes the staging area of the dQ rows (bf16: nothing is added later)
    {
        char* wl = smem + wave * 32 * ROWB;
        const int lr = lane & 31;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2_t ov;
                ov[0] = pack_bf16x2(dq[d][4 * g] * a.scale, dq[d][4 * g + 1] * a.scale);
                ov[1] = pack_bf16x2(dq[d][4 * g + 2] * a.scale, dq[d][4 * g + 3] * a.scale);
                *reinterpret_cast<u32x2_t*>(wl + lr * ROWB + (swz(lr, d * 4 + g) << 4) + 8 * h) = ov;
            }
        const long long obase_s = seq_base(a.qm, seq) + head * DH;
#pragma unroll
        for (int i = 0; i < CPR / 2; ++i) {
            const int idx = i * 64 + lane, row = idx / CPR;
            if (q0 + row >= a.Sq) continue;
            *reinterpret_cast<u32x4_t*>(a.dq + obase_s + (long long)(q0 + row) * a.qm.pos_stride + swz(row, idx % CPR) * 8) =
                *reinterpret_cast<const u32x4_t*>(wl + idx * 16);
        }
    }
}

template <int DH, int NW>
__global__ void __launch_bounds__(64 * NW) attn_bwd_dkv_kernel(const AttnBwdArgs a) {
    constexpr int QT = 64, ROWB = DH * 2, CPR = DH / 8, TILE = QT * ROWB, KS = DH / 16, DT = DH / 32;
    constexpr int SLABS = TILE / 1024, LPW = SLABS / NW, LPT = 2 * LPW + 2;
    constexpr int STAGE = 2 * TILE + 512;            // Q tile | dO tile | lse log2e [64] | D [64]
    static_assert(SLABS % NW == 0, "every wave stages the same number of pieces (counted vmcnt)");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ktiles = (a.Sk + 32 * NW - 1) / (32 * NW);
    const int seq = blockIdx.x / ktiles, ktile_i = blockIdx.x % ktiles, head = blockIdx.y;
    const int key0 = ktile_i * (32 * NW) + wave * 32;
    const int ki = key0 + (lane & 31), h = lane >> 5;
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page_a);
    auto swz = [&](int row, int chunk) -> int { return attn_swz<CPR>(row, chunk); };
    bf16x8_t kf[KS], vf[KS];
    {
        const bool ok = ki < a.Sk;
        const long long off = seq_base(a.km, seq) + (long long)(ok ? ki : 0) * a.km.pos_stride + head * DH;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ok) {
                kf[ks] = *reinterpret_cast<const bf16x8_t*>(a.k + off + ks * 16 + h * 8);
                vf[ks] = *reinterpret_cast<const bf16x8_t*>(a.v + off + ks * 16 + h * 8);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) { kf[ks][e] = 0; vf[ks][e] = 0; }
            }
        }
    }
    f32x16_t dk[DT], dv[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[d][r] = 0.f; dv[d][r] = 0.f; }
    const float c2 = a.scale * 1.4426950408889634f;
    const unsigned drop_key = attn_drop_seqkey(a.drop_key, seq, a.nhead, head);
    const long long qbase = seq_base(a.qm, seq) + head * DH;
    const long long obase_s = seq_base(a.om, seq) + head * DH;
    const long long otok_s = seq_base(a.om, seq);
    const int blk_key_min = ktile_i * (32 * NW);
    const int q_begin = a.causal ? (blk_key_min / QT) * QT : 0;      // queries before the first key of the block see none of its keys
    const int ntile = a.Sq > q_begin ? (a.Sq - q_begin + QT - 1) / QT : 0;
    const int g16 = lane >> 4, rr = (lane >> 2) & 3, qq = lane & 3;

    int st_row[LPW], st_lc[LPW];
#pragma unroll
    for (int i = 0; i < LPW; ++i) {
        const int idx = (wave + i * NW) * 64 + lane;
        st_row[i] = idx / CPR;
        st_lc[i] = swz(st_row[i], idx % CPR);
    }
    auto stage = [&](int t, int buf) {
        char* qt_ = smem + buf * STAGE;
        const int qs = q_begin + t * QT;
#pragma unroll
        for (int i = 0; i < LPW; ++i) {
            const int slab = wave + i * NW;
            const int qrow = qs + st_row[i];
            const bf16_t* pq = zero;
            const bf16_t* pd = zero;
            if (t < ntile && qrow < a.Sq) {
                pq = a.q + qbase + (long long)qrow * a.qm.pos_stride + st_lc[i] * 8;
                pd = a.dO + obase_s + (long long)qrow * a.om.pos_stride + st_lc[i] * 8;
            }
            __builtin_amdgcn_global_load_lds(GLB_PTR(pq), LDS_PTR(qt_ + slab * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(GLB_PTR(pd), LDS_PTR(qt_ + TILE + slab * 1024), 16, 0, 0);
        }
        // lse / D of the tile's 64 queries: 4 B per lane, every wave writes the same 256 B (same data)
        {
            const int qrow = qs + lane;
            const float* pl = reinterpret_cast<const float*>(zero);
            const float* pd = reinterpret_cast<const float*>(zero);
            if (t < ntile && qrow < a.Sq) {
                const long long tok = (otok_s + (long long)qrow * a.om.pos_stride) / a.C;
                pl = a.lse + tok * a.nhead + head;
                pd = a.D + tok * a.nhead + head;
            }
            __builtin_amdgcn_global_load_lds(GLB_PTR(pl), LDS_PTR(qt_ + 2 * TILE), 4, 0, 0);
            __builtin_amdgcn_global_load_lds(GLB_PTR(pd), LDS_PTR(qt_ + 2 * TILE + 256), 4, 0, 0);
        }
    };

    if (ntile > 0) {
        stage(0, 0);
        stage(1, 1);
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LPT) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    // loop-invariant fragment offsets; the tile loop is unrolled over the three ring slots so that the slot base is an immediate
    uint32_t r_off[2][KS], t_off[2][2][DT][2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int row = qt * 32 + (lane & 31);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) r_off[qt][ks] = (uint32_t)(row * ROWB + (swz(row, ks * 2 + h) << 4));
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int r0 = qt * 32 + 16 * s2 + 4 * (g16 >> 1) + rr, r1 = r0 + 8;
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                const int col = d * 32 + 16 * (g16 & 1) + 4 * qq;
                t_off[qt][s2][d][0] = attn_lds_offset(smem) + (uint32_t)(r0 * ROWB + (swz(r0, col >> 3) << 4) + (col & 7) * 2);
                t_off[qt][s2][d][1] = attn_lds_offset(smem) + (uint32_t)(r1 * ROWB + (swz(r1, col >> 3) << 4) + (col & 7) * 2);
            }
        }
    }
    auto tile_body = [&](auto slot_c, int t) {
        constexpr int SLOT = decltype(slot_c)::value;
        constexpr int QBASE = SLOT * STAGE, OBASE = QBASE + TILE;
        const int qs = q_begin + t * QT;
        stage(t + 2, SLOT == 0 ? 2 : SLOT - 1);
        const float* lse_l = reinterpret_cast<const float*>(smem + QBASE + 2 * TILE);
        const float* D_l = lse_l + 64;
        const bool edge = (qs + QT > a.Sq) || (key0 + 32 > a.Sk) || (a.causal && key0 + 31 > qs);
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            // dO^T / Q^T fragments of this 32-query half for dV^T += dO^T P and dK^T += Q^T dS, requested first: they land
            // under the S / dP products and the element-wise phase
            bf16x4_t dlo[2][DT], dhi[2][DT], qlo[2][DT], qhi[2][DT];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int d = 0; d < DT; ++d) {
                    dlo[s2][d] = attn_tr16i<OBASE>(t_off[qt][s2][d][0]); dhi[s2][d] = attn_tr16i<OBASE>(t_off[qt][s2][d][1]);
                    qlo[s2][d] = attn_tr16i<QBASE>(t_off[qt][s2][d][0]); qhi[s2][d] = attn_tr16i<QBASE>(t_off[qt][s2][d][1]);
                }
            f32x16_t sacc, pacc, ds;
#pragma unroll
            for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; pacc[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8_t qfr = *reinterpret_cast<const bf16x8_t*>(smem + QBASE + r_off[qt][ks]);
                const bf16x8_t dofr = *reinterpret_cast<const bf16x8_t*>(smem + OBASE + r_off[qt][ks]);
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qfr, kf[ks], sacc, 0, 0, 0);            // S[q][key]
                pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dofr, vf[ks], pacc, 0, 0, 0);           // dP[q][key]
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ql0 = qt * 32 + 8 * g + 4 * h;                                               // rows 4 g .. 4 g + 3 of this lane
                const f32x4_t l4 = *reinterpret_cast<const f32x4_t*>(lse_l + ql0);
                const f32x4_t d4 = *reinterpret_cast<const f32x4_t*>(D_l + ql0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g + e;
                    float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[r], c2, -l4[e] * 1.4426950408889634f));
                    if (edge) {
                        const int qg = qs + ql0 + e;
                        if (qg >= a.Sq || ki >= a.Sk || (a.causal && ki > qg)) p = 0.f;
                    }
                    float pd = p, dp = pacc[r];
                    if (a.drop_thr) {
                        const bool keep = attn_drop_keep(drop_key, a.drop_thr, qs + ql0 + e, ki, a.Sk);
                        pd = keep ? p * a.drop_scale : 0.f;
                        dp = keep ? dp * a.drop_scale : 0.f;
                    }
                    sacc[r] = pd;                                                                       // dropped weights: dV^T += dO^T P
                    ds[r] = p * (dp - d4[e]);                                                           // dS / scale
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                u32x4_t pw, dw;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    pw[e] = pack_bf16x2(sacc[8 * s2 + 2 * e], sacc[8 * s2 + 2 * e + 1]);
                    dw[e] = pack_bf16x2(ds[8 * s2 + 2 * e], ds[8 * s2 + 2 * e + 1]);
                }
                const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, pw), df = __builtin_bit_cast(bf16x8_t, dw);
#pragma unroll
                for (int d = 0; d < DT; ++d) {
                    asm volatile("" : "+v"(dlo[s2][d]), "+v"(dhi[s2][d]), "+v"(qlo[s2][d]), "+v"(qhi[s2][d]));
                    const bf16x8_t doT = __builtin_shufflevector(dlo[s2][d], dhi[s2][d], 0, 1, 2, 3, 4, 5, 6, 7);
                    const bf16x8_t qT = __builtin_shufflevector(qlo[s2][d], qhi[s2][d], 0, 1, 2, 3, 4, 5, 6, 7);
                    dv[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(doT, pf, dv[d], 0, 0, 0);           // dV^T += dO^T P
                    dk[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qT, df, dk[d], 0, 0, 0);            // dK^T += Q^T dS
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LPT) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    for (int t = 0; t < ntile; t += 3) {
        tile_body(std::integral_constant<int, 0>{}, t);
        if (t + 1 < ntile) tile_body(std::integral_constant<int, 1>{}, t + 1);
        if (t + 2 < ntile) tile_body(std::integral_constant<int, 2>{}, t + 2);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                 // ring memory becomes the fp32 staging area of the dK / dV rows
    {
        float* fl = reinterpret_cast<float*>(smem) + wave * 32 * DH;
        const int lr = lane & 31;
        const long long kb_s = seq_base(a.dkm, seq) + head * DH;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4_t f;
#pragma unroll
                for (int e = 0; e < 4; ++e) f[e] = a.fuse_self ? __builtin_fmaf(dk[d][4 * g + e], a.scale, dv[d][4 * g + e]) : dk[d][4 * g + e] * a.scale;
                rows_put_f32<DH>(fl, lr, h, d * 4 + g, f);
            }
#pragma unroll
        for (int i = 0; i < CPR / 2; ++i) {
            const int idx = i * 64 + lane, row = idx / CPR, c = idx % CPR;
            if (key0 + row >= a.Sk) continue;
            float f[8];
            rows_get_f32<DH>(fl, row, c, f);
            const long long o = kb_s + (long long)(key0 + row) * a.dkm.pos_stride + c * 8;
            if (a.fuse_self) {                       // dk row += dq_in row: the buffer then holds dQ + dK + dV
                float r[8];
                unpack8(*reinterpret_cast<const u32x4_t*>(a.dq_in + o), r);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] += r[e];
            }
            *reinterpret_cast<u32x4_t*>(a.dk + o) = pack8(f);
        }
        if (!a.fuse_self) {
#pragma unroll
            for (int d = 0; d < DT; ++d)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4_t f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) f[e] = dv[d][4 * g + e];
                    rows_put_f32<DH>(fl, lr, h, d * 4 + g, f);
                }
#pragma unroll
            for (int i = 0; i < CPR / 2; ++i) {
                const int idx = i * 64 + lane, row = idx / CPR, c = idx % CPR;
                if (key0 + row >= a.Sk) continue;
                float f[8];
                rows_get_f32<DH>(fl, row, c, f);
                *reinterpret_cast<u32x4_t*>(a.dv + kb_s + (long long)(key0 + row) * a.dkm.pos_stride + c * 8) = pack8(f);
            }
        }
    }
}

// Backward for the packed short sequences (see attn_small_fwd_kernel): everything a sequence needs is in the wave's own 32 rows,
// so one kernel produces du = dQ + dK + dV.  S = U U^T is symmetric, so one product serves both orientations; dP is needed in
// both (dP^T = U dO^T with lane = query for dQ, dP = dO U^T with lane = key for dK / dV).
template <int DH, int TP>
__global__ void __launch_bounds__(128) attn_small_bwd_kernel(const AttnBwdArgs a) {
    constexpr int ROWB = DH * 2, CPR = DH / 8, WTILE = 32 * ROWB, KS = DH / 16, DT = DH / 32, PCS = WTILE / 1024, SPW = 32 / TP;
    constexpr int WSTAGE = 2 * WTILE + 256;          // U rows | dO rows | lse [32] | D [32]
    __shared__ __attribute__((aligned(1024))) char smem[2 * WSTAGE];
    const int lane = threadIdx.x & 63, h = lane >> 5, lr = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long wg = (long long)blockIdx.x * 2 + wave;
    const int head = (int)(wg % a.nhead);
    const long long seq0 = (wg / a.nhead) * SPW;
    if (seq0 >= a.nseq) return;
    char* lds = smem + wave * WSTAGE;
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page_a);
#pragma unroll
    for (int i = 0; i < PCS; ++i) {
        const int idx = i * 64 + lane, row = idx / CPR;
        const bf16_t* su = zero;
        const bf16_t* sd = zero;
        if (small_row_valid<TP>(row, seq0, a.nseq, a.Sq)) {
            const int seq = (int)(seq0 + row / TP), pos = row & (TP - 1), c = attn_swz<CPR>(row, idx % CPR) * 8;
            su = a.q + seq_base(a.qm, seq) + (long long)pos * a.qm.pos_stride + head * DH + c;
            sd = a.dO + seq_base(a.om, seq) + (long long)pos * a.om.pos_stride + head * DH + c;
        }
        __builtin_amdgcn_global_load_lds(GLB_PTR(su), LDS_PTR(lds + i * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(GLB_PTR(sd), LDS_PTR(lds + WTILE + i * 1024), 16, 0, 0);
    }
    const bool my_ok = small_row_valid<TP>(lr, seq0, a.nseq, a.Sq);
    const int mypos = lr & (TP - 1);
    {   // lanes 0..31: lse of row lr, lanes 32..63: D of row lr
        const float* src = reinterpret_cast<const float*>(zero);
        if (my_ok) {
            const long long tok = (seq_base(a.om, (int)(seq0 + lr / TP)) + (long long)mypos * a.om.pos_stride) / a.C;
            src = (h ? a.D : a.lse) + tok * a.nhead + head;
        }
        __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(lds + 2 * WTILE), 4, 0, 0);
    }
    const int g16 = lane >> 4, rr = (lane >> 2) & 3, qq = lane & 3;
    const uint32_t lds_off = attn_lds_offset(lds);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    bf16x8_t uf[KS], df[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int o = lr * ROWB + (attn_swz<CPR>(lr, ks * 2 + h) << 4);
        uf[ks] = *reinterpret_cast<const bf16x8_t*>(lds + o);
        df[ks] = *reinterpret_cast<const bf16x8_t*>(lds + WTILE + o);
    }
    const float* lse_l = reinterpret_cast<const float*>(lds + 2 * WTILE);
    const float* D_l = lse_l + 32;
    const float my_lse2 = lse_l[lr] * 1.4426950408889634f, my_D = D_l[lr];
    bf16x4_t ulo[2][DT], uhi[2][DT], dlo[2][DT], dhi[2][DT];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        const int r0 = 16 * s2 + 4 * (g16 >> 1) + rr, r1 = r0 + 8;
#pragma unroll
        for (int d = 0; d < DT; ++d) {
            const int col = d * 32 + 16 * (g16 & 1) + 4 * qq;
            const uint32_t o0 = lds_off + (uint32_t)(r0 * ROWB + (attn_swz<CPR>(r0, col >> 3) << 4) + (col & 7) * 2);
            const uint32_t o1 = lds_off + (uint32_t)(r1 * ROWB + (attn_swz<CPR>(r1, col >> 3) << 4) + (col & 7) * 2);
            ulo[s2][d] = attn_tr16(o0); uhi[s2][d] = attn_tr16(o1);
            dlo[s2][d] = attn_tr16i<WTILE>(o0); dhi[s2][d] = attn_tr16i<WTILE>(o1);
        }
    }
    f32x16_t sacc, pt, pn;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; pt[r] = 0.f; pn[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(uf[ks], uf[ks], sacc, 0, 0, 0);      // S (symmetric)
        pt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(uf[ks], df[ks], pt, 0, 0, 0);          // dP^T[key][query]
        pn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(df[ks], uf[ks], pn, 0, 0, 0);          // dP[query][key]
    }
    const float c2 = a.scale * 1.4426950408889634f;
    f32x16_t dsa, dsb;               // sacc is reused for P (lane = key orientation)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4_t l4 = *reinterpret_cast<const f32x4_t*>(lse_l + 8 * g + 4 * h);
        const f32x4_t d4 = *reinterpret_cast<const f32x4_t*>(D_l + 8 * g + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int r = 4 * g + e, row = 8 * g + 4 * h + e, rpos = row & (TP - 1);
            const bool pair = my_ok && ((row ^ lr) & ~(TP - 1)) == 0 && rpos < a.Sq;
            const bool ok_a = pair && (!a.causal || rpos <= mypos);          // key = row, query = this lane
            const bool ok_b = pair && (!a.causal || mypos <= rpos);          // query = row, key = this lane
            const float s2v = sacc[r] * c2;
            const float pa = ok_a ? __builtin_amdgcn_exp2f(s2v - my_lse2) : 0.f;
            const float pb = ok_b ? __builtin_amdgcn_exp2f(s2v - l4[e] * 1.4426950408889634f) : 0.f;
            dsa[r] = pa * (pt[r] - my_D);
            dsb[r] = pb * (pn[r] - d4[e]);
            sacc[r] = pb;
        }
    }
    f32x16_t acck[DT], accv[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acck[d][r] = 0.f; accv[d][r] = 0.f; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        u32x4_t wa, wb, wp;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            wa[e] = pack_bf16x2(dsa[8 * s2 + 2 * e], dsa[8 * s2 + 2 * e + 1]);
            wb[e] = pack_bf16x2(dsb[8 * s2 + 2 * e], dsb[8 * s2 + 2 * e + 1]);
            wp[e] = pack_bf16x2(sacc[8 * s2 + 2 * e], sacc[8 * s2 + 2 * e + 1]);
        }
        const bf16x8_t fa = __builtin_bit_cast(bf16x8_t, wa), fb = __builtin_bit_cast(bf16x8_t, wb), fp = __builtin_bit_cast(bf16x8_t, wp);
#pragma unroll
        for (int d = 0; d < DT; ++d) {
            asm volatile("" : "+v"(ulo[s2][d]), "+v"(uhi[s2][d]), "+v"(dlo[s2][d]), "+v"(dhi[s2][d]));
            const bf16x8_t uT = __builtin_shufflevector(ulo[s2][d], uhi[s2][d], 0, 1, 2, 3, 4, 5, 6, 7);
            const bf16x8_t dT = __builtin_shufflevector(dlo[s2][d], dhi[s2][d], 0, 1, 2, 3, 4, 5, 6, 7);
            acck[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(uT, fa, acck[d], 0, 0, 0);       // dQ^T += K^T dS^T
            acck[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(uT, fb, acck[d], 0, 0, 0);       // dK^T += Q^T dS
            accv[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dT, fp, accv[d], 0, 0, 0);       // dV^T += dO^T P
        }
    }
    // du rows go back through the wave's U rows in LDS (all fragment reads are done) and leave as whole 16-B chunks, 8 lanes per row
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float f[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) f[e] = __builtin_fmaf(acck[d][4 * g + e], a.scale, accv[d][4 * g + e]);
            u32x2_t ov;
            ov[0] = pack_bf16x2(f[0], f[1]);
            ov[1] = pack_bf16x2(f[2], f[3]);
            *reinterpret_cast<u32x2_t*>(lds + lr * ROWB + (attn_swz<CPR>(lr, d * 4 + g) << 4) + 8 * h) = ov;
        }
#pragma unroll
    for (int i = 0; i < PCS; ++i) {
        const int idx = i * 64 + lane, row = idx / CPR;
        if (!small_row_valid<TP>(row, seq0, a.nseq, a.Sq)) continue;
        const u32x4_t v = *reinterpret_cast<const u32x4_t*>(lds + idx * 16);
        *reinterpret_cast<u32x4_t*>(a.dq + seq_base(a.qm, (int)(seq0 + row / TP)) + (long long)(row & (TP - 1)) * a.qm.pos_stride + head * DH +
                                    attn_swz<CPR>(row, idx % CPR) * 8) = v;
    }
}

// Backward of the conditioned packed form (attn_small_fwd_kernel<.., CROSS>): dQ per row as above; dK / dV are gradients of the CONDITION rows,
// i.e. sums over every sequence of a clip.  Rounds 1-5 produced them per sequence (two more tensors of the activation's size) and summed them with
// torch (a bf16 -> fp32 copy and a reduction each).  Here a wave owns a run of packed groups of ONE (clip, head): the K / V tiles are loaded once,
// dK^T / dV^T accumulate in registers over the whole run, and leave as fp32 atomics onto the (clip, frame, channel) rows -- one tile per wave.
// Orientation: "a" = lane is the query (S^T = K U^T, dP^T = V dO^T -> dS^T -> dQ^T += K^T dS^T), "b" = lane is the key (S = U K^T, dP = dO V^T -> dS, P ->
// dK^T += U^T dS, dV^T += dO^T P); S is not symmetric any more, so both orientations are multiplied out.
template <int DH, int TP>
__global__ void __launch_bounds__(128) attn_smallx_bwd_kernel(const AttnBwdArgs a, float* __restrict__ dk32, float* __restrict__ dv32, int wpc, int gpw) {
    constexpr int ROWB = DH * 2, CPR = DH / 8, WTILE = 32 * ROWB, KS = DH / 16, DT = DH / 32, PCS = WTILE / 1024, SPW = 32 / TP;
    constexpr int WSTAGE = 4 * WTILE + 256;          // U rows | dO rows | K rows | V rows | lse [32] | D [32]
    __shared__ __attribute__((aligned(1024))) char smem[2 * WSTAGE];
    const int lane = threadIdx.x & 63, h = lane >> 5, lr = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long wid = (long long)blockIdx.x * 2 + wave;
    const int part = (int)(wid % wpc);
    const long long ch = wid / wpc;
    const int head = (int)(ch % a.nhead);
    const int clip = (int)(ch / a.nhead);
    const int nclip = a.nseq / a.km.n_inner, G = a.km.n_inner / SPW;
    if (clip >= nclip) return;
    const int g_lo = part * gpw, g_hi = min(G, g_lo + gpw);
    if (g_lo >= g_hi) return;
    char* lds = smem + wave * WSTAGE;
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page_a);
    const int mypos = lr & (TP - 1);
    // condition rows of the clip: row r <- frame r % TP (the SPW copies are identical)
#pragma unroll
    for (int i = 0; i < PCS; ++i) {
        const int idx = i * 64 + lane, row = idx / CPR, pos = row & (TP - 1);
        const bf16_t* sk = zero;
        const bf16_t* sv = zero;
        if (pos < a.Sk) {
            const long long o = seq_base(a.km, clip * a.km.n_inner) + (long long)pos * a.km.pos_stride + head * DH + attn_swz<CPR>(row, idx % CPR) * 8;
            sk = a.k + o; sv = a.v + o;
        }
        __builtin_amdgcn_global_load_lds(GLB_PTR(sk), LDS_PTR(lds + 2 * WTILE + i * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(GLB_PTR(sv), LDS_PTR(lds + 3 * WTILE + i * 1024), 16, 0, 0);
    }
    const int g16 = lane >> 4, rr = (lane >> 2) & 3, qq = lane & 3;
    const uint32_t lds_off = attn_lds_offset(lds);
    const float c2 = a.scale * 1.4426950408889634f;
    f32x16_t acckk[DT], accv[DT];                    // dK^T, dV^T of the wave's condition rows, summed over the run
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acckk[d][r] = 0.f; accv[d][r] = 0.f; }

    for (int g = g_lo; g < g_hi; ++g) {
        const long long seq0 = (long long)clip * a.km.n_inner + (long long)g * SPW;
#pragma unroll
        for (int i = 0; i < PCS; ++i) {
            const int idx = i * 64 + lane, row = idx / CPR;
            const bf16_t* su = zero;
            const bf16_t* sd = zero;
            if ((row & (TP - 1)) < a.Sq) {
                const int seq = (int)(seq0 + row / TP), pos = row & (TP - 1), c = attn_swz<CPR>(row, idx % CPR) * 8;
                su = a.q + seq_base(a.qm, seq) + (long long)pos * a.qm.pos_stride + head * DH + c;
                sd = a.dO + seq_base(a.om, seq) + (long long)pos * a.om.pos_stride + head * DH + c;
            }
            __builtin_amdgcn_global_load_lds(GLB_PTR(su), LDS_PTR(lds + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(GLB_PTR(sd), LDS_PTR(lds + WTILE + i * 1024), 16, 0, 0);
        }
        const bool my_ok = mypos < a.Sq;
        {   // lanes 0..31: lse of row lr, lanes 32..63: D of row lr
            const float* src = reinterpret_cast<const float*>(zero);
            if (my_ok) {
                const long long tok = (seq_base(a.om, (int)(seq0 + lr / TP)) + (long long)mypos * a.om.pos_stride) / a.C;
                src = (h ? a.D : a.lse) + tok * a.nhead + head;
            }
            __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(lds + 4 * WTILE), 4, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

        bf16x8_t uf[KS], df[KS], kf[KS], vf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int o = lr * ROWB + (attn_swz<CPR>(lr, ks * 2 + h) << 4);
            uf[ks] = *reinterpret_cast<const bf16x8_t*>(lds + o);
            df[ks] = *reinterpret_cast<const bf16x8_t*>(lds + WTILE + o);
            kf[ks] = *reinterpret_cast<const bf16x8_t*>(lds + 2 * WTILE + o);
            vf[ks] = *reinterpret_cast<const bf16x8_t*>(lds + 3 * WTILE + o);
        }
        const float* lse_l = reinterpret_cast<const float*>(lds + 4 * WTILE);
        const float* D_l = lse_l + 32;
        const float my_lse2 = lse_l[lr] * 1.4426950408889634f, my_D = D_l[lr];
        f32x16_t st, pt, sn, pn;
#pragma unroll
        for (int r = 0; r < 16; ++r) { st[r] = 0.f; pt[r] = 0.f; sn[r] = 0.f; pn[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], uf[ks], st, 0, 0, 0);          // S^T[key][query]
            pt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[ks], df[ks], pt, 0, 0, 0);          // dP^T[key][query]
            sn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(uf[ks], kf[ks], sn, 0, 0, 0);          // S[query][key]
            pn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(df[ks], vf[ks], pn, 0, 0, 0);          // dP[query][key]
        }
        f32x16_t dsa, dsb;               // sn is reused for P (lane = key orientation)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const f32x4_t l4 = *reinterpret_cast<const f32x4_t*>(lse_l + 8 * gq + 4 * h);
            const f32x4_t d4 = *reinterpret_cast<const f32x4_t*>(D_l + 8 * gq + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * gq + e, row = 8 * gq + 4 * h + e, rpos = row & (TP - 1);
                const bool pair = my_ok && ((row ^ lr) & ~(TP - 1)) == 0 && rpos < a.Sq;
                const bool ok_a = pair && (!a.causal || rpos <= mypos);          // key = row, query = this lane
                const bool ok_b = pair && (!a.causal || mypos <= rpos);          // query = row, key = this lane
                const float pa = ok_a ? __builtin_amdgcn_exp2f(st[r] * c2 - my_lse2) : 0.f;
                const float pb = ok_b ? __builtin_amdgcn_exp2f(sn[r] * c2 - l4[e] * 1.4426950408889634f) : 0.f;
                dsa[r] = pa * (pt[r] - my_D);
                dsb[r] = pb * (pn[r] - d4[e]);
                sn[r] = pb;
            }
        }
        f32x16_t accq[DT];
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) accq[d][r] = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            u32x4_t wa, wb, wp;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                wa[e] = pack_bf16x2(dsa[8 * s2 + 2 * e], dsa[8 * s2 + 2 * e + 1]);
                wb[e] = pack_bf16x2(dsb[8 * s2 + 2 * e], dsb[8 * s2 + 2 * e + 1]);
                wp[e] = pack_bf16x2(sn[8 * s2 + 2 * e], sn[8 * s2 + 2 * e + 1]);
            }
            const bf16x8_t fa = __builtin_bit_cast(bf16x8_t, wa), fb = __builtin_bit_cast(bf16x8_t, wb), fp = __builtin_bit_cast(bf16x8_t, wp);
            const int r0 = 16 * s2 + 4 * (g16 >> 1) + rr, r1 = r0 + 8;
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                const int col = d * 32 + 16 * (g16 & 1) + 4 * qq;
                const uint32_t o0 = lds_off + (uint32_t)(r0 * ROWB + (attn_swz<CPR>(r0, col >> 3) << 4) + (col & 7) * 2);
                const uint32_t o1 = lds_off + (uint32_t)(r1 * ROWB + (attn_swz<CPR>(r1, col >> 3) << 4) + (col & 7) * 2);
                bf16x4_t ulo = attn_tr16(o0), uhi = attn_tr16(o1);
                bf16x4_t dlo = attn_tr16i<WTILE>(o0), dhi = attn_tr16i<WTILE>(o1);
                bf16x4_t klo = attn_tr16i<2 * WTILE>(o0), khi = attn_tr16i<2 * WTILE>(o1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                asm volatile("" : "+v"(ulo), "+v"(uhi), "+v"(dlo), "+v"(dhi), "+v"(klo), "+v"(khi));
                const bf16x8_t uT = __builtin_shufflevector(ulo, uhi, 0, 1, 2, 3, 4, 5, 6, 7);
                const bf16x8_t dT = __builtin_shufflevector(dlo, dhi, 0, 1, 2, 3, 4, 5, 6, 7);
                const bf16x8_t kT = __builtin_shufflevector(klo, khi, 0, 1, 2, 3, 4, 5, 6, 7);
                accq[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kT, fa, accq[d], 0, 0, 0);          // dQ^T += K^T dS^T
                acckk[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(uT, fb, acckk[d], 0, 0, 0);        // dK^T += U^T dS
                accv[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dT, fp, accv[d], 0, 0, 0);          // dV^T += dO^T P
            }
        }
        // dQ rows go back through the wave's U rows in LDS (all fragment reads of this group are done) and leave as whole 16-B chunks
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                u32x2_t ov;
                ov[0] = pack_bf16x2(accq[d][4 * gq] * a.scale, accq[d][4 * gq + 1] * a.scale);
                ov[1] = pack_bf16x2(accq[d][4 * gq + 2] * a.scale, accq[d][4 * gq + 3] * a.scale);
                *reinterpret_cast<u32x2_t*>(lds + lr * ROWB + (attn_swz<CPR>(lr, d * 4 + gq) << 4) + 8 * h) = ov;
            }
#pragma unroll
        for (int i = 0; i < PCS; ++i) {
            const int idx = i * 64 + lane, row = idx / CPR;
            if ((row & (TP - 1)) >= a.Sq) continue;
            const u32x4_t v = *reinterpret_cast<const u32x4_t*>(lds + idx * 16);
            *reinterpret_cast<u32x4_t*>(a.dq + seq_base(a.qm, (int)(seq0 + row / TP)) + (long long)(row & (TP - 1)) * a.qm.pos_stride + head * DH +
                                        attn_swz<CPR>(row, idx % CPR) * 8) = v;
        }
        // (the next group's DMA overwrites the U / dO rows: the reads above are complete -- lgkmcnt(0) is implied by the data dependence of the stores)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    // flush: lane = condition row lr -> frame lr % TP; the SPW copies of a frame (lanes lr, lr + TP, ...) are summed first; accumulator (d, r) of lane half h is
    // channel 32 d + (r & 3) + 8 (r >> 2) + 4 h
    const bool flush = mypos < a.Sk && (lr / TP) == 0;
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float vk = acckk[d][r] * a.scale, vv = accv[d][r];
#pragma unroll
            for (int o = TP; o < 32; o <<= 1) { vk += __shfl_xor(vk, o, 64); vv += __shfl_xor(vv, o, 64); }
            if (flush) {
                const long long o = ((long long)clip * a.Sk + mypos) * a.Ckv + head * DH + 32 * d + (r & 3) + 8 * (r >> 2) + 4 * h;
                atomicAdd(dk32 + o, vk);
                atomicAdd(dv32 + o, vv);
            }
        }
}

// q, k, v, dO: as forward (dO / out / resid share out_map).  Self-attention (q == k == v): du receives dQ + dK + dV.
// Otherwise dq gets dQ (q-map) and dk / dv (kv-map addressing, caller-provided buffers) get dK / dV.
static int attention_bwd_impl(const void* q, const void* k, const void* v, const void* out, const void* resid, const void* dO,
                              const float* lse, float* D_ws, void* dq, void* dk, void* dv, int nseq, int nhead, int d_head, int Sq,
                              int Sk, const int64_t* q_map, const int64_t* kv_map, const int64_t* out_map, const int64_t* dkv_map,
                              float scale, int causal, int out_channels, int64_t out_tokens, float dropout_p, uint64_t seed, void* stream) {
    GENIE_CHECK_ARG(q && k && v && out && dO && lse && D_ws && dq, "genie_attention_bwd: null pointer");
    GENIE_CHECK_ARG(d_head == 8 || d_head == 16 || d_head == 32 || d_head == 64 || d_head == 128, "genie_attention_bwd: d_head %d not in {8, 16, 32, 64, 128}", d_head);
    const bool self = (q == k && k == v);
    GENIE_CHECK_ARG(self || (dk && dv), "genie_attention_bwd: dk/dv required when k/v differ from q");
    AttnBwdArgs a;
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.dO = (const bf16_t*)dO; a.lse = lse; a.D = D_ws;
    a.lse2 = D_ws + out_tokens * nhead;
    a.negD = D_ws + 2 * out_tokens * nhead;
    a.dq = (bf16_t*)dq; a.dk = self ? (bf16_t*)dq : (bf16_t*)dk; a.dv = (bf16_t*)dv; a.dq_in = (const bf16_t*)dq;
    a.qm = mk_map(q_map); a.km = mk_map(kv_map); a.om = mk_map(out_map); a.dkm = dkv_map ? mk_map(dkv_map) : a.km;
    a.nseq = nseq; a.nhead = nhead; a.Sq = Sq; a.Sk = Sk; a.C = out_channels; a.Ckv = 0; a.scale = scale; a.causal = causal;
    a.kv_same = (k == v) ? 1 : 0; a.fuse_self = self ? 1 : 0; a.xcd_swizzle = 0;
    a.out = (const bf16_t*)out; a.resid = (const bf16_t*)resid;
    hipStream_t s = (hipStream_t)stream;
    GENIE_CHECK_ARG(scale > 0.f, "genie_attention_bwd: scale must be positive (got %g)", (double)scale);
    const bool drop = dropout_p > 0.f;
    if (drop) {
        GENIE_CHECK_ARG(dropout_p < 1.f, "genie_attention_bwd_dropout: dropout_p %g not in [0, 1)", (double)dropout_p);
        GENIE_CHECK_ARG((long long)Sq * Sk < (1ll << 32), "genie_attention_bwd_dropout: Sq * Sk must fit 32 bits");
        const AttnDrop d = attn_drop_of(dropout_p, seed);
        a.drop_thr = d.thr; a.drop_key = d.key; a.drop_scale = d.scale;
    }
    if (d_head < 32) return genie_attn_narrow_bwd(a, d_head, s);                                     // D + dQ, then dK / dV (attention_narrow.hip)
    const unsigned pblocks = (unsigned)((out_tokens + 3) / 4);
    if (d_head == 32) attn_bwd_prep_kernel<32><<<pblocks, 256, 0, s>>>((const bf16_t*)dO, (const bf16_t*)out, (const bf16_t*)resid, D_ws, out_tokens, out_channels, nhead, lse, D_ws + out_tokens * nhead, D_ws + 2 * out_tokens * nhead);
    else if (d_head == 64) attn_bwd_prep_kernel<64><<<pblocks, 256, 0, s>>>((const bf16_t*)dO, (const bf16_t*)out, (const bf16_t*)resid, D_ws, out_tokens, out_channels, nhead, lse, D_ws + out_tokens * nhead, D_ws + 2 * out_tokens * nhead);
    else attn_bwd_prep_kernel<128><<<pblocks, 256, 0, s>>>((const bf16_t*)dO, (const bf16_t*)out, (const bf16_t*)resid, D_ws, out_tokens, out_channels, nhead, lse, D_ws + out_tokens * nhead, D_ws + 2 * out_tokens * nhead);
    GENIE_CHECK_LAUNCH();
    GENIE_CHECK_ARG(scale > 0.f, "genie_attention_bwd: scale must be positive (got %g)", (double)scale);
    if (!drop && self && Sq == Sk && Sq <= 32 && same_map(a.qm, a.km) && small_attn_mode()) {
        const int tp = Sq <= 8 ? 8 : (Sq <= 16 ? 16 : 32);
        const long long waves = ((long long)nseq + 32 / tp - 1) / (32 / tp) * nhead;
        const unsigned blocks = (unsigned)((waves + 1) / 2);
#define GENIE_ATTN_SMALL(DHv)                                                                            \
    do {                                                                                                 \
        if (tp == 8) attn_small_bwd_kernel<DHv, 8><<<blocks, 128, 0, s>>>(a);                            \
        else if (tp == 16) attn_small_bwd_kernel<DHv, 16><<<blocks, 128, 0, s>>>(a);                     \
        else attn_small_bwd_kernel<DHv, 32><<<blocks, 128, 0, s>>>(a);                                   \
    } while (0)
        if (d_head == 32) GENIE_ATTN_SMALL(32); else if (d_head == 64) GENIE_ATTN_SMALL(64); else GENIE_ATTN_SMALL(128);
#undef GENIE_ATTN_SMALL
        GENIE_CHECK_LAUNCH();
        return GENIE_OK;
    }
    int nwq = (Sq + 31) / 32; nwq = nwq >= 3 ? 4 : nwq;
    int nwk = (Sk + 31) / 32; nwk = nwk >= 3 ? 4 : nwk;
    const int qtiles = (Sq + 32 * nwq - 1) / (32 * nwq), ktiles = (Sk + 32 * nwk - 1) / (32 * nwk);
    const int tile = 64 * d_head * 2;
    const int lds_q = 3 * (a.kv_same ? tile : 2 * tile);
    const int lds_k = 3 * (2 * tile + 512);
    dim3 gq((unsigned)(nseq * qtiles), nhead), gk((unsigned)(nseq * ktiles), nhead);
#define GENIE_ATTN_DQ(DHv, NWv)                                                                          \
    do {                                                                                                 \
        auto kq = a.kv_same ? attn_bwd_dq_kernel<DHv, NWv, true> : attn_bwd_dq_kernel<DHv, NWv, false>;  \
        if (lds_q > 65536) GENIE_CHECK_ARG(hipFuncSetAttribute((const void*)kq, hipFuncAttributeMaxDynamicSharedMemorySize, lds_q) == hipSuccess, "hipFuncSetAttribute failed"); \
        kq<<<gq, 64 * NWv, lds_q, s>>>(a);                                                               \
    } while (0)
#define GENIE_ATTN_DKV(DHv, NWv)                                                                         \
    do {                                                                                                 \
        auto kk = attn_bwd_dkv_kernel<DHv, NWv>;                                                         \
        if (lds_k > 65536) GENIE_CHECK_ARG(hipFuncSetAttribute((const void*)kk, hipFuncAttributeMaxDynamicSharedMemorySize, lds_k) == hipSuccess, "hipFuncSetAttribute failed"); \
        kk<<<gk, 64 * NWv, lds_k, s>>>(a);                                                               \
    } while (0)
    const int lean = drop ? 0 : genie_attn_lean_bwd_mask(a, d_head);       // d_head 64: the register-lean kernels (attention_lean.hip) where they apply
#define GENIE_ATTN_BWD_DH(DHv)                                                                           \
    do {                                                                                                 \
        if (lean & 1) { if (int rc = genie_attn_lean_bwd_dq(a, s)) return rc; }                          \
        else if (nwq == 1) GENIE_ATTN_DQ(DHv, 1); else if (nwq == 2) GENIE_ATTN_DQ(DHv, 2); else GENIE_ATTN_DQ(DHv, 4);      \
        if (lean & 2) { if (int rc = genie_attn_lean_bwd_dkv(a, s)) return rc; }                         \
        else if (nwk == 1) GENIE_ATTN_DKV(DHv, 1); else if (nwk == 2) GENIE_ATTN_DKV(DHv, 2); else GENIE_ATTN_DKV(DHv, 4);   \
    } while (0)
    if (d_head == 32) GENIE_ATTN_BWD_DH(32);
    else if (d_head == 64) GENIE_ATTN_BWD_DH(64);
    else GENIE_ATTN_BWD_DH(128);
#undef GENIE_ATTN_BWD_DH
#undef GENIE_ATTN_DKV
#undef GENIE_ATTN_DQ
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}

extern "C" int genie_attention_bwd(const void* q, const void* k, const void* v, const void* out, const void* resid, const void* dO,
                                   const float* lse, float* D_ws, void* dq, void* dk, void* dv, int nseq, int nhead, int d_head, int Sq,
                                   int Sk, const int64_t* q_map, const int64_t* kv_map, const int64_t* out_map, const int64_t* dkv_map,
                                   float scale, int causal, int out_channels, int64_t out_tokens, void* stream) {
    return attention_bwd_impl(q, k, v, out, resid, dO, lse, D_ws, dq, dk, dv, nseq, nhead, d_head, Sq, Sk, q_map, kv_map, out_map, dkv_map, scale, causal,
                              out_channels, out_tokens, 0.f, 0, stream);
}

// Backward of genie_attention_fwd_dropout: the same (dropout_p, seed) reproduce the forward's mask.  `out` is the dropped output the forward
// stored (D = rowsum(dO o out) then carries the mask); dS = P o (M dP / (1 - p) - D), dV = (P o M / (1 - p))^T dO.
extern "C" int genie_attention_bwd_dropout(const void* q, const void* k, const void* v, const void* out, const void* resid, const void* dO,
                                           const float* lse, float* D_ws, void* dq, void* dk, void* dv, int nseq, int nhead, int d_head, int Sq,
                                           int Sk, const int64_t* q_map, const int64_t* kv_map, const int64_t* out_map, const int64_t* dkv_map,
                                           float scale, int causal, int out_channels, int64_t out_tokens, float dropout_p, uint64_t seed, void* stream) {
    GENIE_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, "genie_attention_bwd_dropout: dropout_p %g not in [0, 1)", (double)dropout_p);
    return attention_bwd_impl(q, k, v, out, resid, dO, lse, D_ws, dq, dk, dv, nseq, nhead, d_head, Sq, Sk, q_map, kv_map, out_map, dkv_map, scale, causal,
                              out_channels, out_tokens, dropout_p, seed, stream);
}


// Conditioned short sequences (see attn_smallx_bwd_kernel): q-map sequences of Sq = Sk <= 32 positions against K / V rows shared by the n_inner sequences of a
// clip (kv_map inner stride 0).  dq: q-map addressing (bf16); dk_f32 / dv_f32: fp32 [nseq / n_inner][Sk][kv_channels], ACCUMULATED (the caller zeroes them).
// Returns GENIE_ERR_ARG when the shape is not this form (the caller then takes genie_attention_bwd).
extern "C" int genie_attention_bwd_cond(const void* q, const void* k, const void* v, const void* out, const void* resid, const void* dO, const float* lse,
                                        float* D_ws, void* dq, float* dk_f32, float* dv_f32, int nseq, int nhead, int d_head, int S, const int64_t* q_map,
                                        const int64_t* kv_map, const int64_t* out_map, float scale, int causal, int out_channels, int kv_channels,
                                        int64_t out_tokens, void* stream) {
    GENIE_CHECK_ARG(q && k && v && out && dO && lse && D_ws && dq && dk_f32 && dv_f32 && q_map && kv_map && out_map, "genie_attention_bwd_cond: null pointer");
    GENIE_CHECK_ARG(scale > 0.f && nseq >= 1 && nhead >= 1 && S >= 1, "genie_attention_bwd_cond: bad scale / empty problem");
    AttnBwdArgs a;
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.dO = (const bf16_t*)dO; a.lse = lse; a.D = D_ws;
    a.lse2 = D_ws + out_tokens * nhead; a.negD = D_ws + 2 * out_tokens * nhead;
    a.dq = (bf16_t*)dq; a.dk = nullptr; a.dv = nullptr; a.dq_in = nullptr;
    a.qm = mk_map(q_map); a.km = mk_map(kv_map); a.om = mk_map(out_map); a.dkm = a.km;
    a.nseq = nseq; a.nhead = nhead; a.Sq = S; a.Sk = S; a.C = out_channels; a.Ckv = kv_channels; a.scale = scale; a.causal = causal;
    a.kv_same = (k == v) ? 1 : 0; a.fuse_self = 0; a.xcd_swizzle = 0; a.out = (const bf16_t*)out; a.resid = (const bf16_t*)resid;
    GENIE_CHECK_ARG(small_cond_ok(a, q, k, v, d_head), "genie_attention_bwd_cond: not a conditioned short-sequence problem (S = %d <= 32, d_head %d in {32, 64}, kv map inner "
                    "stride 0, n_inner %d a multiple of the packed group)", S, d_head, a.km.n_inner);
    GENIE_CHECK_ARG(kv_channels >= nhead * d_head && out_channels >= nhead * d_head, "genie_attention_bwd_cond: channels");
    hipStream_t s = (hipStream_t)stream;
    const unsigned pblocks = (unsigned)((out_tokens + 3) / 4);
    if (d_head == 32) attn_bwd_prep_kernel<32><<<pblocks, 256, 0, s>>>((const bf16_t*)dO, (const bf16_t*)out, (const bf16_t*)resid, D_ws, out_tokens, out_channels, nhead, lse, D_ws + out_tokens * nhead, D_ws + 2 * out_tokens * nhead);
    else attn_bwd_prep_kernel<64><<<pblocks, 256, 0, s>>>((const bf16_t*)dO, (const bf16_t*)out, (const bf16_t*)resid, D_ws, out_tokens, out_channels, nhead, lse, D_ws + out_tokens * nhead, D_ws + 2 * out_tokens * nhead);
    GENIE_CHECK_LAUNCH();
    const int tp = S <= 8 ? 8 : (S <= 16 ? 16 : 32);
    const int nclip = nseq / a.km.n_inner, G = a.km.n_inner / (32 / tp);
    // waves per (clip, head): about 8192 waves in all, at least 8 packed groups per wave (each wave ends with one 32 x d_head tile of atomics per gradient)
    long long wpc = 8192 / ((long long)nclip * nhead);
    if (wpc < 1) wpc = 1;
    if (wpc > (G + 7) / 8) wpc = (G + 7) / 8;
    const int gpw = (int)((G + wpc - 1) / wpc);
    wpc = (G + gpw - 1) / gpw;
    const long long waves = (long long)nclip * nhead * wpc;
    const unsigned blocks = (unsigned)((waves + 1) / 2);
#define GENIE_ATTN_SMALLX(DHv)                                                                                   \
    do {                                                                                                         \
        if (tp == 8) attn_smallx_bwd_kernel<DHv, 8><<<blocks, 128, 0, s>>>(a, dk_f32, dv_f32, (int)wpc, gpw);     \
        else if (tp == 16) attn_smallx_bwd_kernel<DHv, 16><<<blocks, 128, 0, s>>>(a, dk_f32, dv_f32, (int)wpc, gpw); \
        else attn_smallx_bwd_kernel<DHv, 32><<<blocks, 128, 0, s>>>(a, dk_f32, dv_f32, (int)wpc, gpw);            \
    } while (0)
    if (d_head == 32) GENIE_ATTN_SMALLX(32); else GENIE_ATTN_SMALLX(64);
#undef GENIE_ATTN_SMALLX
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}
