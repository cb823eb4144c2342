// Gather-GEMM on gfx950 MFMA: the one kernel behind Conv3d forward, dgrad (stride 1, strided by
// parity class, and through the depth-to-space-time shuffle) and the 1x1x1 convolutions.
//
//   D[m][n] = sum_{tap j} sum_{c < nch_j}  SRC[pix(m) * step + off_j][c0_j + c] * WGT[row(n)][wofs_j + c]
//
//   m  -> (n, to, ho, wo) on the row grid (N, To, Ho, Wo);  SRC is CL bf16 (N, Ts, Hs, Ws, Cs)
//   zero padding (causal front padding in time, symmetric in space) = predicated gather: lanes whose
//   source coordinate falls outside load from a zero page -- F.pad is never materialised
//   (reference: genie/module/video.py:160,185).
//
// Tile: BM=128 rows x BN cols x BK=64, 4 waves, v_mfma_f32_32x32x16_bf16, fp32 accumulate.
// Staging: global_load_lds_dwordx4 (16 B/lane, no VGPR round trip), double-buffered LDS.  LDS rows
// are 128 B (64 bf16); 16-B sub-chunk c of row r is stored at physical slot c ^ ((r >> 1) & 7) so
// that the ds_read_b128 fragment reads (32 rows x one k-slot) are bank-conflict free; because the
// DMA writes lane-linear, the swizzle is applied to the per-lane *source* address.
#include <stdlib.h>
#include "igemm_common.h"

static __device__ __attribute__((aligned(256))) uint32_t g_zero_page[64];


template <int BN, int WM, int WN, bool SMALLC>
__global__ void __launch_bounds__(256) igemm_kernel(const IgemmArgs a) {
    constexpr int BM = 128;
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    constexpr int A_LOADS = BM * 8 / 256, B_LOADS = BN * 8 / 256;
    static_assert(WM * WN == 4, "4 waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // XCD-aware, bijective block -> tile map: consecutive ids (same XCD) share the A row tile
    int tile_m, tile_n;
    {
        const int nb = a.tiles_m * a.tiles_n, b = blockIdx.x;
        const int q = nb >> 3, r = nb & 7, xcd = b & 7;
        const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
        tile_n = id % a.tiles_n;
        tile_m = id / a.tiles_n;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page);

    // the tap table lives in LDS behind the two stages; pad0 <- linear element offset of the tap inside the source
    GenieTap* lds_taps = reinterpret_cast<GenieTap*>(smem + 2 * STAGE);
    for (int i = tid; i < a.ntaps; i += 256) {
        GenieTap t = a.taps[i];
        t.pad0 = ((t.dt * a.Hs + t.dh) * a.Ws + t.dw) * a.Cs + t.c0;
        lds_taps[i] = t;
    }
    __syncthreads();

    // ---- per-thread gather state for the A rows this lane stages ----
    // a_t/a_h/a_w: source coordinate of tap (0,0,0); a_base: its linear element offset (+ this lane's 16-B sub-chunk)
    int a_n[A_LOADS], a_t[A_LOADS], a_h[A_LOADS], a_w[A_LOADS], a_lc[A_LOADS];
    unsigned a_base[A_LOADS];
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
        const int row = (i * 4 + wave) * 8 + (lane >> 3);
        a_lc[i] = (lane & 7) ^ ((row >> 1) & 7);
        int m = m0 + row;
        if (m < a.M) {
            const int wo = m % a.Wo; m /= a.Wo;
            const int ho = m % a.Ho; m /= a.Ho;
            const int to = m % a.To; m /= a.To;
            a_n[i] = m; a_t[i] = to * a.st; a_h[i] = ho * a.sh; a_w[i] = wo * a.sw;
            a_base[i] = (((unsigned)(m * a.Ts + a_t[i]) * a.Hs + a_h[i]) * a.Ws + a_w[i]) * a.Cs + a_lc[i] * 8;
        } else {
            a_n[i] = -1; a_t[i] = a_h[i] = a_w[i] = 0; a_base[i] = 0;
            a_t[i] = -(1 << 20);      // fails every range check below
        }
    }
    int b_row[B_LOADS], b_lc[B_LOADS];
    const bf16_t* b_ptr[B_LOADS];   // weight row start + this lane's 16-B sub-chunk
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) {
        const int row = (i * 4 + wave) * 8 + (lane >> 3);
        b_lc[i] = (lane & 7) ^ ((row >> 1) & 7);
        const int n = n0 + row;
        b_row[i] = n < a.Ncols ? (a.perm_f > 1 ? (n % a.perm_c) * a.perm_f + n / a.perm_c : n) : -1;
        b_ptr[i] = a.wgt + (size_t)(b_row[i] < 0 ? 0 : b_row[i]) * a.w_row_stride + b_lc[i] * 8;
    }

    const int cpt = a.Cs >> 3;   // SMALLC: 16-B sub-chunks per tap
    int s_tap = 0, s_cb = 0;     // (tap, channel block) of the NEXT chunk to stage (uniform)
    int kc_begin = 0, kc_end = a.nk;
    if (a.split_k > 1) {
        kc_begin = blockIdx.y * a.chunks_per_split;
        kc_end = kc_begin + a.chunks_per_split;
        if (kc_end > a.nk) kc_end = a.nk;
        if (!SMALLC) {               // advance the (tap, channel block) cursor to chunk kc_begin
            int left = kc_begin;
            while (s_tap < a.ntaps) {
                const int n = (a.taps[s_tap].nch + 63) >> 6;
                if (left < n) break;
                left -= n;
                ++s_tap;
            }
            s_cb = left;
        }
    }
    GenieTap cur_tap = lds_taps[s_tap < a.ntaps ? s_tap : 0];

    auto stage = [&](int kc, int buf) {
        char* abase = smem + buf * STAGE;
        char* bbase = abase + A_BYTES;
        if (!SMALLC) {
            // current tap (wave-uniform, kept in SGPRs); branch-free validity: out-of-range lanes read the zero page
            const int t_dt = __builtin_amdgcn_readfirstlane(cur_tap.dt), t_dh = __builtin_amdgcn_readfirstlane(cur_tap.dh);
            const int t_dw = __builtin_amdgcn_readfirstlane(cur_tap.dw), t_wofs = __builtin_amdgcn_readfirstlane(cur_tap.wofs);
            const int t_delta = __builtin_amdgcn_readfirstlane(cur_tap.pad0), t_nch = __builtin_amdgcn_readfirstlane(cur_tap.nch);
            const int cbase = s_cb * 64;
#pragma unroll
            for (int i = 0; i < A_LOADS; ++i) {
                const int t = a_t[i] + t_dt, h = a_h[i] + t_dh, w = a_w[i] + t_dw;
                const bool ok = ((unsigned)t < (unsigned)a.Ts) & ((unsigned)h < (unsigned)a.Hs) & ((unsigned)w < (unsigned)a.Ws) &
                                (cbase + a_lc[i] * 8 < t_nch);
                const bf16_t* p = a.src + (a_base[i] + (unsigned)(t_delta + cbase));
                p = ok ? p : zero;
                __builtin_amdgcn_global_load_lds(GLB_PTR(p), LDS_PTR(abase + (i * 4 + wave) * 1024), 16, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < B_LOADS; ++i) {
                const bool ok = (b_row[i] >= 0) & (cbase + b_lc[i] * 8 < t_nch);
                const bf16_t* p = b_ptr[i] + (t_wofs + cbase);
                p = ok ? p : zero;
                __builtin_amdgcn_global_load_lds(GLB_PTR(p), LDS_PTR(bbase + (i * 4 + wave) * 1024), 16, 0, 0);
            }
            if (++s_cb * 64 >= t_nch) {
                s_cb = 0;
                ++s_tap;
                cur_tap = lds_taps[s_tap < a.ntaps ? s_tap : 0];      // LDS read overlaps the MFMA phase that follows
            }
        } else {
            // several taps per 64-wide K chunk: each 16-B sub-chunk g = kc*8 + lc belongs to tap g / cpt
#pragma unroll
            for (int i = 0; i < A_LOADS; ++i) {
                const int g = kc * 8 + a_lc[i];
                const int tap = g / cpt;
                bool ok = a_n[i] >= 0 && tap < a.ntaps;
                unsigned off = 0;
                if (ok) {
                    const GenieTap tp = lds_taps[tap];
                    const int t = a_t[i] + tp.dt, h = a_h[i] + tp.dh, w = a_w[i] + tp.dw;
                    ok = (unsigned)t < (unsigned)a.Ts && (unsigned)h < (unsigned)a.Hs && (unsigned)w < (unsigned)a.Ws;
                    off = (((unsigned)(a_n[i] * a.Ts + t) * a.Hs + h) * a.Ws + w) * a.Cs + tp.c0 + (g - tap * cpt) * 8;
                }
                const bf16_t* p = ok ? a.src + off : zero;
                __builtin_amdgcn_global_load_lds(GLB_PTR(p), LDS_PTR(abase + (i * 4 + wave) * 1024), 16, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < B_LOADS; ++i) {
                const int g = kc * 8 + b_lc[i];
                const bool ok = b_row[i] >= 0 && g < a.ntaps * cpt;   // packed weights: K contiguous [tap][Cs]
                const bf16_t* p = ok ? a.wgt + (size_t)b_row[i] * a.w_row_stride + g * 8 : zero;
                __builtin_amdgcn_global_load_lds(GLB_PTR(p), LDS_PTR(bbase + (i * 4 + wave) * 1024), 16, 0, 0);
            }
        }
    };

    f32x16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment read addresses (bytes within a stage), per k-step XOR applied below
    int a_rd[TM], a_sw[TM], b_rd[TN], b_sw[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = wm * (TM * 32) + i * 32 + (lane & 31);
        a_rd[i] = row * 128;
        a_sw[i] = (row >> 1) & 7;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int row = wn * (TN * 32) + j * 32 + (lane & 31);
        b_rd[j] = A_BYTES + row * 128;
        b_sw[j] = (row >> 1) & 7;
    }
    const int khalf = lane >> 5;

    if (kc_begin < kc_end) stage(kc_begin, 0);
    __syncthreads();
    for (int kc = kc_begin; kc < kc_end; ++kc) {
        const int cur = (kc - kc_begin) & 1;
        if (kc + 1 < kc_end) stage(kc + 1, cur ^ 1);
        const char* base = smem + cur * STAGE;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8_t af[TM], bfr[TN];
            const int lc = ks * 2 + khalf;
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[i] = *reinterpret_cast<const bf16x8_t*>(base + a_rd[i] + ((lc ^ a_sw[i]) << 4));
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bfr[j] = *reinterpret_cast<const bf16x8_t*>(base + b_rd[j] + ((lc ^ b_sw[j]) << 4));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- split-K: dump the fp32 partial tile, the finish kernel does bias / resid / act / layout ----
    if (a.split_k > 1) {
        igemm_store_partials<TM, TN>(a, acc, blockIdx.y, m0, n0, wm, wn, lane);
        return;
    }

    igemm_epilogue<BM, TM, TN>(a, acc, smem, m0, n0, wm, wn, tid, lane);
}

// split-K finish: sum the partial tiles, then the same epilogue (bias, resid, act, destination mapping)
__global__ void __launch_bounds__(256) igemm_splitk_finish_kernel(const IgemmArgs a) {
    const long long total = (long long)a.M * a.Nstore;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        int m = (int)(i / a.Nstore);
        const int n = (int)(i - (long long)m * a.Nstore);
        float v = 0.f;
        for (int s = 0; s < a.split_k; ++s) v += a.ws[((size_t)s * a.M + m) * a.ws_ld + n];
        if (a.bias && n < a.Ncols) v += a.bias[a.perm_f > 1 ? (n % a.perm_c) * a.perm_f + n / a.perm_c : n];
        const int wo = m % a.Wo; m /= a.Wo;
        const int ho = m % a.Ho; m /= a.Ho;
        const int to = m % a.To; m /= a.To;
        const unsigned ro = (((unsigned)(m * a.Td + to * a.dmt + a.dot) * a.Hd + ho * a.dmh + a.doh) * a.Wd + wo * a.dmw + a.dow) * a.Cd;
        const int sub = n / a.shuf_c, ch = n - sub * a.shuf_c;
        const int r = sub % a.shuf_r, q = (sub / a.shuf_r) % a.shuf_q, p = sub / (a.shuf_r * a.shuf_q);
        const unsigned off = ro + ((p * a.Hd + q) * a.Wd + r) * a.Cd + ch;
        if (a.resid) v += bf16_to_f32(a.resid[off]);
        if (a.act == 1) v = silu_f(v);
        a.dst[off] = f32_to_bf16(v);
    }
}

// Same, four consecutive columns per thread (16-B partial reads, 8-B residual reads / stores, one row decode per four outputs).
// Needs Nstore, shuf_c and Cd to be multiples of 4 -- the epilogue's vec4 condition.  The finish pass runs after every split-K
// launch (136 per tokenizer step at B = 8); the scalar form above moved 2.2 TB/s.
__global__ void __launch_bounds__(256) igemm_splitk_finish4_kernel(const IgemmArgs a) {
    const int n4 = a.Nstore >> 2;
    const long long total = (long long)a.M * n4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        int m = (int)(i / n4);
        const int n = (int)(i - (long long)m * n4) * 4;
        f32x4_t v = {0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < a.split_k; ++s) {
            const f32x4_t w = *reinterpret_cast<const f32x4_t*>(a.ws + ((size_t)s * a.M + m) * a.ws_ld + n);
            v[0] += w[0]; v[1] += w[1]; v[2] += w[2]; v[3] += w[3];
        }
        if (a.bias) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ne = n + e;
                if (ne < a.Ncols) v[e] += a.bias[a.perm_f > 1 ? (ne % a.perm_c) * a.perm_f + ne / a.perm_c : ne];
            }
        }
        const int wo = m % a.Wo; m /= a.Wo;
        const int ho = m % a.Ho; m /= a.Ho;
        const int to = m % a.To; m /= a.To;
        const unsigned ro = (((unsigned)(m * a.Td + to * a.dmt + a.dot) * a.Hd + ho * a.dmh + a.doh) * a.Wd + wo * a.dmw + a.dow) * a.Cd;
        const int sub = n / a.shuf_c, ch = n - sub * a.shuf_c;
        const int r = sub % a.shuf_r, q = (sub / a.shuf_r) % a.shuf_q, p = sub / (a.shuf_r * a.shuf_q);
        const unsigned off = ro + ((p * a.Hd + q) * a.Wd + r) * a.Cd + ch;
        if (a.resid) {
            const u32x2_t rv = *reinterpret_cast<const u32x2_t*>(a.resid + off);
            v[0] += __uint_as_float(rv[0] << 16); v[1] += __uint_as_float(rv[0] & 0xffff0000u);
            v[2] += __uint_as_float(rv[1] << 16); v[3] += __uint_as_float(rv[1] & 0xffff0000u);
        }
        if (a.act == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
        }
        u32x2_t ov;
        ov[0] = pack_bf16x2(v[0], v[1]);
        ov[1] = pack_bf16x2(v[2], v[3]);
        *reinterpret_cast<u32x2_t*>(a.dst + off) = ov;
    }
}

int genie_igemm_splitk_finish(const IgemmArgs& a, hipStream_t s) {      // also used by the kw-triple kernels (conv_igemm3.hip)
    const bool vec4 = (a.shuf_c & 3) == 0 && (a.Nstore & 3) == 0 && (a.Cd & 3) == 0 && (a.ws_ld & 3) == 0;
    long long total = (long long)a.M * (vec4 ? a.Nstore >> 2 : a.Nstore);
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    if (vec4) hipLaunchKernelGGL(igemm_splitk_finish4_kernel, dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(igemm_splitk_finish_kernel, dim3(grid), dim3(256), 0, s, a);
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}

template <int BN, int WM, int WN, bool SMALLC>
static int launch_igemm(const IgemmArgs& a, hipStream_t s) {
    constexpr int STAGE = 128 * 128 + BN * 128;
    const int lds = 2 * STAGE + 256 * (int)sizeof(GenieTap);
    auto k = igemm_kernel<BN, WM, WN, SMALLC>;
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            genie_set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return GENIE_ERR_HIP;
        }
        configured = true;
    }
    hipLaunchKernelGGL(k, dim3(a.tiles_m * a.tiles_n, a.split_k > 1 ? a.split_k : 1), dim3(256), lds, s, a);
    GENIE_CHECK_LAUNCH();
    if (a.split_k > 1) return genie_igemm_splitk_finish(a, s);
    return GENIE_OK;
}

int genie_conv_igemm3_try(const GenieConvDesc* d, IgemmArgs a, hipStream_t s);   // conv_igemm3.hip
int genie_conv_gemm_try(const GenieConvDesc* d, IgemmArgs a, hipStream_t s);     // conv_gemm.hip

extern "C" int genie_conv_igemm(const GenieConvDesc* d, void* stream) {
    GENIE_CHECK_ARG(d, "genie_conv_igemm: null descriptor");
    GENIE_CHECK_ARG(d->src && d->wgt && d->dst && d->taps, "genie_conv_igemm: null tensor pointer");
    GENIE_CHECK_ARG(d->ntaps >= 1 && d->ntaps <= 256, "genie_conv_igemm: ntaps %d out of range [1, 256]", d->ntaps);
    GENIE_CHECK_ARG(d->Cs % 8 == 0 && d->w_row_stride % 8 == 0, "genie_conv_igemm: Cs=%d and w_row_stride=%d must be multiples of 8", d->Cs, d->w_row_stride);
    GENIE_CHECK_ARG(d->N > 0 && d->To > 0 && d->Ho > 0 && d->Wo > 0 && d->Ncols > 0, "genie_conv_igemm: empty problem");
    GENIE_CHECK_ARG((long long)d->N * d->Ts * d->Hs * d->Ws * d->Cs < (1ll << 31), "genie_conv_igemm: source tensor exceeds 2^31 elements");
    GENIE_CHECK_ARG((long long)d->N * d->Td * d->Hd * d->Wd * d->Cd < (1ll << 31), "genie_conv_igemm: destination tensor exceeds 2^31 elements");
    GENIE_CHECK_ARG(d->shuf_c >= 1 && d->shuf_q >= 1 && d->shuf_r >= 1, "genie_conv_igemm: bad shuffle spec");
    IgemmArgs a;
    a.src = (const bf16_t*)d->src; a.wgt = (const bf16_t*)d->wgt; a.dst = (bf16_t*)d->dst;
    a.resid = (const bf16_t*)d->resid; a.bias = d->bias; a.taps = d->taps; a.ntaps = d->ntaps;
    a.N = d->N; a.Ts = d->Ts; a.Hs = d->Hs; a.Ws = d->Ws; a.Cs = d->Cs;
    a.To = d->To; a.Ho = d->Ho; a.Wo = d->Wo; a.st = d->st; a.sh = d->sh; a.sw = d->sw;
    const long long M = (long long)d->N * d->To * d->Ho * d->Wo;
    GENIE_CHECK_ARG(M < (1ll << 31), "genie_conv_igemm: too many rows");
    a.M = (int)M; a.Ncols = d->Ncols;
    a.w_row_stride = d->w_row_stride; a.perm_c = d->perm_c; a.perm_f = d->perm_f < 1 ? 1 : d->perm_f;
    a.Td = d->Td; a.Hd = d->Hd; a.Wd = d->Wd; a.Cd = d->Cd;
    a.dmt = d->dmt; a.dmh = d->dmh; a.dmw = d->dmw; a.dot = d->dot; a.doh = d->doh; a.dow = d->dow;
    a.shuf_c = d->shuf_c; a.shuf_q = d->shuf_q; a.shuf_r = d->shuf_r;
    // without a shuffle the pad channels [Ncols, Cd) of the destination are written as zeros
    a.Nstore = (d->shuf_c >= d->Ncols) ? (d->Cd > d->Ncols ? d->Cd : d->Ncols) : d->Ncols;
    if (d->shuf_c >= d->Ncols) a.shuf_c = a.Nstore;
    a.act = d->act;
    const bool smallc = d->small_c != 0;
    if (smallc) {
        GENIE_CHECK_ARG(d->Cs == 8 || d->Cs == 16 || d->Cs == 32, "genie_conv_igemm: small_c needs Cs in {8,16,32}, got %d", d->Cs);
        GENIE_CHECK_ARG(d->ntaps <= 32, "genie_conv_igemm: small_c supports at most 32 taps");
        a.nk = cdiv((long long)d->ntaps * (d->Cs / 8), 8);
    } else {
        // chunk count comes from the tap table (host copy supplied by the caller)
        GENIE_CHECK_ARG(d->nk >= 1, "genie_conv_igemm: nk %d", d->nk);
        a.nk = d->nk;
    }
    a.tiles_m = cdiv(M, 128);
    hipStream_t s = (hipStream_t)stream;
    a.split_k = 1; a.chunks_per_split = a.nk; a.ws = nullptr; a.ws_ld = 0;
    a.tiles_n = cdiv(a.Nstore, 128);
    // GroupNorm fusion requests: honoured by the 256-row kw-triple kernels only (conv_igemm3.hip); everything else ignores them
    genie_note_gn_fused(0);
    a.gn_sums = nullptr; a.gnb_x = nullptr; a.gnb_gamma = a.gnb_beta = a.gnb_mean = a.gnb_rstd = nullptr; a.gnb_part = nullptr;
    a.gnb_act = 0; a.gnb_nblk = 0; a.gn_rows = d->To * d->Ho * d->Wo;
    if (d->gn_sums || d->gnb_x) {
        const bool plain = d->shuf_c >= d->Ncols && d->shuf_q == 1 && d->shuf_r == 1 && a.perm_f == 1 && d->dmt == 1 && d->dmh == 1 && d->dmw == 1 &&
                           d->dot == 0 && d->doh == 0 && d->dow == 0 && d->Td == d->To && d->Hd == d->Ho && d->Wd == d->Wo &&
                           (a.Nstore & 3) == 0 && (d->Cd & 3) == 0 && a.Nstore <= d->Cd;
        if (plain && a.gn_rows % 256 == 0) {
            a.gn_sums = (double*)d->gn_sums;
            if (d->gnb_x) {
                GENIE_CHECK_ARG(d->gnb_mean && d->gnb_rstd && d->gnb_part, "genie_conv_igemm: gnb_x needs gnb_mean, gnb_rstd and gnb_part");
                GENIE_CHECK_ARG(d->gnb_nblk == a.gn_rows / 256, "genie_conv_igemm: gnb_nblk %d != rows per sample / 256 = %d", d->gnb_nblk, a.gn_rows / 256);
                a.gnb_x = (const bf16_t*)d->gnb_x; a.gnb_gamma = d->gnb_gamma; a.gnb_beta = d->gnb_beta; a.gnb_mean = d->gnb_mean;
                a.gnb_rstd = d->gnb_rstd; a.gnb_part = d->gnb_part; a.gnb_act = d->gnb_act; a.gnb_nblk = d->gnb_nblk;
            }
        }
    }
    {
        int rc = genie_conv_gemm_try(d, a, s);
        if (rc <= 0) return rc;
        rc = genie_conv_igemm3_try(d, a, s);
        if (rc <= 0) return rc;
    }
    a.gn_sums = nullptr; a.gnb_x = nullptr;              // the generic kernel's tiles are 128 rows: no GroupNorm fusion
    if (a.Nstore <= 32) {
        a.tiles_n = cdiv(a.Nstore, 32);
        genie_note_variant(smallc ? GENIE_VARIANT_IGEMM_32_SMALLC : GENIE_VARIANT_IGEMM_32);
        return smallc ? launch_igemm<32, 4, 1, true>(a, s) : launch_igemm<32, 4, 1, false>(a, s);
    }
    a.tiles_n = cdiv(a.Nstore, 128);
    // split-K: too few output tiles to fill 256 CUs and a long reduction (low-resolution layers, upsample dgrad)
    if (!smallc && d->splitk_ws && a.tiles_m * a.tiles_n < 192 && a.nk >= 8) {
        const int tiles = a.tiles_m * a.tiles_n;
        static const int target = getenv("GENIE_SPLITK_TARGET") ? atoi(getenv("GENIE_SPLITK_TARGET")) : 512;
        int sk = (target + tiles - 1) / tiles;           // ~2 blocks per CU (measured best on the 512-channel 4x8x8 layers: 0.057 ms vs 0.069 at 768)
        if (sk > a.nk / 4) sk = a.nk / 4;                // >= 4 chunks (256 k) per split
        const long long per = (long long)M * a.Nstore * 4;
        if ((long long)sk * per > d->splitk_ws_bytes) sk = (int)(d->splitk_ws_bytes / per);
        if (sk >= 2) {
            a.chunks_per_split = cdiv(a.nk, sk);
            a.split_k = cdiv(a.nk, a.chunks_per_split);
            a.ws = (float*)d->splitk_ws;
            a.ws_ld = a.Nstore;
        }
    }
    genie_note_variant(smallc ? GENIE_VARIANT_IGEMM_128_SMALLC : GENIE_VARIANT_IGEMM_128);
    return smallc ? launch_igemm<128, 2, 2, true>(a, s) : launch_igemm<128, 2, 2, false>(a, s);
}
