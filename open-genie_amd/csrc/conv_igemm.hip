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
#include "common.h"
#include "genie_hip.h"

static __device__ __attribute__((aligned(256))) uint32_t g_zero_page[64];

struct IgemmArgs {
    const bf16_t* src;
    const bf16_t* wgt;
    bf16_t* dst;
    const bf16_t* resid;
    const float* bias;
    const GenieTap* taps;
    int ntaps;
    int N, Ts, Hs, Ws, Cs;
    int To, Ho, Wo;
    int st, sh, sw;
    int M, Ncols, Nstore;
    int w_row_stride;
    int perm_c, perm_f;
    int Td, Hd, Wd, Cd;
    int dmt, dmh, dmw, dot, doh, dow;
    int shuf_c, shuf_q, shuf_r;
    int tiles_m, tiles_n;
    int nk;           // total K chunks
    int act;          // 0 none, 1 silu (epilogue)
};

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

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

    // SMALLC: the tap table lives in LDS behind the two stages
    GenieTap* lds_taps = reinterpret_cast<GenieTap*>(smem + 2 * STAGE);
    if (SMALLC) {
        for (int i = tid; i < a.ntaps * (int)(sizeof(GenieTap) / 4); i += 256)
            reinterpret_cast<int*>(lds_taps)[i] = reinterpret_cast<const int*>(a.taps)[i];
        __syncthreads();
    }

    // ---- per-thread gather state for the A rows this lane stages ----
    int a_n[A_LOADS], a_t[A_LOADS], a_h[A_LOADS], a_w[A_LOADS], a_lc[A_LOADS];
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
        } else {
            a_n[i] = -1; a_t[i] = a_h[i] = a_w[i] = 0;
        }
    }
    int b_row[B_LOADS], b_lc[B_LOADS];
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) {
        const int row = (i * 4 + wave) * 8 + (lane >> 3);
        b_lc[i] = (lane & 7) ^ ((row >> 1) & 7);
        const int n = n0 + row;
        b_row[i] = n < a.Ncols ? (a.perm_f > 1 ? (n % a.perm_c) * a.perm_f + n / a.perm_c : n) : -1;
    }

    const int cpt = a.Cs >> 3;   // SMALLC: 16-B sub-chunks per tap
    int s_tap = 0, s_cb = 0;     // (tap, channel block) of the NEXT chunk to stage (uniform)

    auto stage = [&](int kc, int buf) {
        char* abase = smem + buf * STAGE;
        char* bbase = abase + A_BYTES;
        if (!SMALLC) {
            const GenieTap tp = a.taps[s_tap];
            const int cbase = s_cb * 64;
#pragma unroll
            for (int i = 0; i < A_LOADS; ++i) {
                const int t = a_t[i] + tp.dt, h = a_h[i] + tp.dh, w = a_w[i] + tp.dw;
                const int c = cbase + a_lc[i] * 8;
                const bool ok = a_n[i] >= 0 && (unsigned)t < (unsigned)a.Ts && (unsigned)h < (unsigned)a.Hs &&
                                (unsigned)w < (unsigned)a.Ws && c < tp.nch;
                const unsigned off = (((unsigned)(a_n[i] * a.Ts + t) * a.Hs + h) * a.Ws + w) * a.Cs + tp.c0 + c;
                const bf16_t* p = ok ? a.src + off : zero;
                __builtin_amdgcn_global_load_lds(GLB_PTR(p), LDS_PTR(abase + (i * 4 + wave) * 1024), 16, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < B_LOADS; ++i) {
                const int c = cbase + b_lc[i] * 8;
                const bool ok = b_row[i] >= 0 && c < tp.nch;
                const bf16_t* p = ok ? a.wgt + (size_t)b_row[i] * a.w_row_stride + tp.wofs + c : zero;
                __builtin_amdgcn_global_load_lds(GLB_PTR(p), LDS_PTR(bbase + (i * 4 + wave) * 1024), 16, 0, 0);
            }
            if (++s_cb * 64 >= tp.nch) { s_cb = 0; ++s_tap; }
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

    stage(0, 0);
    __syncthreads();
    for (int kc = 0; kc < a.nk; ++kc) {
        const int cur = kc & 1;
        if (kc + 1 < a.nk) stage(kc + 1, cur ^ 1);
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

    // ---- epilogue: dest offset = rowoff[m] + coloff[n]; rowoff staged through LDS ----
    int* rowoff = reinterpret_cast<int*>(smem);
    if (tid < BM) {
        int m = m0 + tid;
        int off = -1;
        if (m < a.M) {
            const int wo = m % a.Wo; m /= a.Wo;
            const int ho = m % a.Ho; m /= a.Ho;
            const int to = m % a.To; m /= a.To;
            off = (int)((((unsigned)(m * a.Td + to * a.dmt + a.dot) * a.Hd + ho * a.dmh + a.doh) * a.Wd + wo * a.dmw + a.dow) * a.Cd);
        }
        rowoff[tid] = off;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (TN * 32) + j * 32 + (lane & 31);
        if (n >= a.Nstore) continue;
        const int sub = n / a.shuf_c, ch = n - sub * a.shuf_c;
        const int r = sub % a.shuf_r, q = (sub / a.shuf_r) % a.shuf_q, p = sub / (a.shuf_r * a.shuf_q);
        const int coloff = ((p * a.Hd + q) * a.Wd + r) * a.Cd + ch;
        float bias = 0.f;
        if (a.bias && n < a.Ncols) bias = a.bias[a.perm_f > 1 ? (n % a.perm_c) * a.perm_f + n / a.perm_c : n];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r16 = 0; r16 < 16; ++r16) {
                const int row = wm * (TM * 32) + i * 32 + (r16 & 3) + 8 * (r16 >> 2) + 4 * khalf;
                const int ro = rowoff[row];
                if (ro < 0) continue;
                float v = acc[i][j][r16] + bias;
                if (a.resid) v += bf16_to_f32(a.resid[(unsigned)ro + coloff]);
                if (a.act == 1) v = silu_f(v);
                a.dst[(unsigned)ro + coloff] = f32_to_bf16(v);
            }
        }
    }
}

template <int BN, int WM, int WN, bool SMALLC>
static int launch_igemm(const IgemmArgs& a, hipStream_t s) {
    constexpr int STAGE = 128 * 128 + BN * 128;
    const int lds = 2 * STAGE + (SMALLC ? 32 * (int)sizeof(GenieTap) : 0);
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
    hipLaunchKernelGGL(k, dim3(a.tiles_m * a.tiles_n), dim3(256), lds, s, a);
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}

extern "C" int genie_conv_igemm(const GenieConvDesc* d, void* stream) {
    GENIE_CHECK_ARG(d, "genie_conv_igemm: null descriptor");
    GENIE_CHECK_ARG(d->src && d->wgt && d->dst && d->taps, "genie_conv_igemm: null tensor pointer");
    GENIE_CHECK_ARG(d->ntaps >= 1, "genie_conv_igemm: ntaps %d", d->ntaps);
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
    if (a.Nstore <= 32) {
        a.tiles_n = cdiv(a.Nstore, 32);
        return smallc ? launch_igemm<32, 4, 1, true>(a, s) : launch_igemm<32, 4, 1, false>(a, s);
    }
    a.tiles_n = cdiv(a.Nstore, 128);
    return smallc ? launch_igemm<128, 2, 2, true>(a, s) : launch_igemm<128, 2, 2, false>(a, s);
}
