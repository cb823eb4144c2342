// Pointwise (1x1x1, stride 1) convolutions and Linear layers as a plain GEMM:  DST[m][n] = sum_k SRC[m][k] * W[n][k]
// (reference: the 1x1 residual projections of module/video.py, the ST-block feed-forwards of module/attention.py and the
// MaskGIT vocabulary head dynamics.py:44 -- Linear(512 -> 2^18) on 4096 tokens is 1.1 TFLOP per direction).
//
// The generic gather-GEMM (conv_igemm.hip) runs these at 465-580 TFLOP/s: a 128 x 128 tile per block, two LDS stages and a
// fresh pipeline fill for every tile, and with K = 512 a tile is only 8 K-steps long.  This kernel is PERSISTENT: one block per
// CU walks its tiles, and the K-tile stream (256 x 64 of SRC + 128 x 64 of W per stage, three stages, LDS-DMA two K-tiles ahead,
// counted vmcnt + one raw barrier per K-tile) runs straight through tile boundaries, so the first K-tiles of the next tile are
// already in flight while the epilogue of the finished one stores.  8 waves, 64 x 64 accumulators per wave (same fragment
// layout and epilogue as the kw-triple kernels).
#include "igemm_common.h"

static __device__ __attribute__((aligned(256))) uint32_t g_zero_page_g[64];

struct GemmArgs {
    IgemmArgs g;
    int nkt;            // K tiles of 64
    int ntiles;
};

__global__ void __launch_bounds__(512) gemm_pw_kernel(const GemmArgs p) {
    constexpr int BM = 256, BN = 128, NWAVE = 8, WN = 2, TM = 2, TN = 2;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES, NSTAGE = 3;
    constexpr int A_LOADS = 4, B_LOADS = 2, NLOAD = A_LOADS + B_LOADS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const IgemmArgs& a = p.g;
    char* const ring = smem;
    char* const epi = smem + NSTAGE * STAGE;          // 1 KiB of row offsets for the epilogue (never touched by the DMA)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page_g);
    const int G = (int)gridDim.x;

    // tile t of this block: rounds of G tiles, inside a round consecutive ids sit on one XCD; tile_n runs fastest, so the tiles
    // an XCD works on at a time share their SRC rows (and, for the wide vocabulary head, all of SRC stays in L2 / MALL)
    auto tile_of = [&](int it, int& m0, int& n0) -> bool {
        const int base = it * G;
        const int left = p.ntiles - base;
        if (left <= 0) return false;
        const int g = left < G ? left : G;                  // size of this round
        if ((int)blockIdx.x >= g) return false;
        const int id = base + xcd_tile_id(g, blockIdx.x);
        m0 = (id / a.tiles_n) * BM;
        n0 = (id % a.tiles_n) * BN;
        return true;
    };

    // ---- loader cursor: (round, K tile) of the NEXT stage to issue ----
    int l_it = 0, l_k = 0;
    bool l_live;
    unsigned a_src[A_LOADS];        // element offset of (row, this lane's 16-B chunk), ~0u = past the last row
    const bf16_t* b_src[B_LOADS];   // nullptr = past the last weight row
    auto loader_setup = [&]() {
        int m0, n0;
        l_live = tile_of(l_it, m0, n0);
        if (!l_live) return;
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            const int row = i * 64 + (tid >> 3);
            const int lc = (tid & 7) ^ ((row >> 1) & 7);
            const int m = m0 + row;
            a_src[i] = m < a.M ? (unsigned)m * (unsigned)a.Cs + lc * 8 : ~0u;
        }
#pragma unroll
        for (int j = 0; j < B_LOADS; ++j) {
            const int row = j * 64 + (tid >> 3);
            const int lc = (tid & 7) ^ ((row >> 1) & 7);
            const int n = n0 + row;
            b_src[j] = n < a.Ncols ? a.wgt + (size_t)n * a.w_row_stride + lc * 8 : nullptr;
        }
    };
    auto issue = [&](int slot) {                    // always NLOAD DMAs per thread (zero page when there is nothing left)
        char* abuf = ring + slot * STAGE;
        char* bbuf = abuf + A_BYTES;
        const int kofs = l_k * 64;
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            const bf16_t* q = (l_live && a_src[i] != ~0u) ? a.src + (a_src[i] + (unsigned)kofs) : zero;
            __builtin_amdgcn_global_load_lds(GLB_PTR(q), LDS_PTR(abuf + (i * NWAVE + wave) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < B_LOADS; ++j) {
            const bf16_t* q = (l_live && b_src[j]) ? b_src[j] + kofs : zero;
            __builtin_amdgcn_global_load_lds(GLB_PTR(q), LDS_PTR(bbuf + (j * NWAVE + wave) * 1024), 16, 0, 0);
        }
        if (l_live && ++l_k == p.nkt) {
            l_k = 0;
            ++l_it;
            loader_setup();
        }
    };

    // fragment read offsets inside a stage (XOR swizzle: not additive in the k-step)
    unsigned a_off[4][TM], b_off[4][TN];
    const int khalf = lane >> 5;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = wm * 64 + i * 32 + (lane & 31);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) a_off[ks][i] = (unsigned)(row * 128 + (((ks * 2 + khalf) ^ ((row >> 1) & 7)) << 4));
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int row = wn * 64 + j * 32 + (lane & 31);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) b_off[ks][j] = (unsigned)(A_BYTES + row * 128 + (((ks * 2 + khalf) ^ ((row >> 1) & 7)) << 4));
    }

    loader_setup();
    issue(0);
    issue(1);
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NLOAD) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    int slot = 0;                                   // stage holding the K tile about to be consumed
    for (int it = 0;; ++it) {
        int m0, n0;
        if (!tile_of(it, m0, n0)) break;
        f32x16_t acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int k = 0; k < p.nkt; ++k) {
            issue(slot == 0 ? 2 : slot - 1);        // the stage consumed one K tile ago (every wave is past that barrier)
            __builtin_amdgcn_sched_barrier(0);
            const char* st = ring + slot * STAGE;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                bf16x8_t af[TM], bfr[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const bf16x8_t*>(st + a_off[ks][i]);
#pragma unroll
                for (int j = 0; j < TN; ++j) bfr[j] = *reinterpret_cast<const bf16x8_t*>(st + b_off[ks][j]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            }
            // everything but the stage just issued has landed -> the next K tile is complete; publish it / fence this one
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NLOAD) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            slot = slot == 2 ? 0 : slot + 1;
        }
        igemm_epilogue<BM, TM, TN>(a, acc, epi, m0, n0, wm, wn, tid, lane);
        __syncthreads();                            // the next tile's epilogue rewrites the row offsets
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// 256 x 256 tile for wide outputs (N > 128): 8 waves as 2 x 4, a wave owns 128 x 64 (TM = 4, TN = 2: 6 fragment reads per
// 8 MFMAs instead of 4 per 4).  K tiles are 32 wide (64-B LDS rows, 32 KiB per stage), FOUR stages, LDS-DMA three K tiles ahead;
// the wait at the end of K tile g leaves only the newest stage in flight, so tiles g + 1 and g + 2 are published by barrier g and
// the first k-step of tile g + 1 can be read BEFORE that barrier (every wave leaves a barrier with MFMA operands in registers).
// Per 1024 MFMA cycles of a SIMD the block moves 32 KiB from L2 to LDS -- the same ratio as the kw-triple conv kernel; the
// 256 x 128 tile above moves 48 KiB.
__global__ void __launch_bounds__(512) gemm_pw256_kernel(const GemmArgs p) {
    constexpr int BM = 256, BN = 256, NWAVE = 8, WN = 4, TM = 4, TN = 2;
    constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64, STAGE = A_BYTES + B_BYTES, NSTAGE = 4;
    constexpr int A_LOADS = 2, B_LOADS = 2, NLOAD = A_LOADS + B_LOADS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const IgemmArgs& a = p.g;
    char* const ring = smem;
    char* const epi = smem + NSTAGE * STAGE;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page_g);
    const int G = (int)gridDim.x;

    auto tile_of = [&](int it, int& m0, int& n0) -> bool {
        const int base = it * G;
        const int left = p.ntiles - base;
        if (left <= 0) return false;
        const int g = left < G ? left : G;
        if ((int)blockIdx.x >= g) return false;
        const int id = base + xcd_tile_id(g, blockIdx.x);
        m0 = (id / a.tiles_n) * BM;
        n0 = (id % a.tiles_n) * BN;
        return true;
    };

    int l_it = 0, l_k = 0;
    bool l_live;
    unsigned a_src[A_LOADS];
    const bf16_t* b_src[B_LOADS];
    auto loader_setup = [&]() {
        int m0, n0;
        l_live = tile_of(l_it, m0, n0);
        if (!l_live) return;
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            const int row = (i * NWAVE + wave) * 16 + (lane >> 2);
            const int lc = (lane & 3) ^ ((row >> 2) & 3);
            const int m = m0 + row;
            a_src[i] = m < a.M ? (unsigned)m * (unsigned)a.Cs + lc * 8 : ~0u;
        }
#pragma unroll
        for (int j = 0; j < B_LOADS; ++j) {
            const int row = (j * NWAVE + wave) * 16 + (lane >> 2);
            const int lc = (lane & 3) ^ ((row >> 2) & 3);
            const int n = n0 + row;
            b_src[j] = n < a.Ncols ? a.wgt + (size_t)n * a.w_row_stride + lc * 8 : nullptr;
        }
    };
    auto issue = [&](int slot) {
        char* abuf = ring + slot * STAGE;
        char* bbuf = abuf + A_BYTES;
        const int kofs = l_k * 32;
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            const bf16_t* q = (l_live && a_src[i] != ~0u) ? a.src + (a_src[i] + (unsigned)kofs) : zero;
            __builtin_amdgcn_global_load_lds(GLB_PTR(q), LDS_PTR(abuf + (i * NWAVE + wave) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < B_LOADS; ++j) {
            const bf16_t* q = (l_live && b_src[j]) ? b_src[j] + kofs : zero;
            __builtin_amdgcn_global_load_lds(GLB_PTR(q), LDS_PTR(bbuf + (j * NWAVE + wave) * 1024), 16, 0, 0);
        }
        if (l_live && ++l_k == p.nkt) {
            l_k = 0;
            ++l_it;
            loader_setup();
        }
    };

    unsigned a_off[2][TM], b_off[2][TN];
    const int khalf = lane >> 5;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = wm * 128 + i * 32 + (lane & 31);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) a_off[ks][i] = (unsigned)(row * 64 + (((ks * 2 + khalf) ^ ((row >> 2) & 3)) << 4));
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int row = wn * 64 + j * 32 + (lane & 31);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) b_off[ks][j] = (unsigned)(A_BYTES + row * 64 + (((ks * 2 + khalf) ^ ((row >> 2) & 3)) << 4));
    }
    auto read_ks = [&](const char* st, int ks, bf16x8_t (&fa)[TM], bf16x8_t (&fb)[TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const bf16x8_t*>(st + a_off[ks][i]);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const bf16x8_t*>(st + b_off[ks][j]);
    };

    loader_setup();
    issue(0);
    issue(1);
    issue(2);
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NLOAD) : "memory");      // K tiles 0 and 1 have landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    bf16x8_t fa0[TM], fb0[TN];
    read_ks(ring, 0, fa0, fb0);
    int slot = 0;
    for (int it = 0;; ++it) {
        int m0, n0;
        if (!tile_of(it, m0, n0)) break;
        f32x16_t acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int k = 0; k < p.nkt; ++k) {
            const char* st = ring + slot * STAGE;
            const char* stn = ring + ((slot + 1) & 3) * STAGE;
            bf16x8_t fa1[TM], fb1[TN];
            read_ks(st, 1, fa1, fb1);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0[i], fb0[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            issue((slot + 3) & 3);                  // the stage consumed one K tile ago (every wave is past that barrier)
            __builtin_amdgcn_sched_barrier(0);
            read_ks(stn, 0, fa0, fb0);              // first k-step of the next K tile: published one barrier ago
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1[i], fb1[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NLOAD) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            slot = (slot + 1) & 3;
        }
        igemm_epilogue<BM, TM, TN>(a, acc, epi, m0, n0, wm, wn, tid, lane);
        __syncthreads();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// Called by genie_conv_igemm (conv_igemm.hip) with the generic arguments filled in.  Returns 1 when the problem is not eligible
// (the caller falls back to the generic kernel), 0 on launch, < 0 on error.
int genie_conv_gemm_try(const GenieConvDesc* d, IgemmArgs a, hipStream_t s) {
    static const int mode = getenv("GENIE_GEMM_PW") ? atoi(getenv("GENIE_GEMM_PW")) : 1;
    if (!mode || !d->pointwise || d->ntaps != 1 || d->small_c) return 1;
    if (d->st != 1 || d->sh != 1 || d->sw != 1 || d->To != d->Ts || d->Ho != d->Hs || d->Wo != d->Ws) return 1;
    if (a.perm_f > 1 || d->nk < 1 || d->Cs < d->nk * 64) return 1;          // K = 64 nk channels, all inside the source row
    if (a.Nstore < 96) return 1;                                             // narrow outputs: the 32-column tile of the generic kernel
    const bool wide = a.Nstore > 128 && mode != 2;                           // GENIE_GEMM_PW=2: always the 256 x 128 tile
    GemmArgs p;
    p.g = a;
    p.g.tiles_m = cdiv(a.M, 256);
    p.g.tiles_n = cdiv(a.Nstore, wide ? 256 : 128);
    p.nkt = wide ? d->nk * 2 : d->nk;
    const long long ntiles = (long long)p.g.tiles_m * p.g.tiles_n;
    if (ntiles < 160 || ntiles >= (1ll << 30)) return 1;                     // few tiles: split-K of the generic kernel fills the chip
    p.ntiles = (int)ntiles;
    constexpr int lds = 3 * (256 * 128 + 128 * 128) + 1024, lds_wide = 4 * (256 * 64 + 256 * 64) + 1024;
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_pw_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_pw256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_wide);
        if (e != hipSuccess) {
            genie_set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return GENIE_ERR_HIP;
        }
        configured = true;
    }
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
            genie_set_error("hipGetDeviceProperties failed");
            return GENIE_ERR_HIP;
        }
        ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int grid = p.ntiles < ncu ? p.ntiles : ncu;
    genie_note_variant(GENIE_VARIANT_GEMM_PW);
    if (wide) hipLaunchKernelGGL(gemm_pw256_kernel, dim3(grid), dim3(512), lds_wide, s, p);
    else hipLaunchKernelGGL(gemm_pw_kernel, dim3(grid), dim3(512), lds, s, p);
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}
