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
    int nkt;            // K tiles (64 wide for the 256 x 128 tile, 32 wide for the 256 x 256 tile)
    int ntiles;         // work items: output tiles x splits
    int splits;         // 256 x 256 tile only: K splits per output tile (1: none)
    int kt_per_split;
    int dbg;
    int pair;           // interleaved column order of a wave's two MFMA tiles + 16-byte stores (gemm_epilogue_pair)
    int m_fast;         // 256 x 256 tile only: row tiles run fastest in the work order
};

// Epilogue of the two GEMM kernels when the destination is plain (no shuffle / bias permutation, 8-channel aligned rows): the wave's
// two 32-column MFMA tiles are INTERLEAVED in output space -- MFMA tile j, column c is output column 8 (c >> 2) + 4 j + (c & 3) of the
// wave's 64 (the B fragments are read from the matching weight rows) -- so after the 4 x 4 lane-quad transposes a lane holds 8
// consecutive channels of one row: 16-byte stores, 8 quads x 16 B = one whole 128-byte line per row and instruction.  (The shared
// epilogue writes 8-byte pieces, two instructions per line: the K = 512 vocabulary head spent as long storing as multiplying.)
template <int BMROWS, int TM>
__device__ __forceinline__ void gemm_epilogue_pair(const IgemmArgs& a, f32x16_t (&acc)[TM][2], char* smem, int m0, int nw0, int wm, int tid, int lane) {
    const int khalf = lane >> 5, jq = lane & 3;
    int* rowoff = reinterpret_cast<int*>(smem);
    if (tid < BMROWS) {
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
    const int n8 = nw0 + 8 * ((lane & 31) >> 2);                              // first of this lane's 8 output columns
    const bool colok = n8 < a.Nstore;
    float bias8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bias8[e] = (a.bias && colok && n8 + e < a.Ncols) ? a.bias[n8 + e] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        u32x4_t rres[4];
        if (a.resid) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ro = rowoff[wm * (TM * 32) + i * 32 + 8 * g + 4 * khalf + jq];
                rres[g] = u32x4_t{0u, 0u, 0u, 0u};
                if (ro >= 0 && colok) rres[g] = *reinterpret_cast<const u32x4_t*>(a.resid + (unsigned)ro + n8);
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 * j + e] = acc[i][j][4 * g + e];
#pragma unroll
                for (int k = 0; k < 4; k += 2) {                               // partner lane ^ 1 swaps the off-diagonal of each 2 x 2
                    const float send = (jq & 1) ? v[4 * j + k] : v[4 * j + k + 1];
                    const float recv = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), 0xB1, 0xF, 0xF, true));
                    if (jq & 1) v[4 * j + k] = recv; else v[4 * j + k + 1] = recv;
                }
#pragma unroll
                for (int k = 0; k < 2; ++k) {                                  // partner lane ^ 2
                    const float send = (jq & 2) ? v[4 * j + k] : v[4 * j + k + 2];
                    const float recv = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), 0x4E, 0xF, 0xF, true));
                    if (jq & 2) v[4 * j + k] = recv; else v[4 * j + k + 2] = recv;
                }
            }
            // v[0..7] = row (8 g + 4 khalf + jq), columns n8 .. n8 + 7
            const int ro = rowoff[wm * (TM * 32) + i * 32 + 8 * g + 4 * khalf + jq];
            if (ro < 0 || !colok) continue;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += bias8[e];
            if (a.resid) {
                float rf[8];
                unpack8(rres[g], rf);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += rf[e];
            }
            if (a.act == 1) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
            }
            *reinterpret_cast<u32x4_t*>(a.dst + (unsigned)ro + n8) = pack8(v);
        }
    }
}

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
            const int lc = (tid & 7) ^ (p.pair ? (((row >> 2) & 6) | ((row >> 1) & 1)) : ((row >> 1) & 7));   // see b_off
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
        // p.pair: MFMA tile j, column c computes output column 8 (c >> 2) + 4 j + (c & 3) of the wave's 64 (see gemm_epilogue_pair)
        const int row = wn * 64 + (p.pair ? 8 * ((lane & 31) >> 2) + 4 * j + (lane & 3) : j * 32 + (lane & 31));
        // swizzle key: the 16 rows a 16-lane group reads must take all 8 values with both row parities -- consecutive rows: (row >> 1) & 7;
        // interleaved rows 8 g + 4 j + q: (g & 3, q >> 1)
        const int key = p.pair ? (((row >> 2) & 6) | ((row >> 1) & 1)) : ((row >> 1) & 7);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) b_off[ks][j] = (unsigned)(A_BYTES + row * 128 + (((ks * 2 + khalf) ^ key) << 4));
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
        if (p.pair) gemm_epilogue_pair<BM, TM>(a, acc, epi, m0, n0 + wn * 64, wm, tid, lane);
        else igemm_epilogue<BM, TM, TN>(a, acc, epi, m0, n0, wm, wn, tid, lane);
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

    // work item = (tile, K split); p.splits == 1: the whole reduction.  Consecutive ids sit on one XCD at the same time: with splits the
    // tiles of ONE K range are consecutive (they share that range of both operands through the XCD's L2); p.m_fast makes the row
    // tiles of one column tile consecutive (few rows, many columns -- the vocabulary head: every weight tile is fetched once
    // instead of once per row tile)
    auto tile_of = [&](int it, int& m0, int& n0, int& k0, int& kn, int& split) -> bool {
        const int base = it * G;
        const int left = p.ntiles - base;
        if (left <= 0) return false;
        const int g = left < G ? left : G;
        if ((int)blockIdx.x >= g) return false;
        int id = base + xcd_tile_id(g, blockIdx.x);
        split = 0; k0 = 0; kn = p.nkt;
        if (p.splits > 1) {
            const int tiles = a.tiles_m * a.tiles_n;
            split = id / tiles; id -= split * tiles;
            k0 = split * p.kt_per_split;
            kn = p.nkt - k0 < p.kt_per_split ? p.nkt - k0 : p.kt_per_split;
        }
        if (p.m_fast) {
            m0 = (id % a.tiles_m) * BM;
            n0 = (id / a.tiles_m) * BN;
        } else {
            m0 = (id / a.tiles_n) * BM;
            n0 = (id % a.tiles_n) * BN;
        }
        return true;
    };

    int l_it = 0, l_k = 0, l_k0 = 0, l_kn = 0;
    bool l_live;
    unsigned a_src[A_LOADS];
    const bf16_t* b_src[B_LOADS];
    auto loader_setup = [&]() {
        int m0, n0, sp;
        l_live = tile_of(l_it, m0, n0, l_k0, l_kn, sp);
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
            const int lc = (lane & 3) ^ ((p.pair ? row >> 3 : row >> 2) & 3);                                   // see b_off
            const int n = n0 + row;
            b_src[j] = n < a.Ncols ? a.wgt + (size_t)n * a.w_row_stride + lc * 8 : nullptr;
        }
    };
    auto issue = [&](int slot) {
        char* abuf = ring + slot * STAGE;
        char* bbuf = abuf + A_BYTES;
        const int kofs = (l_k0 + l_k) * 32;
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
        if (l_live && ++l_k == l_kn) {
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
        const int row = wn * 64 + (p.pair ? 8 * ((lane & 31) >> 2) + 4 * j + (lane & 3) : j * 32 + (lane & 31));
        const int key = (p.pair ? row >> 3 : row >> 2) & 3;                   // interleaved rows 8 g + 4 j + q: the key is g & 3
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) b_off[ks][j] = (unsigned)(A_BYTES + row * 64 + (((ks * 2 + khalf) ^ key) << 4));
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
        int m0, n0, k0, kn, split;
        if (!tile_of(it, m0, n0, k0, kn, split)) break;
        f32x16_t acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int k = 0; k < kn; ++k) {
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
        if (p.dbg & 1) {
            if (acc[0][0][0] == 1234.5f) a.dst[0] = 0;
        } else if (p.splits > 1) {
            igemm_store_partials<TM, TN>(a, acc, split, m0, n0, wm, wn, lane);      // fp32 partial tile; igemm_splitk_finish sums them
        } else {
            if (p.pair) gemm_epilogue_pair<BM, TM>(a, acc, epi, m0, n0 + wn * 64, wm, tid, lane);
            else igemm_epilogue<BM, TM, TN>(a, acc, epi, m0, n0, wm, wn, tid, lane);
            __syncthreads();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

int genie_igemm_splitk_finish(const IgemmArgs& a, hipStream_t s);               // conv_igemm.hip

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
    p.g.gn_sums = nullptr; p.g.gnb_x = nullptr;                              // no GroupNorm fusion in these epilogues
    p.g.tiles_m = cdiv(a.M, 256);
    p.g.tiles_n = cdiv(a.Nstore, wide ? 256 : 128);
    p.nkt = wide ? d->nk * 2 : d->nk;
    long long ntiles = (long long)p.g.tiles_m * p.g.tiles_n;
    p.splits = 1; p.kt_per_split = p.nkt;
    p.dbg = getenv("GENIE_GEMM_DBG") ? atoi(getenv("GENIE_GEMM_DBG")) : 0;
    static const int pair_on = getenv("GENIE_GEMM_PW_PAIR") ? atoi(getenv("GENIE_GEMM_PW_PAIR")) : 1;
    p.pair = pair_on && a.perm_f == 1 && a.shuf_c >= a.Nstore && a.shuf_q == 1 && a.shuf_r == 1 && (a.Nstore & 7) == 0 && (a.Cd & 7) == 0;
    static const int mfast = getenv("GENIE_GEMM_PW_MFAST") ? atoi(getenv("GENIE_GEMM_PW_MFAST")) : 1;
    p.m_fast = mfast && wide && p.g.tiles_m <= 32 && p.g.tiles_n > p.g.tiles_m;
    if (ntiles >= (1ll << 30)) return 1;
    if (ntiles < 160) {
        // few output tiles.  A long reduction over a wide output (the vocabulary head's backward-data pass: 3072 x 512 outputs, K = 2^18)
        // is cut into K splits of the 256 x 256 tile, about one work item per CU, fp32 partial tiles in the caller's scratch;
        // everything else goes to the generic kernel's split-K.
        static const int pw_split = getenv("GENIE_GEMM_PW_SPLITK") ? atoi(getenv("GENIE_GEMM_PW_SPLITK")) : 1;
        if (!pw_split || !wide || !d->splitk_ws || p.nkt < 256) return 1;
        const long long per = (long long)a.M * a.Nstore * 4;
        long long sk = 256 / ntiles;
        if (sk > p.nkt / 64) sk = p.nkt / 64;                                // >= 64 K tiles (2048 k) per split
        if (sk * per > d->splitk_ws_bytes) sk = d->splitk_ws_bytes / per;
        if (sk < 2) return 1;
        p.kt_per_split = cdiv(p.nkt, (int)sk);
        p.splits = cdiv(p.nkt, p.kt_per_split);
        p.g.split_k = p.splits;
        p.g.ws = (float*)d->splitk_ws;
        p.g.ws_ld = a.Nstore;
        ntiles *= p.splits;
        p.pair = 0;                                                          // the partial-tile store uses the plain column order
    }
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
    if (p.splits > 1) return genie_igemm_splitk_finish(p.g, s);
    return GENIE_OK;
}
