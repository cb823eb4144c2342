// Gather-GEMM for stride-1 convolutions whose taps come in kw-triples (dw = -1, 0, +1): Conv3d k=(*,*,3) forward and its
// backward-data pass (reference: nn.Conv3d(k=3, padding=1) in VideoResidualBlock video.py:580-620, CausalConv3d video.py:178-192,
// the ST-block FFN conv attention.py:429-438, the DepthToSpaceTimeUpsample conv video.py:396-402).
//
// Why a second kernel: at a 128x128x64 tile the generic kernel moves 32 KB through the CU's vector-memory path per 16 MFMAs
// per wave, i.e. 64 B/clk/CU at full MFMA rate -- exactly that path's peak, so it tops out near 35 % of the MFMA roofline.
// The three taps of a kw-triple read the SAME activation rows shifted by one pixel, so this kernel stages each activation
// tile ONCE per triple as an LDS image with explicit zero columns left and right of every image row
//
//     image row  (hl, wp) = hl * (W + 2) + wp,   wp = w + 1,   wp = 0 and wp = W + 1 are zero (the conv's w padding)
//
// and the three taps read it at row offsets +0 / +1 / +2: no masks in the MFMA loop, a third of the activation traffic.
// With BM = 256 rows per block the weight tile is amortised over twice the rows: 27 KB per K-tile per 256x128 tile
// (26 B/clk/CU at full rate) instead of 64.
//
//   M-tile  : BM consecutive output pixels = BM / W complete image rows (BM % W == 0, any (n, t, h) per row)
//   K order : for (dt, dh, channel block of 64): one image; for s in 0..2: one 128 x 64 weight tile
//   LDS     : 2 images (1.25 BM rows x 128 B) + 2 weight tiles (16 KB), all DMA'd (global_load_lds_dwordx4);
//             16-B chunk c of row r sits at slot c ^ ((r >> 1) & 7) (source-side swizzle, conflict-free ds_read_b128)
//   waves   : BM / 64 x 2, each 64 x 64 (2 x 2 v_mfma_f32_32x32x16_bf16 tiles), fp32 accumulate
//   pipeline: the next weight tile and 1-2 glds rounds of the next image are issued before the MFMAs of the current K-tile;
//             waits are counted (s_waitcnt vmcnt(N)) so that image rounds stay in flight across the workgroup barrier
#include "igemm3_common.h"

#ifndef GENIE_TRI_WIDE_DEFAULT
#define GENIE_TRI_WIDE_DEFAULT 1          // 256 x 256 kw-triple tiles where Cout >= 256 (measured +5...9 % per layer; GENIE_TRI_WIDE=0 switches them off)
#endif

static __device__ __attribute__((aligned(256))) uint32_t g_zero_page3[64];

template <int BM, bool PIPE>
__global__ void __launch_bounds__(BM * 2) igemm3_kernel(const Igemm3Args p, const GenieTriStep* __restrict__ steps) {
    // `steps` is a separate __restrict__ argument so that the (wave-uniform) table reads become scalar loads
    constexpr int BN = 128, NT = BM * 2, NWAVE = NT / 64;
    constexpr int WN = 2;                               // wave grid (BM / 64) x 2, wave tile 64 x 64
    constexpr int TM = 2, TN = 2;
    constexpr int RPR = NT / 8;                         // LDS rows written by one glds round of the whole block
    constexpr int A_ROUNDS = 5;                         // image rows <= 1.25 BM for W >= 8
    constexpr int A_BYTES = A_ROUNDS * RPR * 128;
    constexpr int B_BYTES = BN * 128;
    constexpr int B_LOADS = BN / RPR;                   // 2 (512 threads) or 4 (256 threads)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const IgemmArgs& a = p.g;
    int nsteps = p.nsteps;
    if (a.split_k > 1) {                                // blockIdx.y = K split: a contiguous range of the step table
        const int s0 = (int)blockIdx.y * a.chunks_per_split;
        steps += s0;
        nsteps = nsteps - s0 < a.chunks_per_split ? nsteps - s0 : a.chunks_per_split;
    }

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int W = a.Wo, H = a.Ho, T = a.To, WP = p.WP;

    int tile_m, tile_n;
    {
        const int id = xcd_tile_id(a.tiles_m * a.tiles_n, blockIdx.x);
        tile_n = id % a.tiles_n;
        tile_m = id / a.tiles_n;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    if (a.split_k <= 1) {
        int first;
        tri_trim_range(steps, nsteps, m0, BM, a.M, H, W, T, first, nsteps);
        steps += first;
    }
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page3);

    // ---- per-thread staging state: the image rows this lane fills (one per round) ----
    // a_base: element offset of the row's home pixel (+ this lane's logical 16-B chunk); a_th: (t << 16) | h of the home pixel,
    // or a value that fails every range check for rows that are padding / outside the tile / beyond M.
    unsigned a_base[A_ROUNDS];
    int a_th[A_ROUNDS];
    const int row0_id = m0 / W;                         // (n, t, h) row index of the tile's first image row
#pragma unroll
    for (int i = 0; i < A_ROUNDS; ++i) {
        const int r = i * RPR + (tid >> 3);
        const int lc = (tid & 7) ^ ((r >> 1) & 7);
        const int hl = r / WP, w = r - hl * WP - 1;
        const int rowid = row0_id + hl;
        const long long m = (long long)rowid * W + w;
        const bool valid = r < p.img_rows && w >= 0 && w < W && m < a.M;
        a_base[i] = valid ? (unsigned)m * (unsigned)a.Cs + lc * 8 : 0u;
        a_th[i] = valid ? ((((rowid / H) % T) << 16) | (rowid % H)) : (int)0x80000000;
    }
    bool b_ok[B_LOADS];
    const bf16_t* b_ptr[B_LOADS];
#pragma unroll
    for (int j = 0; j < B_LOADS; ++j) {
        const int row = j * RPR + (tid >> 3);
        const int lc = (tid & 7) ^ ((row >> 1) & 7);
        const int n = n0 + row;
        b_ok[j] = n < a.Ncols;
        const int wr = b_ok[j] ? (a.perm_f > 1 ? (n % a.perm_c) * a.perm_f + n / a.perm_c : n) : 0;
        b_ptr[j] = a.wgt + (size_t)wr * a.w_row_stride + lc * 8;
    }

    const int dbg = p.dbg;
    auto stage_a = [&](const GenieTriStep& e, bool live, int i, char* abuf) {
        if (dbg & 4) return;
        const int t = (a_th[i] >> 16) + e.dt, h = (a_th[i] & 0xffff) + e.dh;
        const bool ok = live & ((unsigned)t < (unsigned)T) & ((unsigned)h < (unsigned)H);
        const bf16_t* q = a.src + (int)(a_base[i] + (unsigned)e.a_delta);
        q = ok ? q : zero;
        __builtin_amdgcn_global_load_lds(GLB_PTR(q), LDS_PTR(abuf + (i * NWAVE + wave) * 1024), 16, 0, 0);
    };
    auto stage_b = [&](int wofs, bool live, char* bbuf) {
        if (dbg & 4) return;
#pragma unroll
        for (int j = 0; j < B_LOADS; ++j) {
            const bf16_t* q = (live & b_ok[j]) ? b_ptr[j] + wofs : zero;
            __builtin_amdgcn_global_load_lds(GLB_PTR(q), LDS_PTR(bbuf + (j * NWAVE + wave) * 1024), 16, 0, 0);
        }
    };

    // ---- fragment read addresses ----
    int a_row[TM], b_rd[TN], b_sw[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int pl = wm * 64 + i * 32 + (lane & 31);  // tile-local pixel
        const int hl = pl / W;
        a_row[i] = hl * WP + (pl - hl * W);             // image row of (pixel, dw = -1); + s for dw = s - 1
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int row = wn * 64 + j * 32 + (lane & 31);
        b_rd[j] = row * 128;
        b_sw[j] = (row >> 1) & 7;
    }
    const int khalf = lane >> 5;

    f32x16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto compute = [&](const char* abuf, int s, const char* bbuf) {
        int a_rd[TM], a_sw[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = a_row[i] + s;
            a_rd[i] = row * 128;
            a_sw[i] = (row >> 1) & 7;
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8_t af[TM], bfr[TN];
            const int lc = ks * 2 + khalf;
            if (dbg & 8) {
#pragma unroll
                for (int i = 0; i < TM; ++i) asm volatile("" : "=v"(af[i]));
#pragma unroll
                for (int j = 0; j < TN; ++j) asm volatile("" : "=v"(bfr[j]));
            } else {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const bf16x8_t*>(abuf + a_rd[i] + ((lc ^ a_sw[i]) << 4));
#pragma unroll
                for (int j = 0; j < TN; ++j) bfr[j] = *reinterpret_cast<const bf16x8_t*>(bbuf + b_rd[j] + ((lc ^ b_sw[j]) << 4));
            }
            if (dbg & 16) {
#pragma unroll
                for (int i = 0; i < TM; ++i) asm volatile("" :: "v"(af[i]));
#pragma unroll
                for (int j = 0; j < TN; ++j) asm volatile("" :: "v"(bfr[j]));
            } else {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            }
        }
    };

    // end of a K-tile: everything issued before the last KEEP glds of this wave has landed, then the workgroup meets
    auto sync_keep2 = [&]() {
        if constexpr (PIPE) {
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            if (!(dbg & 32)) __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        } else {
            __syncthreads();
        }
    };
    auto sync_all = [&]() {
        if constexpr (PIPE) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (!(dbg & 32)) __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        } else {
            __syncthreads();
        }
    };

    char* const A0 = smem;
    char* const B0 = smem + 2 * A_BYTES;

    // ---- prologue: image 0 and weight tile 0 ----
    {
        const GenieTriStep e = steps[0];
#pragma unroll
        for (int i = 0; i < A_ROUNDS; ++i) stage_a(e, true, i, A0);
        stage_b(e.wofs0, true, B0);
    }
    sync_all();

    int kpar = 0;                                       // parity of the current K-tile (selects the weight buffer)
    for (int i = 0; i < nsteps; ++i) {
        const GenieTriStep cur = steps[i];
        const bool has_next = i + 1 < nsteps;
        const GenieTriStep nxt = steps[has_next ? i + 1 : i];
        char* const acur = A0 + (i & 1) * A_BYTES;
        char* const anxt = A0 + ((i + 1) & 1) * A_BYTES;
        char* const bcur = B0 + kpar * B_BYTES;
        char* const boff = B0 + (kpar ^ 1) * B_BYTES;
        // s = 0 : weight tile (i, 1) and image rounds 0, 1 of step i + 1 go out first
        stage_b(cur.wofs1, true, boff);
        __builtin_amdgcn_sched_barrier(0);               // the counted waits below rely on this issue order
        stage_a(nxt, has_next, 0, anxt);
        stage_a(nxt, has_next, 1, anxt);
        __builtin_amdgcn_sched_barrier(0);
        compute(acur, 0, bcur);
        sync_keep2();
        // s = 1
        stage_b(cur.wofs2, true, bcur);
        __builtin_amdgcn_sched_barrier(0);
        stage_a(nxt, has_next, 2, anxt);
        stage_a(nxt, has_next, 3, anxt);
        __builtin_amdgcn_sched_barrier(0);
        compute(acur, 1, boff);
        sync_keep2();
        // s = 2 : weight tile (i + 1, 0); the image of step i + 1 must be complete before the barrier
        stage_b(nxt.wofs0, has_next, boff);
        stage_a(nxt, has_next, 4, anxt);
        compute(acur, 2, bcur);
        sync_all();
        kpar ^= 1;
    }

    if (a.split_k > 1) {
        igemm_store_partials<TM, TN>(a, acc, blockIdx.y, m0, n0, wm, wn, lane);
        return;
    }
    igemm_epilogue<BM, TM, TN>(a, acc, smem, m0, n0, wm, wn, tid, lane);
}



// ------------------------------------------------------------------------------------------------------------------------------
// Deep-prefetch variant for the 256-row tile (8 waves, 1 block per CU).  Measured on the lock-step kernel above (timing
// ablations, 256->256 @16x32x32, B = 8): MFMAs alone 0.27 ms, the glds stream alone 0.26 ms, everything 0.43 ms -- with the
// weight tile of K-tile k+1 issued at the start of K-tile k and waited for at its end, every K-tile (~0.5 us of MFMA work)
// exposes a loaded L2/MALL round trip (~0.6 us).  Here weight tiles live in a ring of FOUR slots and are issued THREE K-tiles
// ahead; image rounds go out in the first two K-tiles of a triple.  Per K-tile the wave issues one group
//     s = 0: image rounds 0,1,2 then weight tile k+3        (5 glds)
//     s = 1: image rounds 3,4   then weight tile k+3        (4 glds)
//     s = 2:                         weight tile k+3        (2 glds)
// and ends with a counted wait that retires everything up to weight tile k+1 (the tail of group k-2) -- i.e. leaves groups
// k-1 and k in flight: vmcnt(2+5), vmcnt(5+4); at s = 2 the image rounds of group k-1 must have landed as well (the next
// K-tile reads the new image), they precede that group's weight tile, so vmcnt(2+2).  The barrier after the wait publishes the
// landed data to all waves (RAW) and fences the slot that the NEXT group overwrites (WAR: slot (k+4) & 3 = k & 3 was read in
// K-tile k, the spare image was last read in K-tile 3i-1).
// ------------------------------------------------------------------------------------------------------------------------------
// NWAVE = 8: 2 waves per SIMD, 64 x 64 per wave.  NWAVE = 4 (PRE only): ONE wave per SIMD owning 128 x 64 (TM = 4) -- no contention for
// the matrix pipe between co-resident waves, 6 fragment reads per 8 MFMAs instead of 4 per 4, four waves at the barrier instead
// of eight; each thread issues twice the DMA pieces.
// LEAN (round 3, same idea as wgrad3l_kernel in conv_wgrad3.hip): the LDS-DMA pieces are BUFFER loads -- wave-uniform descriptors (the
// image rebased to this tile minus a margin that keeps every tap's offset positive, the weight pack as it is), a per-lane byte offset
// that never changes and a scalar offset from the step table; a piece whose image row (t + dt, h + dh) is outside the clip, a lane
// whose pixel / weight row does not exist and a stage past the last step carry offset 0x80000000 >= num_records and the DMA writes
// zeros.  Replaces ~9 VALU per piece (64-bit address arithmetic, per-lane (t, h) decode, predicates, zero-page select) by one v_or.

template <bool PRE, bool SPLITK, int NWAVE, bool LEAN>
__device__ __forceinline__ void igemm3d_body(const Igemm3Args& p, const GenieTriStep* __restrict__ steps) {
    constexpr int BM = 256, BN = 128, NT = 64 * NWAVE, WN = 2, WM = NWAVE / WN, TM = BM / (WM * 32), TN = 2;
    constexpr int RPR = NT / 8, A_ROUNDS = BM / RPR;     // DMA rounds per image: the 256 REAL pixels (the zero columns are written once, below)
    constexpr int A_BYTES = 320 * 128, B_BYTES = BN * 128;     // an image holds <= 320 rows with its zero columns
    constexpr int B_LOADS = BN / RPR;                   // 2 (8 waves) or 4 (4 waves)
    static_assert(NWAVE == 8 || (NWAVE == 4 && PRE), "the 4-wave form exists for the pre-read schedule only");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const IgemmArgs& a = p.g;
    int nsteps = p.nsteps;
    if (SPLITK) {                                       // blockIdx.y = K split: a contiguous range of the step table
        const int s0 = (int)blockIdx.y * a.chunks_per_split;
        steps += s0;
        nsteps = nsteps - s0 < a.chunks_per_split ? nsteps - s0 : a.chunks_per_split;
    }

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int W = a.Wo, H = a.Ho, T = a.To, WP = p.WP;

    int tile_m, tile_n;
    if (p.xcdcol) {
        // blocks are dealt round-robin to the 8 XCDs: XCD x takes column tile x % tiles_n and every (8 / tiles_n)-th row tile, so the
        // weight rows an XCD streams (1.75 MB for 256 -> 256 x 27 taps) stay in its 4 MB L2 while the activations stream through
        const int per = 8 / a.tiles_n, x = blockIdx.x & 7;
        tile_n = x % a.tiles_n;
        tile_m = (int)(blockIdx.x >> 3) * per + x / a.tiles_n;
        if (tile_m >= a.tiles_m) return;
    } else {
        const int id = xcd_tile_id(a.tiles_m * a.tiles_n, blockIdx.x);
        tile_n = id % a.tiles_n;
        tile_m = tf_remap(id / a.tiles_n, p.tf_T, p.tf_F);
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    if (!SPLITK) {
        int first;
        tri_trim_range(steps, nsteps, m0, BM, a.M, H, W, T, first, nsteps);
        steps += first;
    }
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page3);

    // Image staging as in igemm3w_kernel: only the 256 real pixels are DMA'd (pixel q of image row hl = q / W -> LDS row
    // hl * (W + 2) + q % W + 1); the zero columns left and right of every image row are written once per buffer here.
    unsigned a_base[A_ROUNDS], a_dst[A_ROUNDS];
    int a_th[A_ROUNDS];
    const int row0_id = m0 / W;
#pragma unroll
    for (int i = 0; i < A_ROUNDS; ++i) {
        const int q = i * RPR + (tid >> 3);
        const int hl = q / W, w = q - hl * W;
        const int r = hl * WP + w + 1;                                       // this lane's LDS row
        const int lc = (tid & 7) ^ ((r >> 1) & 7);
        const int rowid = row0_id + hl;
        const long long m = (long long)rowid * W + w;
        const bool valid = m < a.M;
        a_base[i] = valid ? (unsigned)m * (unsigned)a.Cs + lc * 8 : 0u;
        a_th[i] = valid ? ((((rowid / H) % T) << 16) | (rowid % H)) : (int)0x80000000;
        const int q0 = (i * NWAVE + wave) * 8;                               // first pixel of the wave's 1-KiB piece (8 pixels of one image row)
        a_dst[i] = (unsigned)(((q0 / W) * WP + q0 % W + 1) * 128);
    }
    for (int e = tid; e < 2 * 2 * (BM / 8) * 8; e += NT) {                   // (buffer, side, image row, 16-B chunk)
        const int c = e & 7, hl = (e >> 3) % (BM / 8), side = (e >> 3) / (BM / 8) & 1, buf = e / (2 * (BM / 8) * 8);
        if (hl * W < BM) *reinterpret_cast<u32x4_t*>(smem + buf * A_BYTES + (hl * WP + (side ? W + 1 : 0)) * 128 + c * 16) = u32x4_t{0u, 0u, 0u, 0u};
    }
    __syncthreads();
    bool b_ok[B_LOADS];
    const bf16_t* b_ptr[B_LOADS];
#pragma unroll
    for (int j = 0; j < B_LOADS; ++j) {
        const int row = j * RPR + (tid >> 3);
        const int lc = (tid & 7) ^ ((row >> 1) & 7);
        const int n = n0 + row;
        b_ok[j] = n < a.Ncols;
        const int wr = b_ok[j] ? (a.perm_f > 1 ? (n % a.perm_c) * a.perm_f + n / a.perm_c : n) : 0;
        b_ptr[j] = a.wgt + (size_t)wr * a.w_row_stride + lc * 8;
    }
    // LEAN staging state: descriptors, per-lane byte offsets (OOB where the pixel / weight row does not exist), scalar (t, h) of the
    // image row that holds the wave's 8-pixel piece
    const int margin = (3 * H * W + 2 * W) * a.Cs;                          // elements; |a_delta| <= (2 H + 1) W Cs + Cs
    __amdgpu_buffer_rsrc_t rs_a, rs_b;
    uint32_t voff_a[A_ROUNDS], voff_b[B_LOADS];
    int at_s[A_ROUNDS], ah_s[A_ROUNDS];
    if constexpr (LEAN) {
        rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)(a.src + ((long long)m0 * a.Cs - margin)), (short)0, (int)IG3_OOB, 0x00020000);
        rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)a.wgt, (short)0, (int)IG3_OOB, 0x00020000);
#pragma unroll
        for (int i = 0; i < A_ROUNDS; ++i) {
            const int q = i * RPR + (tid >> 3);
            const int hl = q / W, w = q - hl * W;
            const int r = hl * WP + w + 1;
            const int lc = (tid & 7) ^ ((r >> 1) & 7);
            voff_a[i] = (long long)m0 + q < a.M ? (uint32_t)((q * a.Cs + lc * 8) * 2) : IG3_OOB;
            const int rowid = row0_id + ((i * NWAVE + wave) * 8) / W;       // image row of the wave's piece
            at_s[i] = __builtin_amdgcn_readfirstlane((rowid / H) % T);
            ah_s[i] = __builtin_amdgcn_readfirstlane(rowid % H);
        }
#pragma unroll
        for (int j = 0; j < B_LOADS; ++j) voff_b[j] = b_ok[j] ? (uint32_t)((b_ptr[j] - a.wgt) * 2) : IG3_OOB;
    }
    auto stage_a = [&](const GenieTriStep& e, bool live, int i, char* abuf) {
        if constexpr (LEAN) {
            const bool ok = (int)live & (int)((unsigned)(at_s[i] + e.dt) < (unsigned)T) & (int)((unsigned)(ah_s[i] + e.dh) < (unsigned)H);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, LDS_PTR(abuf + a_dst[i]), 16, voff_a[i] | (ok ? 0u : IG3_OOB),
                                                     ok ? (uint32_t)((e.a_delta + margin) * 2) : 0u, 0, 0);
        } else {
            const int t = (a_th[i] >> 16) + e.dt, h = (a_th[i] & 0xffff) + e.dh;
            const bool ok = live & ((unsigned)t < (unsigned)T) & ((unsigned)h < (unsigned)H);
            const bf16_t* q = a.src + (int)(a_base[i] + (unsigned)e.a_delta);
            q = ok ? q : zero;
            __builtin_amdgcn_global_load_lds(GLB_PTR(q), LDS_PTR(abuf + a_dst[i]), 16, 0, 0);
        }
    };
    auto stage_b = [&](int wofs, bool live, char* bbuf) {
#pragma unroll
        for (int j = 0; j < B_LOADS; ++j) {
            if constexpr (LEAN) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, LDS_PTR(bbuf + (j * NWAVE + wave) * 1024), 16, voff_b[j] | (live ? 0u : IG3_OOB),
                                                         live ? (uint32_t)(wofs * 2) : 0u, 0, 0);
            } else {
                const bf16_t* q = (live & b_ok[j]) ? b_ptr[j] + wofs : zero;
                __builtin_amdgcn_global_load_lds(GLB_PTR(q), LDS_PTR(bbuf + (j * NWAVE + wave) * 1024), 16, 0, 0);
            }
        }
    };

    // fragment read offsets inside an image / a weight tile, precomputed for every (shift, k-step): the XOR swizzle makes them
    // non-additive in the k-step, and computing them per read costs ~3 VALU per ds_read (4 VALU per MFMA in total, measured)
    unsigned a_off[3][4][TM], b_off[4][TN];
    const int khalf = lane >> 5;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int pl = wm * (TM * 32) + i * 32 + (lane & 31);
        const int hl = pl / W;
        const int row0 = hl * WP + (pl - hl * W);       // image row of (pixel, dw = -1); + s for dw = s - 1
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int row = row0 + s;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) a_off[s][ks][i] = (unsigned)(row * 128 + (((ks * 2 + khalf) ^ ((row >> 1) & 7)) << 4));
        }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int row = wn * (TN * 32) + j * 32 + (lane & 31);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) b_off[ks][j] = (unsigned)(row * 128 + (((ks * 2 + khalf) ^ ((row >> 1) & 7)) << 4));
    }

    f32x16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto compute = [&](const char* abuf, int s, const char* bbuf) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8_t af[TM], bfr[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const bf16x8_t*>(abuf + a_off[s][ks][i]);
#pragma unroll
            for (int j = 0; j < TN; ++j) bfr[j] = *reinterpret_cast<const bf16x8_t*>(bbuf + b_off[ks][j]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    };
    auto raw_barrier = [&]() {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    char* const A0 = smem;
    char* const B0 = smem + 2 * A_BYTES;                // four weight slots

    // ---- prologue: image 0, weight tiles 0, 1, 2 ----
    {
        const GenieTriStep e = steps[0];
#pragma unroll
        for (int i = 0; i < A_ROUNDS; ++i) stage_a(e, true, i, A0);
        stage_b(e.wofs0, true, B0);
        stage_b(e.wofs1, true, B0 + B_BYTES);
        stage_b(e.wofs2, true, B0 + 2 * B_BYTES);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    raw_barrier();

    if constexpr (PRE) {
        // Variant with the first k-step of the NEXT K-tile read before the barrier and the glds issued in the middle of the MFMA
        // stream: after a barrier every wave starts on MFMAs it already holds operands for, instead of all eight issuing loads
        // and then waiting an LDS round trip while the matrix pipes idle (SQ counters on the plain version: pipes 56 % busy,
        // waves parked 37 % of their cycles).  Publication is one barrier earlier than consumption:
        //   at the barrier ending K-tile k, weight tiles k+1 AND k+2 have landed (tile k+3 is issued during K-tile k), and at
        //   s = 1 the image of the next step as well (all its rounds go out at s = 0) -- so the pre-read for K-tile k+1, issued
        //   before that barrier, only touches data published at barrier k-1.
        //   groups: s = 0: 4 image rounds then tile k+3 (6 glds), s = 1, 2: tile k+3 (2 glds); the wait at the end of K-tile k
        //   leaves exactly group k in flight: vmcnt(6) / vmcnt(2) / vmcnt(2).
        //   WAR: slot (k+3) & 3 = (k-1) & 3 was last read in K-tile k-1 (its pre-read targets slot k & 3); the spare image was last
        //   read in K-tile 3i-1 and is refilled from K-tile 3i on.
        bf16x8_t af0[TM], bf0[TN];
        auto preread = [&](const char* abuf, int s, const char* bbuf) {
#pragma unroll
            for (int i = 0; i < TM; ++i) af0[i] = *reinterpret_cast<const bf16x8_t*>(abuf + a_off[s][0][i]);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf0[j] = *reinterpret_cast<const bf16x8_t*>(bbuf + b_off[0][j]);
        };
        auto mfma4 = [&](const bf16x8_t (&fa)[TM], const bf16x8_t (&fb)[TN]) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        };
        auto read_ks = [&](const char* abuf, int s, const char* bbuf, int ks, bf16x8_t (&fa)[TM], bf16x8_t (&fb)[TN]) {
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const bf16x8_t*>(abuf + a_off[s][ks][i]);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const bf16x8_t*>(bbuf + b_off[ks][j]);
        };
        preread(A0, 0, B0);
        int k = 0;
        for (int i = 0; i < nsteps; ++i, k += 3) {
            const bool has_next = i + 1 < nsteps;
            const GenieTriStep nxt = steps[has_next ? i + 1 : i];
            char* const acur = A0 + (i & 1) * A_BYTES;
            char* const anxt = A0 + ((i + 1) & 1) * A_BYTES;
#define GENIE_KTILE(S, ISSUE, NEXT_A, NEXT_S, WAITN)                                                              \
            {                                                                                                    \
                const char* bcur = B0 + ((k + S) & 3) * B_BYTES;                                                 \
                const char* bnext = B0 + ((k + S + 1) & 3) * B_BYTES;                                            \
                bf16x8_t fa1[TM], fb1[TN], fa2[TM], fb2[TN];                                                     \
                read_ks(acur, S, bcur, 1, fa1, fb1);                                                             \
                mfma4(af0, bf0);                                                                                 \
                __builtin_amdgcn_sched_barrier(0);                                                               \
                ISSUE                                                                                            \
                __builtin_amdgcn_sched_barrier(0);                                                               \
                read_ks(acur, S, bcur, 2, fa2, fb2);                                                             \
                mfma4(fa1, fb1);                                                                                 \
                __builtin_amdgcn_sched_barrier(0);                                                               \
                read_ks(acur, S, bcur, 3, fa1, fb1);                                                             \
                mfma4(fa2, fb2);                                                                                 \
                __builtin_amdgcn_sched_barrier(0);                                                               \
                preread(NEXT_A, NEXT_S, bnext);                                                                  \
                mfma4(fa1, fb1);                                                                                 \
                __builtin_amdgcn_sched_barrier(0);                                                               \
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(WAITN) : "memory");                                    \
                raw_barrier();                                                                                   \
            }
            GENIE_KTILE(0,
                        _Pragma("unroll") for (int ar = 0; ar < A_ROUNDS; ++ar) stage_a(nxt, has_next, ar, anxt);
                        __builtin_amdgcn_sched_barrier(0);
                        stage_b(nxt.wofs0, has_next, B0 + ((k + 3) & 3) * B_BYTES);,
                        acur, 1, A_ROUNDS + B_LOADS)
            GENIE_KTILE(1, stage_b(nxt.wofs1, has_next, B0 + ((k + 4) & 3) * B_BYTES);, acur, 2, B_LOADS)
            GENIE_KTILE(2, stage_b(nxt.wofs2, has_next, B0 + ((k + 5) & 3) * B_BYTES);, anxt, 0, B_LOADS)
#undef GENIE_KTILE
        }
    } else {
        int k = 0;                                          // K-tile index (3 i + s); weight tile k sits in slot k & 3
        for (int i = 0; i < nsteps; ++i, k += 3) {
            const GenieTriStep cur = steps[i];
            const bool has_next = i + 1 < nsteps;
            const GenieTriStep nxt = steps[has_next ? i + 1 : i];
            char* const acur = A0 + (i & 1) * A_BYTES;
            char* const anxt = A0 + ((i + 1) & 1) * A_BYTES;
            // ---- s = 0: group = image rounds 0-2 of step i+1, then weight tile k+3 = (i+1, 0) ----
            stage_a(nxt, has_next, 0, anxt);
            stage_a(nxt, has_next, 1, anxt);
            stage_a(nxt, has_next, 2, anxt);
            __builtin_amdgcn_sched_barrier(0);              // the counted waits rely on this issue order
            stage_b(nxt.wofs0, has_next, B0 + ((k + 3) & 3) * B_BYTES);
            __builtin_amdgcn_sched_barrier(0);
            compute(acur, 0, B0 + (k & 3) * B_BYTES);
            asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            raw_barrier();
            // ---- s = 1: image round 3, then weight tile k+4 = (i+1, 1) ----
            stage_a(nxt, has_next, 3, anxt);
            __builtin_amdgcn_sched_barrier(0);
            stage_b(nxt.wofs1, has_next, B0 + ((k + 4) & 3) * B_BYTES);
            __builtin_amdgcn_sched_barrier(0);
            compute(acur, 1, B0 + ((k + 1) & 3) * B_BYTES);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            raw_barrier();
            // ---- s = 2: weight tile k+5 = (i+1, 2); the new image must have landed before the barrier ----
            stage_b(nxt.wofs2, has_next, B0 + ((k + 5) & 3) * B_BYTES);
            __builtin_amdgcn_sched_barrier(0);
            compute(acur, 2, B0 + ((k + 2) & 3) * B_BYTES);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            raw_barrier();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    raw_barrier();

    if (SPLITK) {
        igemm_store_partials<TM, TN>(a, acc, blockIdx.y, m0, n0, wm, wn, lane);
        return;
    }
    igemm_epilogue<BM, TM, TN>(a, acc, smem, m0, n0, wm, wn, tid, lane);
}

template <bool PRE, bool SPLITK, int NWAVE>
__global__ void __launch_bounds__(64 * NWAVE) igemm3d_kernel(const Igemm3Args p, const GenieTriStep* __restrict__ steps) {
    igemm3d_body<PRE, SPLITK, NWAVE, false>(p, steps);
}
// the default schedule (pre-read, no K split, 8 waves) with LEAN staging: its own kernel name, so that traces keep the two apart
__global__ void __launch_bounds__(512) igemm3dl_kernel(const Igemm3Args p, const GenieTriStep* __restrict__ steps) {
    igemm3d_body<true, false, 8, true>(p, steps);
}

int genie_igemm_splitk_finish(const IgemmArgs& a, hipStream_t s);      // conv_igemm.hip

// ------------------------------------------------------------------------------------------------------------------------------
// 256 x 256 tile for layers with >= 256 output channels (GENIE_TRI_WIDE=1 / tri_flags bit 10).  Per 64-deep K-tile the 256 x 128 kernel
// above moves 13.3 KB of image + 16 KB of weights through L2 -> LDS and reads 16 fragments per 16 MFMAs; with 256 columns per block the
// image is shared by twice the MFMAs: 13.3 + 32 KB per 2x the work (-23 % L2 -> LDS bytes per MFMA) and a 64 x 128 wave tile needs 6
// fragment reads per 8 MFMAs instead of 4 per 4 (-25 % LDS reads).  LDS stays at 144 KB by cutting the weight tiles in half along K:
// 256 rows x 64 B (32 channels), ring of FOUR 16-KB slots issued THREE half-tiles ahead, 64-B rows with the 4-slot XOR swizzle of
// gemm_pw256_kernel.  Schedule (pre-read form): half-tile k = 6 i + 2 s + h uses k-steps 2h, 2h + 1 of image i at shift s;
//   during half-tile k: read k-step 2h+1, MFMAs of k-step 2h (pre-read), issue group g(k) = image rounds of step i+1 ([2,1,1,1,0,0]
//   over the six half-tiles) + weight half-tile k+3 (2 glds), pre-read k-step 0 of half-tile k+1, MFMAs of k-step 2h+1,
//   s_waitcnt vmcnt(|g(k)|) -> everything issued before half-tile k has landed -> barrier publishes it.
//   RAW: half-tile k+1's weights were issued in k-2 and published at barrier k-1, before its pre-read in k; the image of step i+1 is
//   complete in group 6i+3, published at barrier 6i+4, pre-read in 6i+5.  WAR: slot (k+3)&3 was last read in half-tile k-1; the
//   spare image was last read in half-tile 6i-1.
// ------------------------------------------------------------------------------------------------------------------------------
// VAR (LEAN only): 0 = production; 2 / 3 / 4 = timing ablations with WRONG results (no DMA in the loop / no MFMA / DMA + barriers only;
// GENIE_TRI_VAR, profiles/r03_igemm3w_ablation.log).  Two schedule changes were measured on top of VAR 0 and dropped:
// waves 4..7 issuing their DMA group at the start of the half-tile (-0.7 %), and the fragment reads in first-use order interleaved with the
// MFMAs by sched_group_barrier instead of the compiler's order, which sinks the pre-read of the next half-tile to just in front of the
// barrier (+-0): the kernel runs at the chip's POWER cap, not at an issue limit.
template <bool SPLITK, bool LEAN = false, int VAR = 0>
__global__ void __launch_bounds__(512) igemm3w_kernel(const Igemm3Args p, const GenieTriStep* __restrict__ steps) {
    constexpr int BM = 256, BN = 256, NWAVE = 8, WN = 2, TM = 2, TN = 4;
    constexpr int RPR = 64, A_ROUNDS = 4;                                    // DMA rounds per image: the 256 REAL pixels (zero columns: below)
    constexpr int A_BYTES = 5 * RPR * 128, B_BYTES = BN * 64;               // 40 KB images (<= 320 rows with the zero columns), 16 KB weight half-tiles
    constexpr int B_LOADS = 2;                                               // 16 rows x 64 B per wave instruction, 256 rows / 8 waves
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const IgemmArgs& a = p.g;
    int nsteps = p.nsteps;
    if (SPLITK) {
        const int s0 = (int)blockIdx.y * a.chunks_per_split;
        steps += s0;
        nsteps = nsteps - s0 < a.chunks_per_split ? nsteps - s0 : a.chunks_per_split;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int W = a.Wo, H = a.Ho, T = a.To, WP = p.WP;
    int tile_m, tile_n;
    {
        const int id = xcd_tile_id(a.tiles_m * a.tiles_n, blockIdx.x);
        tile_n = id % a.tiles_n;
        tile_m = tf_remap(id / a.tiles_n, p.tf_T, p.tf_F);
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    if (!SPLITK) {
        int first;
        tri_trim_range(steps, nsteps, m0, BM, a.M, H, W, T, first, nsteps);
        steps += first;
    }
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page3);

    // Image staging: only the 256 real pixels of the tile are DMA'd (4 rounds of 64 rows); pixel q of image row hl = q / W lands in LDS
    // row hl * (W + 2) + q % W + 1, and the zero columns left and right of every image row are written ONCE per buffer here.  (They used
    // to arrive from a zero page with every image: a fifth round, 20 % of the image pieces.)  A wave's 1-KiB piece = 8 consecutive
    // pixels of one image row (W >= 8).
    unsigned a_base[A_ROUNDS], a_dst[A_ROUNDS];
    int a_th[A_ROUNDS];
    const int row0_id = m0 / W;
#pragma unroll
    for (int i = 0; i < A_ROUNDS; ++i) {
        const int q = i * RPR + (tid >> 3);
        const int hl = q / W, w = q - hl * W;
        const int r = hl * WP + w + 1;                                       // this lane's LDS row
        const int lc = (tid & 7) ^ ((r >> 1) & 7);
        const int rowid = row0_id + hl;
        const long long m = (long long)rowid * W + w;
        const bool valid = m < a.M;
        a_base[i] = valid ? (unsigned)m * (unsigned)a.Cs + lc * 8 : 0u;
        a_th[i] = valid ? ((((rowid / H) % T) << 16) | (rowid % H)) : (int)0x80000000;
        const int q0 = (i * NWAVE + wave) * 8;                               // first pixel of the wave's piece
        a_dst[i] = (unsigned)(((q0 / W) * WP + q0 % W + 1) * 128);
    }
    for (int e = tid; e < 2 * 2 * (BM / 8) * 8; e += 512) {                  // (buffer, side, image row, 16-B chunk); image rows beyond BM / W: skipped
        const int c = e & 7, hl = (e >> 3) % (BM / 8), side = (e >> 3) / (BM / 8) & 1, buf = e / (2 * (BM / 8) * 8);
        if (hl * W < BM) *reinterpret_cast<u32x4_t*>(smem + buf * A_BYTES + (hl * WP + (side ? W + 1 : 0)) * 128 + c * 16) = u32x4_t{0u, 0u, 0u, 0u};
    }
    __syncthreads();
    const bf16_t* b_ptr[B_LOADS];
#pragma unroll
    for (int j = 0; j < B_LOADS; ++j) {
        const int row = (j * NWAVE + wave) * 16 + (lane >> 2);
        const int lc = (lane & 3) ^ ((row >> 2) & 3);
        const int n = n0 + row;
        const int wr = n < a.Ncols ? (a.perm_f > 1 ? (n % a.perm_c) * a.perm_f + n / a.perm_c : n) : -1;
        b_ptr[j] = wr >= 0 ? a.wgt + (size_t)wr * a.w_row_stride + lc * 8 : nullptr;
    }
    // LEAN staging state (see igemm3d_kernel)
    const int margin = (3 * H * W + 2 * W) * a.Cs;
    __amdgpu_buffer_rsrc_t rs_a, rs_b;
    uint32_t voff_a[A_ROUNDS], voff_b[B_LOADS];
    int at_s[A_ROUNDS], ah_s[A_ROUNDS];
    if constexpr (LEAN) {
        rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)(a.src + ((long long)m0 * a.Cs - margin)), (short)0, (int)IG3_OOB, 0x00020000);
        rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)a.wgt, (short)0, (int)IG3_OOB, 0x00020000);
#pragma unroll
        for (int i = 0; i < A_ROUNDS; ++i) {
            const int q = i * RPR + (tid >> 3);
            const int hl = q / W, w = q - hl * W;
            const int r = hl * WP + w + 1;
            const int lc = (tid & 7) ^ ((r >> 1) & 7);
            voff_a[i] = (long long)m0 + q < a.M ? (uint32_t)((q * a.Cs + lc * 8) * 2) : IG3_OOB;
            const int rowid = row0_id + ((i * NWAVE + wave) * 8) / W;
            at_s[i] = __builtin_amdgcn_readfirstlane((rowid / H) % T);
            ah_s[i] = __builtin_amdgcn_readfirstlane(rowid % H);
        }
#pragma unroll
        for (int j = 0; j < B_LOADS; ++j) voff_b[j] = b_ptr[j] ? (uint32_t)((b_ptr[j] - a.wgt) * 2) : IG3_OOB;
    }
    auto stage_a = [&](const GenieTriStep& e, bool live, int i, char* abuf) {
        if constexpr (LEAN) {
            const bool ok = (int)live & (int)((unsigned)(at_s[i] + e.dt) < (unsigned)T) & (int)((unsigned)(ah_s[i] + e.dh) < (unsigned)H);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, LDS_PTR(abuf + a_dst[i]), 16, voff_a[i] | (ok ? 0u : IG3_OOB),
                                                     ok ? (uint32_t)((e.a_delta + margin) * 2) : 0u, 0, 0);
        } else {
            const int t = (a_th[i] >> 16) + e.dt, h = (a_th[i] & 0xffff) + e.dh;
            const bool ok = live & ((unsigned)t < (unsigned)T) & ((unsigned)h < (unsigned)H);
            const bf16_t* q = a.src + (int)(a_base[i] + (unsigned)e.a_delta);
            q = ok ? q : zero;
            __builtin_amdgcn_global_load_lds(GLB_PTR(q), LDS_PTR(abuf + a_dst[i]), 16, 0, 0);
        }
    };
    auto stage_b = [&](int wofs, bool live, char* bbuf) {                      // one 32-channel half of a weight tile
#pragma unroll
        for (int j = 0; j < B_LOADS; ++j) {
            if constexpr (LEAN) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, LDS_PTR(bbuf + (j * NWAVE + wave) * 1024), 16, voff_b[j] | (live ? 0u : IG3_OOB),
                                                         live ? (uint32_t)(wofs * 2) : 0u, 0, 0);
            } else {
                const bf16_t* q = (live && b_ptr[j]) ? b_ptr[j] + wofs : zero;
                __builtin_amdgcn_global_load_lds(GLB_PTR(q), LDS_PTR(bbuf + (j * NWAVE + wave) * 1024), 16, 0, 0);
            }
        }
    };

    unsigned a_off[3][4][TM], b_off[2][TN];
    const int khalf = lane >> 5;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int pl = wm * (TM * 32) + i * 32 + (lane & 31);
        const int hl = pl / W;
        const int row0 = hl * WP + (pl - hl * W);
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int row = row0 + s;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) a_off[s][ks][i] = (unsigned)(row * 128 + (((ks * 2 + khalf) ^ ((row >> 1) & 7)) << 4));
        }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int row = wn * (TN * 32) + j * 32 + (lane & 31);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) b_off[ks][j] = (unsigned)(row * 64 + (((ks * 2 + khalf) ^ ((row >> 2) & 3)) << 4));
    }

    f32x16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    char* const A0 = smem;
    char* const B0 = smem + 2 * A_BYTES;                                      // four weight half-tile slots

    auto read_frag = [&](const char* abuf, int s, int ksa, const char* bbuf, int ksb, bf16x8_t (&fa)[TM], bf16x8_t (&fb)[TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const bf16x8_t*>(abuf + a_off[s][ksa][i]);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const bf16x8_t*>(bbuf + b_off[ksb][j]);
    };
    auto mfma8 = [&](const bf16x8_t (&fa)[TM], const bf16x8_t (&fb)[TN]) {
        if constexpr (VAR == 4) return;
        if constexpr (VAR == 3) {                                            // ablation: the fragments stay live, no matrix work
#pragma unroll
            for (int i = 0; i < TM; ++i) asm volatile("" :: "v"(fa[i]));
#pragma unroll
            for (int j = 0; j < TN; ++j) asm volatile("" :: "v"(fb[j]));
            return;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    };
    auto raw_barrier = [&]() {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    // ---- prologue: image 0, weight half-tiles 0, 1, 2 ----
    {
        const GenieTriStep e = steps[0];
#pragma unroll
        for (int i = 0; i < A_ROUNDS; ++i) stage_a(e, true, i, A0);
        stage_b(e.wofs0, true, B0);
        stage_b(e.wofs0 + 32, true, B0 + B_BYTES);
        stage_b(e.wofs1, true, B0 + 2 * B_BYTES);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    raw_barrier();

    bf16x8_t fa0[TM], fb0[TN];
    read_frag(A0, 0, 0, B0, 0, fa0, fb0);
    int k = 0;                                                                // half-tile counter; weight half-tile k sits in slot k & 3
    for (int i = 0; i < nsteps; ++i, k += 6) {
        const GenieTriStep cur = steps[i];
        const bool has_next = i + 1 < nsteps;
        const GenieTriStep nxt = steps[has_next ? i + 1 : i];
        char* const acur = A0 + (i & 1) * A_BYTES;
        char* const anxt = A0 + ((i + 1) & 1) * A_BYTES;
        // half-tile R (0..5) of this step: shift S = R / 2, channel half HH = R % 2.  NEXT_* = where half-tile R + 1 reads its first k-step.
        // ISSUE = the DMA group of this half-tile (image rounds of step i + 1, then weight half-tile k + R + 3); WAITN = its size.
#define GENIE_HTILE(R, S, HH, ISSUE, NEXT_A, NEXT_S, NEXT_KS, WAITN)                                              \
        {                                                                                                        \
            const char* bcur = B0 + ((k + R) & 3) * B_BYTES;                                                     \
            const char* bnext = B0 + ((k + R + 1) & 3) * B_BYTES;                                                \
            bf16x8_t fa1[TM], fb1[TN];                                                                           \
            if constexpr (VAR < 4) read_frag(acur, S, 2 * HH + 1, bcur, 1, fa1, fb1);                            \
            mfma8(fa0, fb0);                                                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                                   \
            if constexpr (VAR != 2) {                                                                            \
                ISSUE                                                                                            \
            }                                                                                                    \
            __builtin_amdgcn_sched_barrier(0);                                                                   \
            if constexpr (VAR < 4) read_frag(NEXT_A, NEXT_S, NEXT_KS, bnext, 0, fa0, fb0);                       \
            mfma8(fa1, fb1);                                                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                                   \
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(WAITN) : "memory");                                        \
            raw_barrier();                                                                                       \
        }
        // weight half-tile k + R + 3: (step, shift, half) of half-tile index R + 3 -> R = 0: (cur, s1, h1); 1: (cur, s2, h0); 2: (cur, s2, h1);
        //                                                                      3: (nxt, s0, h0); 4: (nxt, s0, h1); 5: (nxt, s1, h0)
        GENIE_HTILE(0, 0, 0,
                    stage_a(nxt, has_next, 0, anxt); stage_a(nxt, has_next, 1, anxt);
                    __builtin_amdgcn_sched_barrier(0);
                    stage_b(cur.wofs1 + 32, true, B0 + ((k + 3) & 3) * B_BYTES);,
                    acur, 0, 2, 4)
        GENIE_HTILE(1, 0, 1,
                    stage_a(nxt, has_next, 2, anxt);
                    __builtin_amdgcn_sched_barrier(0);
                    stage_b(cur.wofs2, true, B0 + ((k + 4) & 3) * B_BYTES);,
                    acur, 1, 0, 3)
        GENIE_HTILE(2, 1, 0,
                    stage_a(nxt, has_next, 3, anxt);
                    __builtin_amdgcn_sched_barrier(0);
                    stage_b(cur.wofs2 + 32, true, B0 + ((k + 5) & 3) * B_BYTES);,
                    acur, 1, 2, 3)
        GENIE_HTILE(3, 1, 1,
                    stage_b(nxt.wofs0, has_next, B0 + ((k + 6) & 3) * B_BYTES);,
                    acur, 2, 0, 2)
        GENIE_HTILE(4, 2, 0,
                    stage_b(nxt.wofs0 + 32, has_next, B0 + ((k + 7) & 3) * B_BYTES);,
                    acur, 2, 2, 2)
        GENIE_HTILE(5, 2, 1,
                    stage_b(nxt.wofs1, has_next, B0 + ((k + 8) & 3) * B_BYTES);,
                    anxt, 0, 0, 2)
#undef GENIE_HTILE
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    raw_barrier();

    if (SPLITK) {
        igemm_store_partials<TM, TN>(a, acc, blockIdx.y, m0, n0, wm, wn, lane);
        return;
    }
    igemm_epilogue<BM, TM, TN>(a, acc, smem, m0, n0, wm, wn, tid, lane);
}

// ------------------------------------------------------------------------------------------------------------------------------
// 256 x 128 tile with 32-channel K-tiles and TWO blocks per CU (round 3): the Cout = 128 layers.  igemm3d_kernel keeps 144 KB of LDS per
// block, so a CU runs ONE block: its prologue (pipeline fill), its epilogue (64 KB of stores, transposes, the residual read) and every
// barrier stall have nothing to hide behind, and with K = 27 x 128 the K loop of these layers is half as long as at 256 channels.  Here
// the K-tile is 32 channels deep (image rows of 64 B: 2 x 20 KB; weight tiles 128 x 64 B in a ring of four 8-KB slots) = 72 KB and the
// accumulators are 64 registers: two blocks (16 waves) share a CU and cover each other's epilogues and barriers.  One step = (dt, dh,
// 32-channel block) = three K-tiles (shifts) of two k-steps = 8 MFMAs per wave between barriers.  LEAN staging (buffer loads).
//   groups per step: s = 0: 2 image pieces of the next step + weight tile k + 3 (3 per wave), s = 1, 2: weight tile k + 3 (1); the wait at
//   the end of K-tile k leaves group k in flight (vmcnt 3 / 1 / 1); publication one barrier ahead of consumption, WAR as in igemm3d_kernel.
// Needs W >= 16 (a 1-KiB piece = 16 pixels of ONE image row), 256 % W == 0.
// ------------------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512, 4) igemm3h_kernel(const Igemm3Args p, const GenieTriStep* __restrict__ steps) {
    constexpr int BM = 256, BN = 128, NWAVE = 8, WN = 2, TM = 2, TN = 2;
    constexpr int A_ROUNDS = 2;                                              // 2 rounds x 8 waves x 16 pixels
    constexpr int A_BYTES = 320 * 64, B_BYTES = BN * 64;                     // <= 320 image rows (W = 16: 16 x 18 = 288), 8-KB weight tiles
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const IgemmArgs& a = p.g;
    int nsteps_h = p.nsteps;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int W = a.Wo, H = a.Ho, T = a.To, WP = p.WP;
    const int id = xcd_tile_id(a.tiles_m * a.tiles_n, blockIdx.x);
    const int tile_n = id % a.tiles_n, tile_m = id / a.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    {
        int first;
        tri_trim_range(steps, nsteps_h, m0, BM, a.M, H, W, T, first, nsteps_h);
        steps += first;
    }
    const int nsub = 2 * nsteps_h;                                           // 32-channel sub-steps
    const int row0_id = m0 / W;

    // ---- staging state ----
    const int margin = (3 * H * W + 2 * W) * a.Cs;
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)(a.src + ((long long)m0 * a.Cs - margin)), (short)0, (int)IG3_OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)a.wgt, (short)0, (int)IG3_OOB, 0x00020000);
    uint32_t voff_a[A_ROUNDS], a_dst[A_ROUNDS], voff_b;
    int at_s[A_ROUNDS], ah_s[A_ROUNDS];
#pragma unroll
    for (int i = 0; i < A_ROUNDS; ++i) {
        const int q0 = (i * NWAVE + wave) * 16;                              // first pixel of the wave's piece: 16 pixels of one image row
        const int q = q0 + (lane >> 2);
        const int hl = q / W, w = q - hl * W;
        const int r = hl * WP + w + 1;                                       // this lane's LDS row
        const int lc = (lane & 3) ^ ((r >> 2) & 3);
        voff_a[i] = (long long)m0 + q < a.M ? (uint32_t)((q * a.Cs + lc * 8) * 2) : IG3_OOB;
        a_dst[i] = (uint32_t)(((q0 / W) * WP + q0 % W + 1) * 64);
        const int rowid = row0_id + q0 / W;
        at_s[i] = __builtin_amdgcn_readfirstlane((rowid / H) % T);
        ah_s[i] = __builtin_amdgcn_readfirstlane(rowid % H);
    }
    {
        const int row = wave * 16 + (lane >> 2);                             // one 1-KiB piece per wave per weight tile: 16 rows x 64 B
        const int lc = (lane & 3) ^ ((row >> 2) & 3);
        const int n = n0 + row;
        const int wr = n < a.Ncols ? (a.perm_f > 1 ? (n % a.perm_c) * a.perm_f + n / a.perm_c : n) : -1;
        voff_b = wr >= 0 ? (uint32_t)(((size_t)wr * a.w_row_stride + lc * 8) * 2) : IG3_OOB;
    }
    // zero columns left and right of every image row, both buffers (written once, never DMA'd)
    for (int e = tid; e < 2 * 2 * (BM / 16) * 4; e += 512) {                 // (buffer, side, image row, 16-B chunk)
        const int c = e & 3, hl = (e >> 2) % (BM / 16), side = ((e >> 2) / (BM / 16)) & 1, buf = e / (2 * (BM / 16) * 4);
        if (hl * W < BM) *reinterpret_cast<u32x4_t*>(smem + buf * A_BYTES + (hl * WP + (side ? W + 1 : 0)) * 64 + c * 16) = u32x4_t{0u, 0u, 0u, 0u};
    }
    __syncthreads();
    // sub-step j = 2 * (table entry) + (32-channel half)
    auto stage_a = [&](int j, int i, char* abuf) {
        const bool live = j < nsub;
        const GenieTriStep e = steps[live ? (j >> 1) : 0];
        const bool ok = (int)live & (int)((unsigned)(at_s[i] + e.dt) < (unsigned)T) & (int)((unsigned)(ah_s[i] + e.dh) < (unsigned)H);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, LDS_PTR(abuf + a_dst[i]), 16, voff_a[i] | (ok ? 0u : IG3_OOB),
                                                 ok ? (uint32_t)((e.a_delta + 32 * (j & 1) + margin) * 2) : 0u, 0, 0);
    };
    auto stage_b = [&](int j, int s, char* bbuf) {                          // weight tile of (sub-step j, shift s)
        const bool live = j < nsub;
        const GenieTriStep e = steps[live ? (j >> 1) : 0];
        const int wofs = (s == 0 ? e.wofs0 : (s == 1 ? e.wofs1 : e.wofs2)) + 32 * (j & 1);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, LDS_PTR(bbuf + wave * 1024), 16, voff_b | (live ? 0u : IG3_OOB), live ? (uint32_t)(wofs * 2) : 0u, 0, 0);
    };

    // fragment read offsets: 64-B rows, 16-B chunk (2 ks + khalf) at slot chunk ^ ((row >> 2) & 3)
    unsigned a_off[3][2][TM], b_off[2][TN];
    const int khalf = lane >> 5;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int pl = wm * (TM * 32) + i * 32 + (lane & 31);
        const int hl = pl / W;
        const int row0 = hl * WP + (pl - hl * W);
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int row = row0 + s;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) a_off[s][ks][i] = (unsigned)(row * 64 + (((ks * 2 + khalf) ^ ((row >> 2) & 3)) << 4));
        }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int row = wn * (TN * 32) + j * 32 + (lane & 31);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) b_off[ks][j] = (unsigned)(row * 64 + (((ks * 2 + khalf) ^ ((row >> 2) & 3)) << 4));
    }

    f32x16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    char* const A0 = smem;
    char* const B0 = smem + 2 * A_BYTES;
    auto read_frag = [&](const char* abuf, int s, int ks, const char* bbuf, bf16x8_t (&fa)[TM], bf16x8_t (&fb)[TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const bf16x8_t*>(abuf + a_off[s][ks][i]);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const bf16x8_t*>(bbuf + b_off[ks][j]);
    };
    auto mfma8 = [&](const bf16x8_t (&fa)[TM], const bf16x8_t (&fb)[TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    };
    auto raw_barrier = [&]() {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // ---- prologue: image 0, weight tiles 0, 1, 2 ----
#pragma unroll
    for (int i = 0; i < A_ROUNDS; ++i) stage_a(0, i, A0);
    stage_b(0, 0, B0);
    stage_b(0, 1, B0 + B_BYTES);
    stage_b(0, 2, B0 + 2 * B_BYTES);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    raw_barrier();

    bf16x8_t fa0[TM], fb0[TN];
    read_frag(A0, 0, 0, B0, fa0, fb0);
    int k = 0;                                                               // K-tile counter (3 j + s); weight tile k sits in slot k & 3
    for (int j = 0; j < nsub; ++j, k += 3) {
        char* const acur = A0 + (j & 1) * A_BYTES;
        char* const anxt = A0 + ((j + 1) & 1) * A_BYTES;
#define GENIE_T_KTILE(S, ISSUE, NEXT_A, NEXT_S, WAITN)                                                            \
        {                                                                                                        \
            const char* bcur = B0 + ((k + S) & 3) * B_BYTES;                                                     \
            const char* bnext = B0 + ((k + S + 1) & 3) * B_BYTES;                                                \
            bf16x8_t fa1[TM], fb1[TN];                                                                           \
            read_frag(acur, S, 1, bcur, fa1, fb1);                                                               \
            mfma8(fa0, fb0);                                                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                                   \
            ISSUE                                                                                                \
            __builtin_amdgcn_sched_barrier(0);                                                                   \
            read_frag(NEXT_A, NEXT_S, 0, bnext, fa0, fb0);                                                       \
            mfma8(fa1, fb1);                                                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                                   \
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(WAITN) : "memory");                                        \
            raw_barrier();                                                                                       \
        }
        // weight tile k + S + 3 = (sub-step j + 1, shift S)
        GENIE_T_KTILE(0,
                      _Pragma("unroll") for (int ar = 0; ar < A_ROUNDS; ++ar) stage_a(j + 1, ar, anxt);
                      __builtin_amdgcn_sched_barrier(0);
                      stage_b(j + 1, 0, B0 + ((k + 3) & 3) * B_BYTES);,
                      acur, 1, 3)
        GENIE_T_KTILE(1, stage_b(j + 1, 1, B0 + ((k + 4) & 3) * B_BYTES);, acur, 2, 1)
        GENIE_T_KTILE(2, stage_b(j + 1, 2, B0 + ((k + 5) & 3) * B_BYTES);, anxt, 0, 1)
#undef GENIE_T_KTILE
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    raw_barrier();
    igemm_epilogue<BM, TM, TN, false>(a, acc, smem, m0, n0, wm, wn, tid, lane);
}

static int launch_igemm3h(const Igemm3Args& p, const GenieTriStep* steps, hipStream_t s) {
    constexpr int lds = 2 * (320 * 64) + 4 * 128 * 64;
    static bool conf = false;
    if (!conf) {
        hipError_t e = hipFuncSetAttribute((const void*)igemm3h_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            genie_set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return GENIE_ERR_HIP;
        }
        conf = true;
    }
    hipLaunchKernelGGL(igemm3h_kernel, dim3(p.g.tiles_m * p.g.tiles_n, 1), dim3(512), lds, s, p, steps);
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}

// The LEAN staging (buffer loads with 32-bit byte offsets) needs the weight pack below 2 GiB; GENIE_TRI_LEAN=0 switches it off (A/B).
static bool ig3_lean_ok(const Igemm3Args& p) {
    static const int on = getenv("GENIE_TRI_LEAN") ? atoi(getenv("GENIE_TRI_LEAN")) : 1;
    return on && p.dbg == 0 && (long long)p.g.Ncols * p.g.w_row_stride * 2 + (1 << 20) < 0x7fffffffll;
}

template <bool SPLITK>
static int launch_igemm3w(const Igemm3Args& p, const GenieTriStep* steps, hipStream_t s) {
    constexpr int lds = 2 * (5 * 64 * 128) + 4 * 256 * 64;
    if (!SPLITK && ig3_lean_ok(p)) {                           // buffer-addressed LDS-DMA form
        static const int var = getenv("GENIE_TRI_VAR") ? atoi(getenv("GENIE_TRI_VAR")) : 0;   // 2..4: timing ablations
        static bool lconf = false;
        const void* fn = var == 2 ? (const void*)igemm3w_kernel<false, true, 2>
                       : var == 3 ? (const void*)igemm3w_kernel<false, true, 3> : var == 4 ? (const void*)igemm3w_kernel<false, true, 4>
                       : (const void*)igemm3w_kernel<false, true, 0>;
        if (!lconf) {
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (e != hipSuccess) {
                genie_set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e));
                return GENIE_ERR_HIP;
            }
            lconf = true;
        }
        const dim3 grid(p.g.tiles_m * p.g.tiles_n, 1);
        switch (var) {
            case 2: hipLaunchKernelGGL((igemm3w_kernel<false, true, 2>), grid, dim3(512), lds, s, p, steps); break;
            case 3: hipLaunchKernelGGL((igemm3w_kernel<false, true, 3>), grid, dim3(512), lds, s, p, steps); break;
            case 4: hipLaunchKernelGGL((igemm3w_kernel<false, true, 4>), grid, dim3(512), lds, s, p, steps); break;
            default: hipLaunchKernelGGL((igemm3w_kernel<false, true, 0>), grid, dim3(512), lds, s, p, steps); break;
        }
        GENIE_CHECK_LAUNCH();
        return GENIE_OK;
    }
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute((const void*)igemm3w_kernel<SPLITK>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            genie_set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return GENIE_ERR_HIP;
        }
        configured = true;
    }
    hipLaunchKernelGGL((igemm3w_kernel<SPLITK>), dim3(p.g.tiles_m * p.g.tiles_n, SPLITK ? p.g.split_k : 1), dim3(512), lds, s, p, steps);
    GENIE_CHECK_LAUNCH();
    if (SPLITK) return genie_igemm_splitk_finish(p.g, s);
    return GENIE_OK;
}



// ------------------------------------------------------------------------------------------------------------------------------
// Persistent form of igemm3d_kernel<true, false>: one block per CU walks its tiles and the K-tile stream of the schedule above
// runs straight through tile boundaries -- during the LAST step of a tile the image and the three weight tiles of the next
// tile's first step are issued (loader state = row addresses / coordinates / weight rows of the tile being LOADED, switched at
// the start of that last step), and the pre-read before the last barrier already holds the next tile's first MFMA operands.
// A 128-channel layer is 18 steps = 54 K-tiles per tile: the pipeline fill (image + three weight tiles, then a full drain) and
// the epilogue were ~10 % of a tile's time with one tile per launch slot.  The epilogue's row offsets live in their own KiB
// behind the rings (the DMA of the next tile is in flight while it runs); its stores only make the counted waits of the next
// K-tiles more conservative (loads retire in order among themselves).
// ------------------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512) igemm3p_kernel(const Igemm3Args p, const GenieTriStep* __restrict__ steps) {
    constexpr int BM = 256, BN = 128, NT = 512, NWAVE = 8, WN = 2, TM = 2, TN = 2;
    constexpr int RPR = NT / 8, A_ROUNDS = 5;
    constexpr int A_BYTES = A_ROUNDS * RPR * 128, B_BYTES = BN * 128;
    constexpr int B_LOADS = BN / RPR;                   // 2
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const IgemmArgs& a = p.g;
    const int nsteps = p.nsteps;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int W = a.Wo, H = a.Ho, T = a.To, WP = p.WP;
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page3);
    const int ntiles = a.tiles_m * a.tiles_n, G = (int)gridDim.x;

    // tile `it` of this block: rounds of G tiles, consecutive ids of a round on one XCD, tile_n fastest (as the one-tile kernel)
    auto tile_of = [&](int it, int& m0, int& n0) -> bool {
        const int base = it * G;
        const int left = ntiles - base;
        if (left <= 0) return false;
        const int g = left < G ? left : G;
        if ((int)blockIdx.x >= g) return false;
        const int id = base + xcd_tile_id(g, blockIdx.x);
        m0 = (id / a.tiles_n) * BM;
        n0 = (id % a.tiles_n) * BN;
        return true;
    };

    // ---- loader state: the tile whose steps are being ISSUED ----
    unsigned a_base[A_ROUNDS];
    int a_th[A_ROUNDS];
    bool b_ok[B_LOADS];
    const bf16_t* b_ptr[B_LOADS];
    bool l_live;
    auto loader_setup = [&](int it) {
        int m0, n0;
        l_live = tile_of(it, m0, n0);
        if (!l_live) return;
        const int row0_id = m0 / W;
#pragma unroll
        for (int i = 0; i < A_ROUNDS; ++i) {
            const int r = i * RPR + (tid >> 3);
            const int lc = (tid & 7) ^ ((r >> 1) & 7);
            const int hl = r / WP, w = r - hl * WP - 1;
            const int rowid = row0_id + hl;
            const long long m = (long long)rowid * W + w;
            const bool valid = r < p.img_rows && w >= 0 && w < W && m < a.M;
            a_base[i] = valid ? (unsigned)m * (unsigned)a.Cs + lc * 8 : 0u;
            a_th[i] = valid ? ((((rowid / H) % T) << 16) | (rowid % H)) : (int)0x80000000;
        }
#pragma unroll
        for (int j = 0; j < B_LOADS; ++j) {
            const int row = j * RPR + (tid >> 3);
            const int lc = (tid & 7) ^ ((row >> 1) & 7);
            const int n = n0 + row;
            b_ok[j] = n < a.Ncols;
            const int wr = b_ok[j] ? (a.perm_f > 1 ? (n % a.perm_c) * a.perm_f + n / a.perm_c : n) : 0;
            b_ptr[j] = a.wgt + (size_t)wr * a.w_row_stride + lc * 8;
        }
    };
    auto stage_a = [&](const GenieTriStep& e, bool live, int i, char* abuf) {
        const int t = (a_th[i] >> 16) + e.dt, h = (a_th[i] & 0xffff) + e.dh;
        const bool ok = live & ((unsigned)t < (unsigned)T) & ((unsigned)h < (unsigned)H);
        const bf16_t* q = a.src + (int)(a_base[i] + (unsigned)e.a_delta);
        q = ok ? q : zero;
        __builtin_amdgcn_global_load_lds(GLB_PTR(q), LDS_PTR(abuf + (i * NWAVE + wave) * 1024), 16, 0, 0);
    };
    auto stage_b = [&](int wofs, bool live, char* bbuf) {
#pragma unroll
        for (int j = 0; j < B_LOADS; ++j) {
            const bf16_t* q = (live & b_ok[j]) ? b_ptr[j] + wofs : zero;
            __builtin_amdgcn_global_load_lds(GLB_PTR(q), LDS_PTR(bbuf + (j * NWAVE + wave) * 1024), 16, 0, 0);
        }
    };

    unsigned a_off[3][4][TM], b_off[4][TN];
    const int khalf = lane >> 5;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int pl = wm * 64 + i * 32 + (lane & 31);
        const int hl = pl / W;
        const int row0 = hl * WP + (pl - hl * W);       // image row of (pixel, dw = -1); + s for dw = s - 1
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int row = row0 + s;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) a_off[s][ks][i] = (unsigned)(row * 128 + (((ks * 2 + khalf) ^ ((row >> 1) & 7)) << 4));
        }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int row = wn * 64 + j * 32 + (lane & 31);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) b_off[ks][j] = (unsigned)(row * 128 + (((ks * 2 + khalf) ^ ((row >> 1) & 7)) << 4));
    }

    f32x16_t acc[TM][TN];
    auto raw_barrier = [&]() {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    char* const A0 = smem;
    char* const B0 = smem + 2 * A_BYTES;                // four weight slots
    char* const epi = B0 + 4 * B_BYTES;                 // 1 KiB of row offsets for the epilogue

    // ---- prologue (once per block): image 0 and weight tiles 0, 1, 2 of the first tile ----
    loader_setup(0);
    {
        const GenieTriStep e = steps[0];
#pragma unroll
        for (int i = 0; i < A_ROUNDS; ++i) stage_a(e, l_live, i, A0);
        stage_b(e.wofs0, l_live, B0);
        stage_b(e.wofs1, l_live, B0 + B_BYTES);
        stage_b(e.wofs2, l_live, B0 + 2 * B_BYTES);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    raw_barrier();

    bf16x8_t af0[TM], bf0[TN];
    auto preread = [&](const char* abuf, int s, const char* bbuf) {
#pragma unroll
        for (int i = 0; i < TM; ++i) af0[i] = *reinterpret_cast<const bf16x8_t*>(abuf + a_off[s][0][i]);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf0[j] = *reinterpret_cast<const bf16x8_t*>(bbuf + b_off[0][j]);
    };
    auto mfma4 = [&](const bf16x8_t (&fa)[TM], const bf16x8_t (&fb)[TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    };
    auto read_ks = [&](const char* abuf, int s, const char* bbuf, int ks, bf16x8_t (&fa)[TM], bf16x8_t (&fb)[TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const bf16x8_t*>(abuf + a_off[s][ks][i]);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const bf16x8_t*>(bbuf + b_off[ks][j]);
    };
    preread(A0, 0, B0);

    int k = 0;                                          // K-tile counter of the whole stream: weight tile k sits in slot k & 3
    int ip = 0;                                         // step counter parity: the image of the current step sits in A0 + ip * A_BYTES
    for (int it = 0;; ++it) {
        int m0, n0;
        if (!tile_of(it, m0, n0)) break;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int i = 0; i < nsteps; ++i, k += 3, ip ^= 1) {
            const bool last = i + 1 == nsteps;
            if (last) loader_setup(it + 1);             // every load of this tile is out: the loader moves on to the next tile
            const bool has_next = last ? l_live : true;
            const GenieTriStep nxt = steps[last ? 0 : i + 1];
            char* const acur = A0 + ip * A_BYTES;
            char* const anxt = A0 + (ip ^ 1) * A_BYTES;
#define GENIE_KTILE(S, ISSUE, NEXT_A, NEXT_S, WAITN)                                                              \
            {                                                                                                    \
                const char* bcur = B0 + ((k + S) & 3) * B_BYTES;                                                 \
                const char* bnext = B0 + ((k + S + 1) & 3) * B_BYTES;                                            \
                bf16x8_t fa1[TM], fb1[TN], fa2[TM], fb2[TN];                                                     \
                read_ks(acur, S, bcur, 1, fa1, fb1);                                                             \
                mfma4(af0, bf0);                                                                                 \
                __builtin_amdgcn_sched_barrier(0);                                                               \
                ISSUE                                                                                            \
                __builtin_amdgcn_sched_barrier(0);                                                               \
                read_ks(acur, S, bcur, 2, fa2, fb2);                                                             \
                mfma4(fa1, fb1);                                                                                 \
                __builtin_amdgcn_sched_barrier(0);                                                               \
                read_ks(acur, S, bcur, 3, fa1, fb1);                                                             \
                mfma4(fa2, fb2);                                                                                 \
                __builtin_amdgcn_sched_barrier(0);                                                               \
                preread(NEXT_A, NEXT_S, bnext);                                                                  \
                mfma4(fa1, fb1);                                                                                 \
                __builtin_amdgcn_sched_barrier(0);                                                               \
                asm volatile("s_waitcnt vmcnt(" #WAITN ")" ::: "memory");                                        \
                raw_barrier();                                                                                   \
            }
            GENIE_KTILE(0,
                        stage_a(nxt, has_next, 0, anxt); stage_a(nxt, has_next, 1, anxt); stage_a(nxt, has_next, 2, anxt);
                        stage_a(nxt, has_next, 3, anxt); stage_a(nxt, has_next, 4, anxt);
                        __builtin_amdgcn_sched_barrier(0);
                        stage_b(nxt.wofs0, has_next, B0 + ((k + 3) & 3) * B_BYTES);,
                        acur, 1, 7)
            GENIE_KTILE(1, stage_b(nxt.wofs1, has_next, B0 + ((k + 4) & 3) * B_BYTES);, acur, 2, 2)
            GENIE_KTILE(2, stage_b(nxt.wofs2, has_next, B0 + ((k + 5) & 3) * B_BYTES);, anxt, 0, 2)
#undef GENIE_KTILE
        }
        igemm_epilogue<BM, TM, TN>(a, acc, epi, m0, n0, wm, wn, tid, lane);
        __syncthreads();                                // the next tile's epilogue rewrites the row offsets
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

static int launch_igemm3p(const Igemm3Args& p, const GenieTriStep* steps, hipStream_t s) {
    constexpr int lds = 2 * (5 * 64 * 128) + 4 * 128 * 128 + 1024;
    static bool configured = false;
    static int ncu = 0;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute((const void*)igemm3p_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        int dev = 0;
        hipDeviceProp_t prop;
        if (e == hipSuccess) e = hipGetDevice(&dev);
        if (e == hipSuccess) e = hipGetDeviceProperties(&prop, dev);
        if (e != hipSuccess) {
            genie_set_error("igemm3p setup failed: %s", hipGetErrorString(e));
            return GENIE_ERR_HIP;
        }
        ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        configured = true;
    }
    const int ntiles = p.g.tiles_m * p.g.tiles_n;
    hipLaunchKernelGGL(igemm3p_kernel, dim3(ntiles < ncu ? ntiles : ncu), dim3(512), lds, s, p, steps);
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}


static int launch_igemm3d_lean(const Igemm3Args& p, const GenieTriStep* steps, hipStream_t s) {
    constexpr int lds = 2 * (5 * 64 * 128) + 4 * 128 * 128;
    static bool lconf = false;
    if (!lconf) {
        hipError_t e = hipFuncSetAttribute((const void*)igemm3dl_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            genie_set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return GENIE_ERR_HIP;
        }
        lconf = true;
    }
    hipLaunchKernelGGL(igemm3dl_kernel, dim3(p.g.tiles_m * p.g.tiles_n, 1), dim3(512), lds, s, p, steps);
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}

template <bool PRE, bool SPLITK, int NWAVE = 8>
static int launch_igemm3d_t(const Igemm3Args& p, const GenieTriStep* steps, hipStream_t s) {
    constexpr int lds = 2 * (5 * 64 * 128) + 4 * 128 * 128;
    if constexpr (PRE && !SPLITK && NWAVE == 8) {
        if (ig3_lean_ok(p) && !p.xcdcol) return launch_igemm3d_lean(p, steps, s);      // buffer-addressed LDS-DMA form of the default schedule
    }
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute((const void*)igemm3d_kernel<PRE, SPLITK, NWAVE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            genie_set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return GENIE_ERR_HIP;
        }
        configured = true;
    }
    const int nblk = p.xcdcol ? cdiv(p.g.tiles_m, 8 / p.g.tiles_n) * 8 : p.g.tiles_m * p.g.tiles_n;
    hipLaunchKernelGGL((igemm3d_kernel<PRE, SPLITK, NWAVE>), dim3(nblk, SPLITK ? p.g.split_k : 1), dim3(64 * NWAVE), lds, s, p, steps);
    GENIE_CHECK_LAUNCH();
    if (SPLITK) return genie_igemm_splitk_finish(p.g, s);
    return GENIE_OK;
}

template <bool PRE>
static int launch_igemm3d(const Igemm3Args& p, const GenieTriStep* steps, hipStream_t s) {
    return p.g.split_k > 1 ? launch_igemm3d_t<PRE, true>(p, steps, s) : launch_igemm3d_t<PRE, false>(p, steps, s);
}

template <int BM, bool PIPE>
static int launch_igemm3(const Igemm3Args& p, const GenieTriStep* steps, hipStream_t s) {
    constexpr int NT = BM * 2;
    constexpr int lds = 2 * (5 * (NT / 8) * 128) + 2 * 128 * 128;
    auto k = igemm3_kernel<BM, PIPE>;
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            genie_set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return GENIE_ERR_HIP;
        }
        configured = true;
    }
    hipLaunchKernelGGL(k, dim3(p.g.tiles_m * p.g.tiles_n, p.g.split_k > 1 ? p.g.split_k : 1), dim3(NT), lds, s, p, steps);
    GENIE_CHECK_LAUNCH();
    if (p.g.split_k > 1) return genie_igemm_splitk_finish(p.g, s);
    return GENIE_OK;
}

// Called by genie_conv_igemm (conv_igemm.hip) with the generic arguments already filled in.  Returns 1 when the problem is not
// eligible (caller falls back to the generic kernel), 0 on launch, < 0 on error.
int genie_conv_igemm3_try(const GenieConvDesc* d, IgemmArgs a, hipStream_t s) {
    if (!d->tri_steps || d->n_tri_steps <= 0 || d->tri_bm < 0) return 1;
    if (d->small_c || d->st != 1 || d->sh != 1 || d->sw != 1) return 1;
    if (d->To != d->Ts || d->Ho != d->Hs || d->Wo != d->Ws) return 1;
    const int W = d->Wo;
    if (W < 8 || a.Nstore <= 32) return 1;
    if (d->Ts >= 32768 || d->Hs >= 65536) return 1;
    int bm = d->tri_bm;
    const int tiles_n = cdiv(a.Nstore, 128);
    int split = 1, per_split = d->n_tri_steps;
    if (bm == 0) {
        const long long t128 = (long long)cdiv(a.M, 128) * tiles_n, t256 = (long long)cdiv(a.M, 256) * tiles_n;
        static const int mode = getenv("GENIE_TRI_SPLITK") ? atoi(getenv("GENIE_TRI_SPLITK")) : 1;
        if (t256 <= 128 && mode && d->splitk_ws && 256 % W == 0 && a.M >= 1024) {
            // Few tiles (low-resolution layers): split the step table over blockIdx.y so that the 256-row deep-prefetch kernel
            // still gets one block per CU; fp32 partials go to the split-K scratch and igemm_splitk_finish_kernel finishes
            // (512 -> 512 @ 4x8x8, B = 8: 486 -> 524 TFLOP/s against the generic kernel's split-K on a box that ran 5 % slow).
            int sk = (int)(256 / t256);
            per_split = cdiv(d->n_tri_steps, sk);
            if (per_split < 6) per_split = 6;                                    // >= 18 K tiles per block
            sk = cdiv(d->n_tri_steps, per_split);
            if (sk >= 2 && (long long)sk * a.M * a.Nstore * 4 <= d->splitk_ws_bytes) {
                split = sk;
                bm = 256;
            }
        }
        if (bm == 0) {
            if (t128 < 256) return 1;                    // few tiles: the generic kernel's split-K fills the chip better
            // 256-row tiles from one block per CU on (GENIE_TRI_BM256_MIN, was 512 = two rounds): 512 -> 512 @4x8x8 at 64 clips 984 -> 1237 TFLOP/s
            static const int bm256_min = getenv("GENIE_TRI_BM256_MIN") ? atoi(getenv("GENIE_TRI_BM256_MIN")) : 256;
            bm = (256 % W == 0 && t256 >= bm256_min) ? 256 : 128;
        }
    }
    if (bm != 128 && bm != 256) {
        genie_set_error("genie_conv_igemm: tri_bm must be 0, 128 or 256 (got %d)", bm);
        return GENIE_ERR_ARG;
    }
    if (bm % W != 0) {
        if (d->tri_bm == 0 && 128 % W == 0) bm = 128;
        else return 1;
    }
    Igemm3Args p;
    p.g = a;
    p.g.tiles_m = cdiv(a.M, bm);
    p.g.tiles_n = tiles_n;
    p.g.split_k = split;
    p.g.chunks_per_split = per_split;                   // steps per split
    p.g.ws = split > 1 ? (float*)d->splitk_ws : nullptr;
    p.g.ws_ld = a.Nstore;
    // GroupNorm fusion lives in the shared epilogue of the 256-row, non-split, non-persistent kernels
    const bool gn_ok = bm == 256 && split == 1 && (d->tri_flags & 128) == 0;
    if (!gn_ok) { p.g.gn_sums = nullptr; p.g.gnb_x = nullptr; }
    const int gn_mask = (p.g.gn_sums ? 1 : 0) | (p.g.gnb_x ? 2 : 0);
    p.nsteps = d->n_tri_steps;
    p.WP = W + 2;
    p.img_rows = (bm / W) * (W + 2);
    p.dbg = d->tri_flags & 60;                          // bits 2-5: timing ablations
    // frame-fastest tile order (igemm3d / igemm3w): whole row tiles per frame, more than one per frame, no K split, every tile complete
    static const int tfast_env = getenv("GENIE_TRI_TFAST") ? atoi(getenv("GENIE_TRI_TFAST")) : 0;     // measured: 272.8 -> 275.0 ms per step, off
    p.tf_T = 0; p.tf_F = 0;
    if (tfast_env && bm == 256 && split == 1 && (d->Ho * d->Wo) % 256 == 0 && d->Ho * d->Wo > 256 && d->To > 1) {
        p.tf_T = d->To;
        p.tf_F = d->Ho * d->Wo / 256;
    }
    static const int xcdcol_env = getenv("GENIE_TRI_XCDCOL") ? atoi(getenv("GENIE_TRI_XCDCOL")) : 0;
    p.xcdcol = (((d->tri_flags & 512) || xcdcol_env) && bm == 256 && (tiles_n == 2 || tiles_n == 4 || tiles_n == 8)) ? 1 : 0;
    const bool pipe = (d->tri_flags & 1) == 0;
    // bit 10 / GENIE_TRI_WIDE=1: 256 x 256 tiles where the layer has >= 256 output channels and still >= one block per CU
    static const int wide_env = getenv("GENIE_TRI_WIDE") ? atoi(getenv("GENIE_TRI_WIDE")) : GENIE_TRI_WIDE_DEFAULT;
    if (((d->tri_flags & 1024) || wide_env) && bm == 256 && split == 1 && a.Nstore >= 256 && p.dbg == 0 && (d->tri_flags & (1 | 2 | 64 | 128 | 256)) == 0 &&
        ((long long)p.g.tiles_m * cdiv(a.Nstore, 256) >= 256 || (d->tri_flags & 2048))) {      // bit 11: even with few tiles (tests)
        p.g.tiles_n = cdiv(a.Nstore, 256);
        genie_note_variant(GENIE_VARIANT_IGEMM3_WIDE);
        genie_note_gn_fused(gn_mask);
        // bit 12 / GENIE_TRI_X=1: the same tile as FOUR waves of 128 x 128 (one per SIMD, conv_igemm3x.hip) -- A/B switch
        static const int x_env = getenv("GENIE_TRI_X") ? atoi(getenv("GENIE_TRI_X")) : 0;
        if (((d->tri_flags & 4096) || x_env) && ig3_lean_ok(p)) return genie_launch_igemm3x(p, d->tri_steps, s);
        return launch_igemm3w<false>(p, d->tri_steps, s);
    }
    // two-blocks-per-CU form (igemm3h_kernel) where the layer has <= 128 output columns: GENIE_TRI_H=0 switches it off
    static const int h_env = getenv("GENIE_TRI_H") ? atoi(getenv("GENIE_TRI_H")) : 1;
    if (h_env && bm == 256 && split == 1 && a.Nstore <= 128 && W >= 16 && 256 % W == 0 && p.dbg == 0 && (d->tri_flags & (1 | 2 | 64 | 128 | 256 | 512)) == 0 &&
        !p.g.gn_sums && !p.g.gnb_x && ig3_lean_ok(p) && ((long long)p.g.tiles_m >= 512 || (d->tri_flags & 2048))) {      // bit 11: even with few tiles (tests)
        genie_note_variant(GENIE_VARIANT_IGEMM3_H);
        genie_note_gn_fused(0);
        return launch_igemm3h(p, d->tri_steps, s);
    }
    genie_note_variant(split > 1 ? GENIE_VARIANT_IGEMM3_256_SPLITK : (bm == 256 ? GENIE_VARIANT_IGEMM3_256 : GENIE_VARIANT_IGEMM3_128));
    genie_note_gn_fused(gn_mask);
    if (bm == 256 && (d->tri_flags & 2) == 0) {                                                // deep-prefetch schedules
        // bit 7: the persistent form.  Measured A/B on one box (B = 8 step): the kernel itself +1 % (1117 -> 1128 TFLOP/s averaged over
        // its launches), the step unchanged (87.8 ms both ways) -- the pipeline fill / drain per tile is not what holds the
        // one-tile kernel at 56 % MFMA-busy.  Kept selectable, off by default.
        if ((d->tri_flags & 128) && split == 1 && (d->tri_flags & 64) == 0 && p.dbg == 0) return launch_igemm3p(p, d->tri_steps, s);
        // bit 8 (or GENIE_TRI_W4=1): one wave per SIMD (4 waves of 128 x 64).  Measured A/B on one box: 1122 -> 1071 TFLOP/s averaged over
        // the launches of a step (-4.5 %; split-K layers 616 -> 584): a single wave per SIMD does not cover its own LDS / barrier
        // latencies, two co-resident waves do.  Kept selectable, off by default.
        static const int w4 = getenv("GENIE_TRI_W4") ? atoi(getenv("GENIE_TRI_W4")) : 0;
        if (((d->tri_flags & 256) || w4) && (d->tri_flags & 64) == 0 && p.dbg == 0)
            return split > 1 ? launch_igemm3d_t<true, true, 4>(p, d->tri_steps, s) : launch_igemm3d_t<true, false, 4>(p, d->tri_steps, s);
        return (d->tri_flags & 64) ? launch_igemm3d<false>(p, d->tri_steps, s) : launch_igemm3d<true>(p, d->tri_steps, s);
    }
    if (bm == 256) return pipe ? launch_igemm3<256, true>(p, d->tri_steps, s) : launch_igemm3<256, false>(p, d->tri_steps, s);
    return pipe ? launch_igemm3<128, true>(p, d->tri_steps, s) : launch_igemm3<128, false>(p, d->tri_steps, s);
}
