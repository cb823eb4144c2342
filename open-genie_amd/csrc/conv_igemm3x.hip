// kw-triple gather-GEMM, 256 x 256 block tile as FOUR waves of 128 x 128 -- one wave per SIMD, 256 accumulator registers per lane in the
// AGPR half of the register file.  Same algorithm, LDS images, weight ring, swizzles and DMA schedule as
// igemm3w_kernel<false, true, 0> (conv_igemm3.hip: 8 waves of 64 x 128); what changes is the arithmetic intensity of the LDS side:
//
//                                   igemm3w (8 waves, 64 x 128 each)      here (4 waves, 128 x 128 each)
//   fragment reads per 16-deep step   8 x (2 A + 4 B) = 48 ds_read_b128      4 x (4 A + 4 B) = 32        (-33 %)
//   MFMAs per wave between barriers   16                                      32
//   waves at the barrier              8                                       4
//
// The MFMA kernels of this library run at the chip's POWER cap (identical launch on zero operands: +40 %), so the lever is energy per
// FLOP: a third fewer LDS fragment bytes per MFMA.  Whether that pays against a single wave per SIMD having nobody to cover its barrier /
// LDS latencies (the 128 x 64-per-wave experiment of round 3 lost 4.5 %) is an A/B question: GENIE_TRI_X=1 / tri_flags bit 12 selects this
// kernel for the layers the wide kernel takes.
//
// MEASURED (profiles/r05_igemm3x_ab.jsonl, same box, interleaved rounds, 64 clips): it LOSES.  256 -> 256 @16x32x32: 1150 vs 1391 TFLOP/s on
// random operands (0.46 vs 0.556 of peak), 1370 vs 1960 on zeros (0.55 vs 0.78); 512 -> 512 @8x16x16: 0.52 vs 0.58.  The zero-operand
// figure is the telling one: without the power cap the eight-wave kernel keeps the matrix pipe 78 % busy and this one 55 % -- with a single
// wave per SIMD every s_barrier, every s_waitcnt and the ~50 scalar / DMA instructions of a half-tile's ISSUE block are pipe-idle time that
// a second resident wave would have filled; the third fewer fragment reads do not buy that back.  (Prescribing the MFMA / ds_read / DMA
// interleave with sched_group_barrier made it 3 x slower still: 0.167 -- hipcc then waits on every read.)  Kept selectable and tested
// (tests/test_gpu_kernels.py::test_conv_triple_wide_kernel[four_waves]), OFF by default.  What would be needed is a hand-placed
// instruction stream per MFMA gap (cdna_hip_programming.md, one-wave-per-SIMD rules), i.e. an assembly kernel.
//
// Compiled WITHOUT -amdgpu-mfma-vgpr-form (Makefile): the 16 accumulator tiles must be allowed into AGPRs.
#include "igemm3_common.h"

namespace {

template <int DUMMY = 0>
__global__ void __attribute__((amdgpu_waves_per_eu(1, 1))) __launch_bounds__(256) igemm3x_kernel(const Igemm3Args p, const GenieTriStep* __restrict__ steps) {
    constexpr int BM = 256, BN = 256, NWAVE = 4, WN = 2, TM = 4, TN = 4;
    constexpr int RPR = 32, A_ROUNDS = 8;                                    // DMA rounds per image: 256 threads x 16 B = 32 image rows of 128 B
    constexpr int A_BYTES = 5 * 64 * 128, B_BYTES = BN * 64;                 // 40 KB images (<= 320 rows with the zero columns), 16 KB weight half-tiles
    constexpr int B_LOADS = 4;                                               // 16 rows x 64 B per wave instruction, 256 rows / 4 waves
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const IgemmArgs& a = p.g;
    int nsteps = p.nsteps;
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
    {
        int first;
        tri_trim_range(steps, nsteps, m0, BM, a.M, H, W, T, first, nsteps);
        steps += first;
    }

    // Image staging (see igemm3w_kernel): only the 256 real pixels of the tile are DMA'd (8 rounds of 32 rows); pixel q of image row
    // hl = q / W lands in LDS row hl * (W + 2) + q % W + 1; the zero columns left and right of every image row are written once per buffer.
    unsigned a_dst[A_ROUNDS];
    const int row0_id = m0 / W;
#pragma unroll
    for (int i = 0; i < A_ROUNDS; ++i) {
        const int q0 = (i * NWAVE + wave) * 8;                               // first pixel of the wave's 1-KiB piece (8 pixels of one image row, W >= 8)
        a_dst[i] = (unsigned)(((q0 / W) * WP + q0 % W + 1) * 128);
    }
    for (int e = tid; e < 2 * 2 * (BM / 8) * 8; e += 256) {                  // (buffer, side, image row, 16-B chunk); image rows beyond BM / W: skipped
        const int c = e & 7, hl = (e >> 3) % (BM / 8), side = (e >> 3) / (BM / 8) & 1, buf = e / (2 * (BM / 8) * 8);
        if (hl * W < BM) *reinterpret_cast<u32x4_t*>(smem + buf * A_BYTES + (hl * WP + (side ? W + 1 : 0)) * 128 + c * 16) = u32x4_t{0u, 0u, 0u, 0u};
    }
    __syncthreads();
    // buffer-addressed LDS-DMA (the LEAN form): the lane's offsets are loop-invariant, rows outside the tensor read zeros from the range check
    const int margin = (3 * H * W + 2 * W) * a.Cs;
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)(a.src + ((long long)m0 * a.Cs - margin)), (short)0, (int)IG3_OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)a.wgt, (short)0, (int)IG3_OOB, 0x00020000);
    uint32_t voff_a[A_ROUNDS], voff_b[B_LOADS];
    int at_s[A_ROUNDS], ah_s[A_ROUNDS];
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
    for (int j = 0; j < B_LOADS; ++j) {
        const int row = (j * NWAVE + wave) * 16 + (lane >> 2);
        const int lc = (lane & 3) ^ ((row >> 2) & 3);
        const int n = n0 + row;
        const int wr = n < a.Ncols ? (a.perm_f > 1 ? (n % a.perm_c) * a.perm_f + n / a.perm_c : n) : -1;
        voff_b[j] = wr >= 0 ? (uint32_t)(((size_t)wr * a.w_row_stride + lc * 8) * 2) : IG3_OOB;
    }
    auto stage_a = [&](const GenieTriStep& e, bool live, int i, char* abuf) {
        const bool ok = (int)live & (int)((unsigned)(at_s[i] + e.dt) < (unsigned)T) & (int)((unsigned)(ah_s[i] + e.dh) < (unsigned)H);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, LDS_PTR(abuf + a_dst[i]), 16, voff_a[i] | (ok ? 0u : IG3_OOB),
                                                 ok ? (uint32_t)((e.a_delta + margin) * 2) : 0u, 0, 0);
    };
    auto stage_b = [&](int wofs, bool live, char* bbuf) {                      // one 32-channel half of a weight tile
#pragma unroll
        for (int j = 0; j < B_LOADS; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, LDS_PTR(bbuf + (j * NWAVE + wave) * 1024), 16, voff_b[j] | (live ? 0u : IG3_OOB),
                                                     live ? (uint32_t)(wofs * 2) : 0u, 0, 0);
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
    auto mfma16 = [&](const bf16x8_t (&fa)[TM], const bf16x8_t (&fb)[TN]) {
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
        // half-tile R (0..5) of this step: shift S = R / 2, channel half HH = R % 2 (schedule: igemm3w_kernel).  ISSUE = the DMA group of the
        // half-tile: two image rounds of step i + 1 during R = 0..3 (eight rounds per image with 256 threads), then weight half-tile k + R + 3
        // (four pieces per wave); WAITN = its size: only that group stays in flight across the barrier.
#define GENIE_XTILE(R, S, HH, ISSUE, NEXT_A, NEXT_S, NEXT_KS, WAITN)                                              \
        {                                                                                                        \
            const char* bcur = B0 + ((k + R) & 3) * B_BYTES;                                                     \
            const char* bnext = B0 + ((k + R + 1) & 3) * B_BYTES;                                                \
            bf16x8_t fa1[TM], fb1[TN];                                                                           \
            read_frag(acur, S, 2 * HH + 1, bcur, 1, fa1, fb1);                                                   \
            mfma16(fa0, fb0);                                                                                    \
            __builtin_amdgcn_sched_barrier(0);                                                                   \
            ISSUE                                                                                                \
            __builtin_amdgcn_sched_barrier(0);                                                                   \
            read_frag(NEXT_A, NEXT_S, NEXT_KS, bnext, 0, fa0, fb0);                                              \
            mfma16(fa1, fb1);                                                                                    \
            __builtin_amdgcn_sched_barrier(0);                                                                   \
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(WAITN) : "memory");                                        \
            raw_barrier();                                                                                       \
        }
        GENIE_XTILE(0, 0, 0,
                    stage_a(nxt, has_next, 0, anxt); stage_a(nxt, has_next, 1, anxt);
                    __builtin_amdgcn_sched_barrier(0);
                    stage_b(cur.wofs1 + 32, true, B0 + ((k + 3) & 3) * B_BYTES);,
                    acur, 0, 2, 6)
        GENIE_XTILE(1, 0, 1,
                    stage_a(nxt, has_next, 2, anxt); stage_a(nxt, has_next, 3, anxt);
                    __builtin_amdgcn_sched_barrier(0);
                    stage_b(cur.wofs2, true, B0 + ((k + 4) & 3) * B_BYTES);,
                    acur, 1, 0, 6)
        GENIE_XTILE(2, 1, 0,
                    stage_a(nxt, has_next, 4, anxt); stage_a(nxt, has_next, 5, anxt);
                    __builtin_amdgcn_sched_barrier(0);
                    stage_b(cur.wofs2 + 32, true, B0 + ((k + 5) & 3) * B_BYTES);,
                    acur, 1, 2, 6)
        GENIE_XTILE(3, 1, 1,
                    stage_a(nxt, has_next, 6, anxt); stage_a(nxt, has_next, 7, anxt);
                    __builtin_amdgcn_sched_barrier(0);
                    stage_b(nxt.wofs0, has_next, B0 + ((k + 6) & 3) * B_BYTES);,
                    acur, 2, 0, 6)
        GENIE_XTILE(4, 2, 0,
                    stage_b(nxt.wofs0 + 32, has_next, B0 + ((k + 7) & 3) * B_BYTES);,
                    acur, 2, 2, 4)
        GENIE_XTILE(5, 2, 1,
                    stage_b(nxt.wofs1, has_next, B0 + ((k + 8) & 3) * B_BYTES);,
                    anxt, 0, 0, 4)
#undef GENIE_XTILE
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    raw_barrier();
    igemm_epilogue<BM, TM, TN>(a, acc, smem, m0, n0, wm, wn, tid, lane);
}

}  // namespace

int genie_launch_igemm3x(const Igemm3Args& p, const GenieTriStep* steps, hipStream_t s) {
    constexpr int lds = 2 * (5 * 64 * 128) + 4 * 256 * 64;
    static bool configured = false;
    if (!configured) {
        const hipError_t e = hipFuncSetAttribute((const void*)igemm3x_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            genie_set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return GENIE_ERR_HIP;
        }
        configured = true;
    }
    hipLaunchKernelGGL((igemm3x_kernel<0>), dim3(p.g.tiles_m * p.g.tiles_n, 1), dim3(256), lds, s, p, steps);
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}
