// Pieces shared by the kw-triple gather-GEMM kernels (conv_igemm3.hip and the one-wave-per-SIMD 128 x 128 wave-tile variant in
// conv_igemm3x.hip): the argument block, the frame-fastest tile order and the zero-frame trimming of the step table.
#pragma once
#include "igemm_common.h"

#define IG3_OOB 0x80000000u

struct Igemm3Args {
    IgemmArgs g;                   // tensors, row grid (To/Ho/Wo == Ts/Hs/Ws), destination mapping, weight permutation, tiles
    int nsteps;                    // entries of the step table, one per (dt, dh, channel block)
    int WP, img_rows;              // W + 2, (BM / W) * WP
    int xcdcol;                    // 1: every XCD works on ONE column tile (its weight slice stays L2-resident); grid is rounded up
    int dbg;                       // timing ablations (results are WRONG when non-zero): 4 no glds in the loop, 8 no ds_read,
                                   // 16 no MFMA, 32 no barrier
    int tf_T, tf_F;                // > 0: row tiles are visited FRAME-FASTEST -- tile index i of the launch order is tile
                                   // ((n * T + t) * F + hb) with (n, hb, t) = unravel(i, (N, F, T)), F = row tiles per frame
};

// Frame-fastest tile order: consecutive row tiles (which run at the same time on one XCD) are the SAME rows of consecutive frames, so
// the dt = -1 / -2 halo a tile reads is the tile its neighbour just streamed (L2 hit); in the linear order that neighbour is a whole
// frame of tiles away and the halo comes back from HBM (FETCH_SIZE of the 128-channel layers: 2.6 x the input).
__device__ __forceinline__ int tf_remap(int i, int T, int F) {
    if (T <= 0) return i;
    const int t = i % T, q = i / T;
    const int hb = q % F, n = q / F;
    return (n * T + t) * F + hb;
}

// Zero-frame skipping.  With the step table sorted by dt (GenieTriStep.rows_per_dt > 0) a row tile that lies inside ONE frame t needs only
// the rows whose source frame t + dt exists; the others would stage zeros and multiply them (time padding: 2 of 3 x 16 (frame, dt) pairs of a
// 16-frame 'same' conv, 3 of a causal one; 1 / 6 at 4 frames).  Trims [steps, steps + nsteps) to that range.
__device__ __forceinline__ void tri_trim_range(const GenieTriStep* __restrict__ steps, int nsteps, int m0, int bm, long long M, int H, int W, int T,
                                               int& first, int& count) {
    const int rpd = __builtin_amdgcn_readfirstlane(steps[0].rows_per_dt), dmin = __builtin_amdgcn_readfirstlane(steps[0].dt_min);
    const unsigned hw = (unsigned)(H * W);
    const unsigned last = (long long)m0 + bm - 1 < M ? (unsigned)(m0 + bm - 1) : (unsigned)(M - 1);
    const unsigned f0 = (unsigned)m0 / hw, f1 = last / hw;
    const int t = (int)(f0 % (unsigned)T), ndt = rpd > 0 ? nsteps / rpd : 0;
    int lo = -t - dmin, hi = T - t - dmin;             // dt index range [lo, hi) with 0 <= t + dt < T
    lo = lo < 0 ? 0 : lo;
    hi = hi > ndt ? ndt : hi;
    const bool trim = rpd > 0 && f0 == f1 && hi > lo;
    // block-uniform by construction; readfirstlane keeps the table pointer in SGPRs (its reads must stay scalar loads: the kernels count
    // their vector-memory operations)
    first = __builtin_amdgcn_readfirstlane(trim ? lo * rpd : 0);
    count = __builtin_amdgcn_readfirstlane(trim ? (hi - lo) * rpd : nsteps);
}


// conv_igemm3x.hip: 256 x 256 block tile, FOUR waves of 128 x 128 (one per SIMD, accumulators in the AGPR half of the register file)
int genie_launch_igemm3x(const Igemm3Args& p, const GenieTriStep* steps, hipStream_t s);
