// Shared pieces of the gather-GEMM kernels (conv_igemm.hip: generic 128-row tiles; conv_igemm3.hip: kw-triple tiles).
#pragma once
#include "common.h"
#include "genie_hip.h"

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
    int split_k;      // > 1: blockIdx.y = K split; partial tiles go to ws (fp32 [split][M][ws_ld]) and a second kernel finishes
    int chunks_per_split;
    float* ws;
    int ws_ld;
    // optional GroupNorm fusion into the epilogue (one group, plain destination, whole row tiles inside a sample):
    double* gn_sums;          // [N][2] += (sum, sum of squares) of the bf16-rounded outputs of sample n -- the statistics of the NEXT GroupNorm
    const bf16_t* gnb_x;      // the destination is the gradient of a GroupNorm(+act) OUTPUT and this is that GroupNorm's INPUT (same layout):
    const float* gnb_gamma;   //   gnb_part[n][row tile][c] = (sum dz, sum dz * xhat) over the tile's rows, dz = dst * act'(xhat * gamma + beta)
    const float* gnb_beta;    //   = what gn_bwd_reduce_kernel (norm.hip) would write with one block per row tile
    const float* gnb_mean;
    const float* gnb_rstd;
    float* gnb_part;
    int gnb_act, gnb_nblk;
    int gn_rows;              // rows (pixels) per sample
};

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// XCD-aware, bijective block -> tile id: consecutive ids land on ONE XCD (blocks are dealt round-robin to the 8 XCDs)
__device__ __forceinline__ int xcd_tile_id(int nb, int b) {
    const int q = nb >> 3, r = nb & 7, xcd = b & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}

// Epilogue shared by the gather-GEMM kernels.  acc[i][j] is the wave's (wm, wn) sub-tile as TM x TN 32x32 MFMA tiles
// (row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5), col = lane & 31); BMROWS = rows of the block tile (<= blockDim.x);
// smem: >= BMROWS ints of LDS that nobody reads any more (the caller has passed its last barrier).
template <int BMROWS, int TM, int TN, bool GN = true>
__device__ __forceinline__ void igemm_epilogue(const IgemmArgs& a, f32x16_t (&acc)[TM][TN], char* smem, int m0, int n0, int wm, int wn,
                                               int tid, int lane) {
    const int khalf = lane >> 5;
    // ---- epilogue: dest offset = rowoff[m] + coloff[n]; rowoff staged through LDS ----
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
    const bool vec4 = (a.shuf_c & 3) == 0 && (a.Nstore & 3) == 0 && (a.Cd & 3) == 0;
    if (vec4) {
        // 4x4 transposes inside lane quads (DPP quad_perm): lane j of a quad ends up with ONE row and FOUR consecutive
        // columns -> 8-byte stores instead of 2-byte ones (4x fewer store instructions)
        const int jq = lane & 3;
        // GroupNorm fusion (see IgemmArgs): the tile lies inside sample gn_n, row tile gn_blk of it
        const bool gn_fwd = GN && BMROWS == 256 && a.gn_sums != nullptr, gn_bwd = GN && BMROWS == 256 && a.gnb_x != nullptr;   // block-uniform; 256-row tiles only
        int gn_n = 0, gn_blk = 0;
        float gmu = 0.f, grs = 0.f, fs = 0.f, fss = 0.f;
        if (gn_fwd || gn_bwd) {
            gn_n = m0 / a.gn_rows;
            gn_blk = (m0 - gn_n * a.gn_rows) / BMROWS;
        }
        if (gn_bwd) { gmu = a.gnb_mean[gn_n]; grs = a.gnb_rstd[gn_n]; }
        float* const gn_red = reinterpret_cast<float*>(smem + 1024);                         // [row wave][256 columns][2], behind the row offsets
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nq = n0 + wn * (TN * 32) + j * 32 + (lane & 28);         // first of this quad's 4 columns
            const bool colok = nq < a.Nstore;
            const int sub = nq / a.shuf_c, ch = nq - sub * a.shuf_c;
            const int r_ = sub % a.shuf_r, q_ = (sub / a.shuf_r) % a.shuf_q, p_ = sub / (a.shuf_r * a.shuf_q);
            const int coloff = ((p_ * a.Hd + q_) * a.Wd + r_) * a.Cd + ch;
            float bias4[4] = {0.f, 0.f, 0.f, 0.f};
            if (a.bias && colok) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int n = nq + e;
                    if (n < a.Ncols) bias4[e] = a.bias[a.perm_f > 1 ? (n % a.perm_c) * a.perm_f + n / a.perm_c : n];
                }
            }
            // residual rows of this column tile: all loads are issued before the first use (one load + dependent add per group
            // serialised TM * 4 global round trips per column tile -- the 1x1 shortcut dgrad, which is pure traffic, ran at half
            // the speed of the same GEMM without a residual)
            float ga4[4] = {0.f, 0.f, 0.f, 0.f}, gb4[4] = {0.f, 0.f, 0.f, 0.f}, cs1[4] = {0.f, 0.f, 0.f, 0.f}, cs2[4] = {0.f, 0.f, 0.f, 0.f};
            u32x2_t gxr[TM][4];
            if (gn_bwd) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int n = nq + e;
                    if (colok && n < a.Ncols) {
                        const float ga = a.gnb_gamma ? a.gnb_gamma[n] : 1.f, be = a.gnb_beta ? a.gnb_beta[n] : 0.f;
                        ga4[e] = grs * ga;
                        gb4[e] = be - gmu * grs * ga;
                    }
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int ro = rowoff[wm * (TM * 32) + i * 32 + 8 * g + 4 * khalf + jq];
                        gxr[i][g] = u32x2_t{0u, 0u};
                        if (ro >= 0 && colok) gxr[i][g] = *reinterpret_cast<const u32x2_t*>(a.gnb_x + (unsigned)ro + coloff);
                    }
            }
            u32x2_t rres[TM][4];
            if (a.resid) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int ro = rowoff[wm * (TM * 32) + i * 32 + 8 * g + 4 * khalf + jq];
                        rres[i][g] = u32x2_t{0u, 0u};
                        if (ro >= 0 && colok) rres[i][g] = *reinterpret_cast<const u32x2_t*>(a.resid + (unsigned)ro + coloff);
                    }
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e];
                    // round 1: partner lane ^ 1 swaps the off-diagonal of each 2x2
#pragma unroll
                    for (int k = 0; k < 4; k += 2) {
                        const float send = (jq & 1) ? v[k] : v[k + 1];
                        const float recv = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), 0xB1, 0xF, 0xF, true));
                        if (jq & 1) v[k] = recv; else v[k + 1] = recv;
                    }
                    // round 2: partner lane ^ 2
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const float send = (jq & 2) ? v[k] : v[k + 2];
                        const float recv = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), 0x4E, 0xF, 0xF, true));
                        if (jq & 2) v[k] = recv; else v[k + 2] = recv;
                    }
                    // now v[e] = value of row (8 g + 4 khalf + jq), column nq + e
                    const int row = wm * (TM * 32) + i * 32 + 8 * g + 4 * khalf + jq;
                    const int ro = rowoff[row];
                    if (ro < 0 || !colok) continue;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += bias4[e];
                    if (a.resid) {
                        const u32x2_t rv = rres[i][g];
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
                    *reinterpret_cast<u32x2_t*>(a.dst + (unsigned)ro + coloff) = ov;
                    if (gn_fwd || gn_bwd) {
                        const float o0 = __uint_as_float(ov[0] << 16), o1 = __uint_as_float(ov[0] & 0xffff0000u);     // what was stored (bf16)
                        const float o2 = __uint_as_float(ov[1] << 16), o3 = __uint_as_float(ov[1] & 0xffff0000u);
                        if (gn_fwd) {
                            fs += (o0 + o1) + (o2 + o3);
                            fss += (o0 * o0 + o1 * o1) + (o2 * o2 + o3 * o3);
                        }
                        if (gn_bwd) {
                            const u32x2_t xr = gxr[i][g];
                            const float xv[4] = {__uint_as_float(xr[0] << 16), __uint_as_float(xr[0] & 0xffff0000u),
                                                 __uint_as_float(xr[1] << 16), __uint_as_float(xr[1] & 0xffff0000u)};
                            const float dv[4] = {o0, o1, o2, o3};
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float z = xv[e] * ga4[e] + gb4[e];
                                const float dz = a.gnb_act == 1 ? dv[e] * silu_grad_f(z) : (a.gnb_act == 2 ? (z > 0.f ? dv[e] : 0.01f * dv[e]) : dv[e]);
                                cs1[e] += dz;
                                cs2[e] += dz * (xv[e] - gmu) * grs;
                            }
                        }
                    }
                }
            }
            if (gn_bwd) {
                // the 8 lanes that share these 4 columns: lane quad (bits 0-1) and the two 32-lane halves (bit 5)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v1 = cs1[e], v2 = cs2[e];
                    v1 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v1), 0xB1, 0xF, 0xF, true));
                    v2 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v2), 0xB1, 0xF, 0xF, true));
                    v1 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v1), 0x4E, 0xF, 0xF, true));
                    v2 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v2), 0x4E, 0xF, 0xF, true));
                    v1 += __shfl_xor(v1, 32, 64);
                    v2 += __shfl_xor(v2, 32, 64);
                    if ((lane & 35) == 0) {
                        float* o = gn_red + ((wm * 256) + wn * (TN * 32) + j * 32 + (lane & 28) + e) * 2;
                        o[0] = v1; o[1] = v2;
                    }
                }
            }
        }
        if (gn_fwd) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { fs += __shfl_xor(fs, o, 64); fss += __shfl_xor(fss, o, 64); }
            if (lane == 0) {
                atomicAdd(a.gn_sums + 2 * gn_n, (double)fs);
                atomicAdd(a.gn_sums + 2 * gn_n + 1, (double)fss);
            }
        }
        if (gn_bwd) {
            __syncthreads();
            constexpr int WMV = BMROWS / (TM * 32);
            const int bn = ((int)(blockDim.x >> 6) / WMV) * (TN * 32);                        // columns of the block tile (<= 256)
            if (tid < bn && n0 + tid < a.Cd) {
                float t1 = 0.f, t2 = 0.f;
#pragma unroll
                for (int w = 0; w < WMV; ++w) { t1 += gn_red[(w * 256 + tid) * 2]; t2 += gn_red[(w * 256 + tid) * 2 + 1]; }
                float* o = a.gnb_part + (((long long)gn_n * a.gnb_nblk + gn_blk) * a.Cd + n0 + tid) * 2;
                o[0] = t1; o[1] = t2;
            }
        }
        return;
    }
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

// split-K: dump the wave's fp32 partial tile to ws[split][m][n]; igemm_splitk_finish_kernel (conv_igemm.hip) sums the splits and
// applies bias / resid / act / the destination mapping.
template <int TM, int TN>
__device__ __forceinline__ void igemm_store_partials(const IgemmArgs& a, f32x16_t (&acc)[TM][TN], int split, int m0, int n0, int wm, int wn,
                                                     int lane) {
    float* wsp = a.ws + (size_t)split * a.M * a.ws_ld;
    const int khalf = lane >> 5;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (TN * 32) + j * 32 + (lane & 31);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r16 = 0; r16 < 16; ++r16) {
                const int m = m0 + wm * (TM * 32) + i * 32 + (r16 & 3) + 8 * (r16 >> 2) + 4 * khalf;
                if (m < a.M && n < a.ws_ld) wsp[(size_t)m * a.ws_ld + n] = acc[i][j][r16];
            }
    }
}
