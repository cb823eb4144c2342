// Fused vocabulary head + masked token cross-entropy of DynamicsModel.compute_loss (reference genie/dynamics.py:62 `self.head`,
// :89-97 gather + F.cross_entropy; SURVEY.md section 8(b) `linear_cross_entropy(h, W, b, target, mask)`).
//
// The reference writes logits = h W^T + b (M rows x V = 2^18 columns, 1 GiB fp32 per sample), gathers the masked rows, soft-maxes them and
// back-propagates through the same matrix.  Here the M x V logits NEVER reach HBM: both directions are flash-attention-shaped sweeps with
// a 512-wide "head", written for one wave per SIMD and the whole 512-entry register file:
//
//   forward   (MODE 0)  per 128 rows of h (stationary, MFMA B operands in registers), sweep W in 32-row tiles through a 4-deep LDS ring:
//                       S^T = W_tile h^T (+ bias), online log-sum-exp per row, and -- when a gradient is wanted -- O^T += W_tile^T P^T,
//                       i.e. softmax(h W^T) W, which IS d loss / d h up to the one-hot term.  The vocabulary is split over blocks
//                       (flash-decoding style partial (max, sum, O) per split) so that 192 row tiles still fill 256 CUs.
//   combine             merges the splits: lse, the target logit (an fp32 dot product), the loss, dh = softmax W - W[target].
//   backward  (MODE 1)  the same kernel body with the roles swapped: per 128 rows of W (stationary), sweep h in 32-row tiles:
//                       P^T = exp(W h^T + b - lse) from the saved lse (no running maximum), dW^T += h_tile^T P, db += row sums.
//   scatter             the one-hot term of dW / db (rows W[target[m]]) and the scale / cast of dh.
//
// FLOPs: 4 M V D forward-with-gradient + 4 M V D backward = 4/3 of the materialising path's 6 M V D, in exchange for ~1 GB per sample
// of logits + gradient traffic and memory, and fp32 (not bf16-rounded) logits inside the softmax.
//
// Layouts follow attention_lean.hip (same swizzle key, LDS-DMA staging through a buffer descriptor whose range check zero-fills rows
// past the end, "swapped" first product so that a lane owns ONE stationary row and the softmax is lane-local, transposing LDS reads for
// the second product).  What is different is the budget: O^T is 32 x 512 fp32 per wave = 256 accumulator registers, the stationary
// fragments another 128, so the kernel runs one wave per SIMD (amdgpu_waves_per_eu(1, 1)), is compiled WITHOUT -amdgpu-mfma-vgpr-form
// (the accumulators must live in the AGPR half), and both products stream their LDS operand: 1 KB per MFMA = half the LDS port.
#include "common.h"
#include "genie_hip.h"
#include "attn_common.h"
#include <stdlib.h>

namespace {

constexpr float LCE_LOG2E = 1.4426950408889634f;

struct LceArgs {
    const bf16_t* A; long long a_pitch; int NA;      // stationary rows (MODE 0: h, MODE 1: W), one per lane column
    const bf16_t* T; long long t_pitch; int NT;      // streamed rows (MODE 0: W, MODE 1: h), 32 per tile
    const float* tvec;                               // per streamed row: MODE 0 bias[v] (may be null), MODE 1 e[m] = -lse[m] or -inf
    int tvec_len;                                    // entries behind tvec (MODE 1: M rounded up to 64, the pad rows hold -inf)
    const float* avec;                               // MODE 1: bias[v] per stationary row (may be null)
    int n_atiles, nsplit, tps, ntiles;               // tiles per split, tiles in all
    int Apad;                                        // n_atiles * 128
    float* part_ml; float* part_o;                   // MODE 0: [nsplit][Apad][2], [nsplit][Apad][DH]
    float* dW; float* db; const float* scale;        // MODE 1: dW [NA][DH] +=, db [NA] += (may be null), device scalar
};

template <int CPR>
__device__ __forceinline__ int lce_key(int row) {
    if (CPR >= 16) return ((row & 3) << 2) | ((row >> 2) & 3);
    return (((row >> 1) & 1) << 2) | ((row >> 2) & 3);
}

__device__ __forceinline__ void lce_swap32(float& a, float& b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float lce_xmax32(float x) { float a = x, b = x; lce_swap32(a, b); return fmaxf(a, b); }
__device__ __forceinline__ float lce_xsum32(float x) { float a = x, b = x; lce_swap32(a, b); return a + b; }

__device__ __forceinline__ void lce_dma16(const void* base, int bytes_left, char* lds, uint32_t voff) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, (short)0, bytes_left, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDS_PTR(lds), 16, voff, 0u, 0, 0);
}
__device__ __forceinline__ void lce_dma4(const void* base, int bytes_left, char* lds, uint32_t voff) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, (short)0, bytes_left, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDS_PTR(lds), 4, voff, 0u, 0, 0);
}

template <int I, int N, class F>
__device__ __forceinline__ void lce_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        lce_static_for<I + 1, N>(f);
    }
}

__device__ __forceinline__ uint32_t lce_opaque(uint32_t x) {      // the value, but not loop-invariant to the optimiser
    asm volatile("" : "+v"(x));
    return x;
}
// S accumulators of the first product live in VGPRs (asm: with the O^T accumulators filling all 256 AGPRs the function is compiled in
// hipcc's AGPR form, where EVERY builtin MFMA's C / D is an AGPR -- 32 more than exist).  NOP: wait states between a VALU write of the
// accumulator (its initialisation) and the MFMA that reads it.
template <bool NOP>
__device__ __forceinline__ void lce_mfma_v(f32x16_t& c, const bf16x8_t a, const bf16x8_t b) {
    if constexpr (NOP) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+&v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void lce_mfma_v0(f32x16_t& c, const bf16x8_t a, const bf16x8_t b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(b));
}
// One step of the first product as ONE asm statement: next = LDS[addr + IMM] (read-ahead), optional s_waitcnt lgkmcnt(W) (W < 0: none),
// acc (+)= cur x b.  ZERO: C = literal 0 (the first step of each accumulator).
template <int IMM, int W, bool ZERO>
__device__ __forceinline__ void lce_step_a(bf16x8_t& next, f32x16_t& acc, const bf16x8_t cur, const bf16x8_t b, uint32_t addr) {
    if constexpr (ZERO) {
        if constexpr (W >= 0) asm volatile("ds_read_b128 %0, %4 offset:%5\n\ts_waitcnt lgkmcnt(%6)\n\tv_mfma_f32_32x32x16_bf16 %1, %2, %3, 0" : "=&v"(next), "=&v"(acc) : "v"(cur), "v"(b), "v"(addr), "i"(IMM), "n"(W));
        else asm volatile("ds_read_b128 %0, %4 offset:%5\n\tv_mfma_f32_32x32x16_bf16 %1, %2, %3, 0" : "=&v"(next), "=&v"(acc) : "v"(cur), "v"(b), "v"(addr), "i"(IMM));
    } else {
        if constexpr (W >= 0) asm volatile("ds_read_b128 %0, %4 offset:%5\n\ts_waitcnt lgkmcnt(%6)\n\tv_mfma_f32_32x32x16_bf16 %1, %2, %3, %1" : "=&v"(next), "+&v"(acc) : "v"(cur), "v"(b), "v"(addr), "i"(IMM), "n"(W));
        else asm volatile("ds_read_b128 %0, %4 offset:%5\n\tv_mfma_f32_32x32x16_bf16 %1, %2, %3, %1" : "=&v"(next), "+&v"(acc) : "v"(cur), "v"(b), "v"(addr), "i"(IMM));
    }
}
template <int W, bool ZERO>
__device__ __forceinline__ void lce_step_a_tail(f32x16_t& acc, const bf16x8_t cur, const bf16x8_t b) {      // the last RA - 1 steps: nothing left to read ahead
    if constexpr (ZERO) {
        if constexpr (W >= 0) asm volatile("s_waitcnt lgkmcnt(%3)\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(cur), "v"(b), "n"(W));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(cur), "v"(b));
    } else {
        if constexpr (W >= 0) asm volatile("s_waitcnt lgkmcnt(%3)\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+&v"(acc) : "v"(cur), "v"(b), "n"(W));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+&v"(acc) : "v"(cur), "v"(b));
    }
}
template <int IMM>
__device__ __forceinline__ bf16x8_t lce_read128(uint32_t lds_addr) {      // ds_read_b128 hipcc neither hoists nor waits for: the caller counts lgkmcnt
    bf16x8_t v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "i"(IMM));
    return v;
}

// DH: feature width (64 / 128 / 256 / 512).  MODE 0: forward sweep (online softmax; WITH_ACC: also O^T = softmax-weighted sum of the
// streamed rows).  MODE 1: backward sweep for the stationary rows' gradient (P from the saved lse; WITH_ACC always).
//
// Tile = 32 streamed rows (KT), ring of four LDS stages, LDS-DMA two tiles ahead (counted vmcnt), one barrier per tile.  Iteration t:
//     A(t)   S^T = T_t A^T, 2 accumulators over even / odd 16-feature steps (no MFMA waits on its predecessor), fragments read by
//            hand-counted ds_read_b128 groups one group ahead;
//     B(t-1) O^T += T_(t-1)^T P_(t-1)^T, 32 MFMAs, WITH the exponentials of tile t spread between them (two per group of four MFMAs) --
//            the softmax VALU work runs in the matrix pipe's shadow instead of between the two products;
// one extra iteration at t = t_end (zero tile, every row masked) drains the last B.  A row maximum that moves by more than 2^8 (rare after
// the first tiles) first finishes B(t-1) on its own, rescales O and l, and then runs the common path with P_(t-1) = 0.
template <int DH, int MODE, bool WITH_ACC, int ABL = 0>
__global__ void __attribute__((amdgpu_waves_per_eu(1, 1))) __launch_bounds__(256) lce_kernel(const LceArgs a) {
    constexpr int KT = 32, ROWB = DH * 2, CPR = DH / 8, TILE = KT * ROWB, KS = DH / 16, DT = DH / 32;
    constexpr int SLABS = TILE / 1024, LPW = SLABS / 4;          // 1-KB LDS-DMA pieces per tile / per wave
    constexpr int RPS = 64 / CPR;                                // rows per piece
    constexpr int NV = (4 / RPS) > 0 ? (4 / RPS) : 1;            // the lane offsets of a wave's pieces repeat with this period ...
    constexpr int VSTEP = NV * 4 * RPS;                          // ... advancing this many rows
    constexpr int STAGE = TILE + 1024;                           // + one 256-B copy of the tile's per-row vector per wave
    constexpr int NK8 = KS < 8 ? KS : 8, ND4 = DT < 4 ? DT : 4;
    static_assert(SLABS % 4 == 0, "every wave stages the same number of pieces");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int split = blockIdx.x / a.n_atiles, atile = blockIdx.x % a.n_atiles;
    const int a0 = atile * 128 + wave * 32;
    const int ai = a0 + (lane & 31);
    const int h = lane >> 5;
    const int t_begin = split * a.tps;
    const int t_end = min(a.ntiles, t_begin + a.tps);

    // stationary fragments: B operand of the first product, lane = (row ai, feature half h)
    bf16x8_t af[KS];
    {
        const bf16_t* arow = a.A + (long long)(ai < a.NA ? ai : 0) * a.a_pitch;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ai < a.NA) af[ks] = *reinterpret_cast<const bf16x8_t*>(arow + ks * 16 + h * 8);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) af[ks][e] = 0;
            }
        }
    }
    f32x16_t oacc[WITH_ACC ? DT : 1];
    if constexpr (WITH_ACC) {
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
    }
    float m_run = -1e30f, l_run = 0.f;                           // MODE 0: running maximum (natural-log units) and sum; MODE 1: l_run = db partial
    float bv = 0.f;                                              // MODE 1: bias of this lane's vocabulary row
    if constexpr (MODE == 1) {
        if (a.avec && ai < a.NA) bv = a.avec[ai];
    }

    // staging offsets (see the file header of attention_lean.hip): LDS slot `slot` of row `row` holds source chunk slot ^ key(row)
    constexpr int NVR = DH == 512 ? 1 : NV;
    uint32_t st_voff[NVR];
    if constexpr (DH == 512) st_voff[0] = (uint32_t)((lane ^ lce_key<CPR>(wave)) * 16);       // row = wave + 4 i: key = (wave << 2) | (i & 3)
    else {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int idx = (wave + 4 * j) * 64 + lane;
            const int row = idx / CPR, slot = idx % CPR;
            st_voff[j] = (uint32_t)((long long)row * a.t_pitch * 2) + (uint32_t)((slot ^ lce_key<CPR>(row)) * 16);
        }
    }
    const int row_bytes = (int)(a.t_pitch * 2);
    const uint32_t vstep_bytes = (uint32_t)((long long)VSTEP * a.t_pitch * 2);
    const int tile_bytes = (int)(KT * a.t_pitch * 2);
    const int all_bytes = (int)((long long)(a.NT - 1) * a.t_pitch * 2) + DH * 2;          // first byte behind the last row (host: < 2^31)
    // every wave issues LPW tile pieces + ONE copy of the per-row vector (its own: no other wave reads it), so the counted waits are uniform
    auto stage = [&](int t, int buf) {
        char* dst = smem + buf * STAGE;
        int left = all_bytes - t * tile_bytes;
        left = (t < t_end && left > 0) ? left : 0;
        const bf16_t* base = a.T + (long long)t * (tile_bytes / 2);
#pragma unroll
        for (int i = 0; i < LPW; ++i)
            if constexpr (DH == 512) {
                const int rb = (wave + 4 * i) * row_bytes, l2 = left - rb;
                lce_dma16((const char*)base + rb, l2 > 0 ? l2 : 0, dst + (wave + 4 * i) * 1024, st_voff[0] ^ (uint32_t)((i & 3) << 4));
            } else lce_dma16(base, left, dst + (wave + 4 * i) * 1024, st_voff[i % NV] + (uint32_t)(i / NV) * vstep_bytes);
        int vleft = (a.tvec_len - t * KT) * 4;
        vleft = (a.tvec && t < t_end && vleft > 0) ? vleft : 0;
        lce_dma4(a.tvec ? (const void*)(a.tvec + (long long)t * KT) : (const void*)a.T, vleft, dst + TILE + wave * 256, (uint32_t)lane * 4u);
    };

    // the first iteration's B reads stage 3 with P = 0: the stage must hold finite numbers
    for (int i = tid; i < STAGE / 16; i += 256) reinterpret_cast<u32x4_t*>(smem + 3 * STAGE)[i] = u32x4_t{0u, 0u, 0u, 0u};
    stage(t_begin, 0);
    stage(t_begin + 1, 1);
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LPW + 1) : "memory");
    __syncthreads();

    // lane offsets of the fragment reads.  Chunk (2 j + h) of row `row` sits at slot (2 j + h) ^ key = ((2 j) ^ (key & 14)) + ((h ^ key) & 1), and
    // the transposed reads' chunk (4 dl + c) of row r at ((4 dl) ^ (key_r & 12)) + ((c ^ key_r) & 3) -- so ONE base and ONE key register per
    // family rebuild every offset with an xor and an add per tile (eight + eight persistent registers otherwise; this kernel has none to spare)
    const uint32_t smem_off = attn_lds_offset(smem);
    uint32_t k_base, k_key, v_base0, v_base1, v_key;
    {
        const int row = lane & 31, key = lce_key<CPR>(row);
        k_base = smem_off + (uint32_t)(row * ROWB + (((h ^ key) & 1) << 4));
        k_key = (uint32_t)((key & 14) << 4);
        const int g16 = lane >> 4, rr = (lane >> 2) & 3, qq = lane & 3;
        const int r0 = 4 * (g16 >> 1) + rr, r1 = r0 + 8;
        const int c = 2 * (g16 & 1) + (qq >> 1);
        v_base0 = smem_off + (uint32_t)(r0 * ROWB + (((c ^ lce_key<CPR>(r0)) & 3) << 4) + (qq & 1) * 8);
        v_base1 = smem_off + (uint32_t)(r1 * ROWB + (((c ^ lce_key<CPR>(r1)) & 3) << 4) + (qq & 1) * 8);
        v_key = (uint32_t)((lce_key<CPR>(r0) & 12) << 4);        // (rows r0 and r0 + 8 share key bits 2..3)
    }
    u32x4_t pw_prev[2] = {u32x4_t{0u, 0u, 0u, 0u}, u32x4_t{0u, 0u, 0u, 0u}};      // P of the previous tile, one MFMA operand per 16-row step

    auto tile_body = [&](auto slot_c, int t) {
        constexpr int SLOT = decltype(slot_c)::value;
        constexpr int KBASE = SLOT * STAGE, PBASE = ((SLOT + 3) & 3) * STAGE;
        constexpr int NP = LPW + 1;                              // DMA pieces a wave issues per tile
        const bool live = t < t_end;
        // With ONE wave per SIMD nothing runs beside this wave but the matrix pipe: whatever is not issued in the shadow of an MFMA is
        // serial time (ablation, 24576 rows: both products at the pipe's pace, 3.3 of 11.3 ms in the code between them).  So:
        //   * the LDS-DMA of tile t + 2 is issued piece by piece BETWEEN the steps of the first product (1.5 instructions per MFMA there);
        //   * the first product starts from literal zeros -- the per-row vector joins in y;
        //   * y, the row maximum and the rescale decision are computed after the first HB MFMAs of the second product have been issued
        //     (they do not depend on tile t), the exponentials between the remaining ones.
        char* dma_dst = smem + ((SLOT + 2) & 3) * STAGE;
        int dma_left = all_bytes - (t + 2) * tile_bytes;
        dma_left = (t + 2 < t_end && dma_left > 0) ? dma_left : 0;
        const bf16_t* dma_base = a.T + (long long)(t + 2) * (tile_bytes / 2);
        int dma_vleft = (a.tvec_len - (t + 2) * KT) * 4;
        dma_vleft = (a.tvec && t + 2 < t_end && dma_vleft > 0) ? dma_vleft : 0;
        auto dma_piece = [&](auto pc) {
            constexpr int i = decltype(pc)::value;
            if constexpr (i < LPW && DH == 512) {
                // one row per piece: the row's byte offset is wave-uniform and moves into the descriptor (base up, bytes left down), the
                // lane part is st_voff[0] ^ ((i & 3) << 4) -- one persistent register instead of four
                const int rb = (wave + 4 * i) * row_bytes;
                const int l2 = dma_left - rb;
                lce_dma16((const char*)dma_base + rb, l2 > 0 ? l2 : 0, dma_dst + (wave + 4 * i) * 1024, lce_opaque(st_voff[0]) ^ (uint32_t)((i & 3) << 4));
            } else if constexpr (i < LPW) lce_dma16(dma_base, dma_left, dma_dst + (wave + 4 * i) * 1024, lce_opaque(st_voff[i % NV]) + (uint32_t)(i / NV) * vstep_bytes);
            else lce_dma4(a.tvec ? (const void*)(a.tvec + (long long)(t + 2) * KT) : (const void*)a.T, dma_vleft, dma_dst + TILE + wave * 256, (uint32_t)lane * 4u);
        };

        // ---- A(t): S^T[streamed row, stationary row] over the DH features, two accumulators over even / odd 16-feature steps ----
        f32x16_t s0, s1;
        {
            // (opaque: hipcc would otherwise hoist all four stages' offsets out of the tile loop -- 64 registers this kernel does not have)
            const uint32_t kb_t = lce_opaque(k_base) + KBASE, kk_t = lce_opaque(k_key);
            uint32_t ko[NK8];
#pragma unroll
            for (int j = 0; j < NK8; ++j) ko[j] = kb_t + ((uint32_t)(32 * j) ^ kk_t);
            // one fragment read per MFMA, issued RA - 1 steps ahead of the MFMA that consumes it
            constexpr int RA = KS < 6 ? KS : 6;
            bf16x8_t tf[RA];
            if constexpr (!(ABL & 4)) lce_static_for<0, RA - 1>([&](auto kc) {
                constexpr int ks = decltype(kc)::value;
                tf[ks % RA] = lce_read128<(ks >> 3) * 256>(ko[ks & 7]);
            });
            if constexpr (ABL & 4) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s0[r] = s1[r] = 0.f;
            }
            // per step ONE asm statement: the read RA - 1 steps ahead, (on even steps) the wait that retires this step's and the next
            // step's fragments, the MFMA -- hipcc pads every asm boundary with an s_nop, and with one wave per SIMD every issue slot counts
            lce_static_for<0, KS>([&](auto kc) {
                constexpr int ks = decltype(kc)::value, nx = ks + RA - 1;
                if constexpr (!(ABL & 4)) {
                    constexpr int left = KS - 1 - ks;                                        // reads issued after this step's own
                    constexpr int pend = (left < RA - 1 ? left : RA - 1) - ((ks & 1) == 0 && left >= 1 ? 1 : 0);
                    constexpr int W = (ks & 1) == 0 || RA < 3 ? pend : -1;                   // odd steps were retired by the even step before
                    f32x16_t& acc = (ks & 1) ? s1 : s0;
                    if constexpr (nx < KS) lce_step_a<(nx >> 3) * 256, W, ks < 2>(tf[nx % RA], acc, tf[ks % RA], af[ks], ko[nx & 7]);
                    else lce_step_a_tail<W, ks < 2>(acc, tf[ks % RA], af[ks]);
                }
                lce_static_for<0, NP>([&](auto pc) {                                         // DMA piece p goes behind step p * KS / NP
                    if constexpr (decltype(pc)::value * KS / NP == ks) dma_piece(pc);
                });
            });
        }

        // ---- B(t - 1): O^T += T_(t-1)^T P_(t-1)^T.  MFMA i: 16-row step s2 = i / DT, feature tile d = i % DT; its two transposed reads are
        //      issued RB - 1 MFMAs ahead, even MFMAs wait for their own and the next MFMA's fragments ----
        constexpr int NB = WITH_ACC && !(ABL & 2) ? 2 * DT : 0, RB = NB < 6 ? (NB > 0 ? NB : 1) : 6;
        constexpr int HB = NB < 4 ? NB : 4;                      // MFMAs issued before tile t's scores are touched
        uint32_t vo[ND4][2];
        bf16x4_t vlo[RB], vhi[RB];
        auto issue_b = [&](auto ic) {
            constexpr int I = decltype(ic)::value, s2 = I / DT, d = I % DT;
            constexpr int IMM = (16 * s2) * ROWB + (d >> 2) * 256;
            vlo[I % RB] = attn_tr16i<IMM>(vo[d & 3][0]);
            vhi[I % RB] = attn_tr16i<IMM>(vo[d & 3][1]);
        };
        auto mfma_b = [&](auto ic) {
            constexpr int I = decltype(ic)::value, s2 = I / DT, d = I % DT;
            if constexpr (I + RB - 1 < NB) issue_b(std::integral_constant<int, I + RB - 1>{});
            constexpr int left = NB - 1 - I;
            constexpr int pend = 2 * ((left < RB - 1 ? left : RB - 1) - ((I & 1) == 0 && left >= 1 ? 1 : 0));
            if constexpr ((I & 1) == 0 || RB < 3) {
                asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(pend) : "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
            const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, pw_prev[s2]);
            asm volatile("" : "+v"(vlo[I % RB]), "+v"(vhi[I % RB]));
            const bf16x8_t vf = __builtin_shufflevector(vlo[I % RB], vhi[I % RB], 0, 1, 2, 3, 4, 5, 6, 7);
            oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oacc[d], 0, 0, 0);
        };
        // per-row vector of tile t (MODE 0: bias; MODE 1: -lse): row r of lane half h is 8 (r >> 2) + 4 h + (r & 3).  Read FIRST: LDS
        // returns in order, so the first counted wait of the MFMAs below also covers these four reads
        f32x4_t tv[4];
        {
            const uint32_t tva = lce_opaque(smem_off + (uint32_t)(TILE + 16 * h)) + (uint32_t)KBASE + (uint32_t)wave * 256u;
            lce_static_for<0, 4>([&](auto gc) { tv[decltype(gc)::value] = __builtin_bit_cast(f32x4_t, lce_read128<32 * decltype(gc)::value>(tva)); });
        }
        if constexpr (NB > 0) {
            const uint32_t vb0_t = lce_opaque(v_base0) + PBASE, vb1_t = lce_opaque(v_base1) + PBASE, vk_t = lce_opaque(v_key);
#pragma unroll
            for (int dl = 0; dl < ND4; ++dl) {
                vo[dl][0] = vb0_t + ((uint32_t)(64 * dl) ^ vk_t);
                vo[dl][1] = vb1_t + ((uint32_t)(64 * dl) ^ vk_t);
            }
            lce_static_for<0, RB - 1>([&](auto ic) { issue_b(ic); });
            lce_static_for<0, HB>([&](auto ic) { mfma_b(ic); });
            __builtin_amdgcn_sched_barrier(0);
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        // the rest of the second product with `fill(e)`, e = 0..15, dealt over its MFMAs
        auto tail_b = [&](auto&& fill) {
            if constexpr (NB > HB) {
                constexpr int NT_ = NB - HB;
                lce_static_for<HB, NB>([&](auto ic) {
                    constexpr int I = decltype(ic)::value, J = I - HB;
                    mfma_b(ic);
                    lce_static_for<0, 16>([&](auto ec) {             // exponential e goes behind tail MFMA e * NT_ / 16
                        if constexpr (decltype(ec)::value * NT_ / 16 == J) fill(ec);
                    });
                });
                __builtin_amdgcn_sched_barrier(0);
            } else {
                lce_static_for<0, 16>([&](auto ec) { fill(ec); });
            }
        };

        // the first product's results meet the VALU here: hipcc pads nothing behind an asm MFMA (8 passes: 11+ wait states; with the HB
        // MFMAs above in between the s_nop is a formality)
        asm volatile("s_nop 7" : "+v"(s0), "+v"(s1), "+v"(tv[0]), "+v"(tv[1]), "+v"(tv[2]), "+v"(tv[3]));
        f32x16_t y;
        {
            const float addv = MODE == 1 ? (live ? bv : -INFINITY) : 0.f;     // MODE 1: bias of this lane's vocabulary row (-inf in the drain iteration)
#pragma unroll
            for (int r = 0; r < 16; ++r) y[r] = (s0[r] + s1[r]) + (MODE == 1 ? tv[r >> 2][r & 3] + addv : tv[r >> 2][r & 3]);
        }

        float pe[16];
        float ps0 = 0.f, ps1 = 0.f;
        u32x4_t pw_new[2];
        float mc = 0.f;
        auto fill = [&](auto ec) {
            constexpr int E = decltype(ec)::value;
            if constexpr (ABL & 1) pe[E] = y[E];
            else pe[E] = __builtin_amdgcn_exp2f(MODE == 0 ? __builtin_fmaf(y[E], LCE_LOG2E, -mc) : y[E] * LCE_LOG2E);
            if constexpr (E & 1) {
                ps1 += pe[E];
                pw_new[E >> 3][(E >> 1) & 3] = pack_bf16x2(pe[E - 1], pe[E]);
            } else {
                ps0 += pe[E];
            }
        };
        if constexpr (MODE == 0) {
            const int k0 = t * KT;
            if (k0 + KT > a.NT || !live) {                       // streamed rows past the end (or the drain iteration): these logits do not exist
                const int lim = (live ? a.NT - k0 : 0) - 4 * h;
#pragma unroll
                for (int r = 0; r < 16; ++r) y[r] = ((r & 3) + 8 * (r >> 2)) < lim ? y[r] : -INFINITY;
            }
            float tmax;
            {
                const float x0 = fmaxf(fmaxf(y[0], y[1]), y[2]), x1 = fmaxf(fmaxf(y[3], y[4]), y[5]), x2 = fmaxf(fmaxf(y[6], y[7]), y[8]);
                const float x3 = fmaxf(fmaxf(y[9], y[10]), y[11]), x4 = fmaxf(fmaxf(y[12], y[13]), y[14]);
                tmax = fmaxf(fmaxf(fmaxf(x0, x1), fmaxf(x2, x3)), fmaxf(x4, y[15]));
            }
            tmax = lce_xmax32(tmax);
            const float m_new = fmaxf(m_run, tmax);
            // deferred maximum (cdna_hip_programming.md T13): rescale only when some row's maximum moved by more than 2^8 (the first tile does).
            // Everything still at the old scale -- the rest of P_(t-1) T_(t-1) -- goes into O first; the common path then repeats those MFMAs with P = 0.
            if (__builtin_amdgcn_ballot_w64((m_new - m_run) * LCE_LOG2E > 8.f) != 0) {
                tail_b([](auto) {});
                const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * LCE_LOG2E);
                l_run *= alpha;
                if constexpr (WITH_ACC) {
#pragma unroll
                    for (int d = 0; d < DT; ++d) {               // one accumulator block at a time (the scheduler would otherwise pull all 256
#pragma unroll                                                   // accumulator registers into VGPRs at once)
                        for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                m_run = m_new;
                pw_prev[0] = pw_prev[1] = u32x4_t{0u, 0u, 0u, 0u};   // the common path below then adds zeros (the fragments it re-reads are finite)
            }
            mc = m_run * LCE_LOG2E;
        }
        tail_b(fill);
        if constexpr (MODE == 0) l_run += lce_xsum32(ps0 + ps1);
        else l_run += ps0 + ps1;                                 // this lane's half of the column sum; the halves meet in the epilogue
        pw_prev[0] = pw_new[0];
        pw_prev[1] = pw_new[1];

        if constexpr (!(ABL & 8)) {
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LPW + 1) : "memory");
            __builtin_amdgcn_s_barrier();
        }
        asm volatile("" ::: "memory");
    };
    for (int t = t_begin; t <= t_end; t += 4) {
        tile_body(std::integral_constant<int, 0>{}, t);
        if (t + 1 <= t_end) tile_body(std::integral_constant<int, 1>{}, t + 1);
        if (t + 2 <= t_end) tile_body(std::integral_constant<int, 2>{}, t + 2);
        if (t + 3 <= t_end) tile_body(std::integral_constant<int, 3>{}, t + 3);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- epilogue ----
    if constexpr (MODE == 0) {
        const long long prow = (long long)split * a.Apad + ai;
        if (h == 0) {
            a.part_ml[prow * 2 + 0] = m_run;
            a.part_ml[prow * 2 + 1] = l_run;
        }
        if constexpr (WITH_ACC) {
            float* orow = a.part_o + prow * DH + 4 * h;
#pragma unroll
            for (int d = 0; d < DT; ++d)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4_t f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) f[e] = oacc[d][4 * g + e];
                    *reinterpret_cast<f32x4_t*>(orow + d * 32 + g * 8) = f;
                }
        }
    } else {
        const float sc = a.scale[0];
        const float colsum = lce_xsum32(l_run);
        if (ai < a.NA) {
            float* wrow = a.dW + (long long)ai * DH + 4 * h;
            if (a.nsplit == 1) {
#pragma unroll
                for (int d = 0; d < DT; ++d)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4_t* p = reinterpret_cast<f32x4_t*>(wrow + d * 32 + g * 8);
                        f32x4_t f = *p;
#pragma unroll
                        for (int e = 0; e < 4; ++e) f[e] += sc * oacc[d][4 * g + e];
                        *p = f;
                    }
                if (a.db && h == 0) a.db[ai] += sc * colsum;
            } else {
#pragma unroll
                for (int d = 0; d < DT; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) unsafeAtomicAdd(wrow + d * 32 + (r >> 2) * 8 + (r & 3), sc * oacc[d][r]);
                if (a.db && h == 0) unsafeAtomicAdd(a.db + ai, sc * colsum);
            }
        }
    }
}

// One wave per row: merge the vocabulary splits of the forward sweep.
//   lse = gmax + log(sum_s l_s e^(m_s - gmax));  target logit = <h[m], W[t]> + b[t] (fp32);  *loss_sum += lse - target logit (valid rows);
//   row_e[m] = -lse (or -inf for rows that are switched off / pad rows up to the next multiple of 64);
//   dh_f32[m] = sum_s O_s e^(m_s - gmax) / L - W[t]   (zeros for rows that are switched off)
constexpr int LCE_CROWS = 8;                                     // rows per wave of the combine kernel
template <int DH>
__global__ void __launch_bounds__(256) lce_combine_kernel(const float* __restrict__ part_ml, const float* __restrict__ part_o, int nsplit, int Apad,
                                                          const bf16_t* __restrict__ hmat, long long h_pitch, int M, int Mpad64,
                                                          const bf16_t* __restrict__ W, long long w_pitch, int V, const float* __restrict__ bias,
                                                          const long long* __restrict__ target, const unsigned char* __restrict__ valid,
                                                          float* __restrict__ row_lse, float* __restrict__ row_e, float* __restrict__ loss_sum,
                                                          float* __restrict__ dh) {
    constexpr int EPL = DH / 64;                                 // features per lane
    // A wave walks LCE_CROWS rows and the workgroup adds its loss terms with ONE atomic: a per-row atomicAdd on the one loss word serialises in L2
    // (24576 of them: 0.25 of this kernel's 0.33 ms).
    __shared__ float wloss[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float loss_acc = 0.f;
    for (int rr = 0; rr < LCE_CROWS; ++rr) {
        const int m = (blockIdx.x * 4 + wave) * LCE_CROWS + rr;
        if (m >= Mpad64) break;
        if (m >= M) {
            if (lane == 0) row_e[m] = -INFINITY;
            continue;
        }
        float gmax = -INFINITY;
        for (int s = 0; s < nsplit; ++s) gmax = fmaxf(gmax, part_ml[((long long)s * Apad + m) * 2]);
        float L = 0.f;
        for (int s = 0; s < nsplit; ++s) {
            const float* ml = part_ml + ((long long)s * Apad + m) * 2;
            L += ml[1] * __expf(ml[0] - gmax);
        }
        const float lse = gmax + __logf(L);
        const bool on = !valid || valid[m];
        const long long t = target[m];
        const bool tok = t >= 0 && t < V;
        float hv[EPL], wv[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            hv[e] = bf16_to_f32(hmat[(long long)m * h_pitch + lane * EPL + e]);
            wv[e] = tok ? bf16_to_f32(W[t * w_pitch + lane * EPL + e]) : 0.f;
        }
        float dot = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) dot = __builtin_fmaf(hv[e], wv[e], dot);
        dot = wave_sum(dot);
        if (lane == 0) {
            row_lse[m] = lse;
            row_e[m] = on ? -lse : -INFINITY;
            // F.cross_entropy raises on a target outside [0, V); a kernel cannot, so the loss is poisoned instead (as genie_masked_ce_fwd)
            if (on) loss_acc += tok ? lse - (dot + (bias ? bias[t] : 0.f)) : __builtin_nanf("");
        }
        if (dh) {
            float acc[EPL];
#pragma unroll
            for (int e = 0; e < EPL; ++e) acc[e] = 0.f;
            if (on) {
                for (int s = 0; s < nsplit; ++s) {
                    const long long pr = (long long)s * Apad + m;
                    const float w = __expf(part_ml[pr * 2] - gmax);
#pragma unroll
                    for (int e = 0; e < EPL; ++e) acc[e] = __builtin_fmaf(part_o[pr * DH + lane * EPL + e], w, acc[e]);
                }
                const float inv = 1.f / L;
#pragma unroll
                for (int e = 0; e < EPL; ++e) acc[e] = acc[e] * inv - wv[e];
            }
#pragma unroll
            for (int e = 0; e < EPL; ++e) dh[(long long)m * DH + lane * EPL + e] = acc[e];
        }
    }
    if (lane == 0) wloss[wave] = loss_acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t = (wloss[0] + wloss[1]) + (wloss[2] + wloss[3]);
        if (t != 0.f) atomicAdd(loss_sum, t);                    // (NaN != 0: the poison gets through)
    }
}

// dh_bf16[m] = dh_f32[m] * *scale
__global__ void __launch_bounds__(256) lce_scale_cast_kernel(const float* __restrict__ src, const float* __restrict__ scale, bf16_t* __restrict__ dst,
                                                             long long dst_pitch, long long M, int D) {
    const float sc = scale[0];
    const int per = D >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < M * per; i += (long long)gridDim.x * 256) {
        const long long m = i / per;
        const int c = (int)(i % per) * 4;
        const f32x4_t f = *reinterpret_cast<const f32x4_t*>(src + m * D + c);
        u32x2_t o;
        o[0] = pack_bf16x2(f[0] * sc, f[1] * sc);
        o[1] = pack_bf16x2(f[2] * sc, f[3] * sc);
        *reinterpret_cast<u32x2_t*>(dst + m * dst_pitch + c) = o;
    }
}

// The one-hot term of the weight / bias gradient: dW[target[m]] -= scale h[m], db[target[m]] -= scale, for the rows that are switched on.
// A wave walks 32 consecutive rows and keeps the sum of a run of equal targets in registers (the reference's compute_loss reads its
// targets after the masked fill, so they are ALL equal: one flush per wave instead of 32 x D atomics on one row).
template <int DH>
__global__ void __launch_bounds__(256) lce_onehot_kernel(const bf16_t* __restrict__ hmat, long long h_pitch, int M, int V,
                                                         const long long* __restrict__ target, const float* __restrict__ row_e,
                                                         const float* __restrict__ scale, float* __restrict__ dW, float* __restrict__ db) {
    constexpr int EPL = DH / 64;
    const int lane = threadIdx.x & 63;
    const int m0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 32;
    const float sc = -scale[0];
    float acc[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) acc[e] = 0.f;
    float cnt = 0.f;
    long long cur = -1;
    auto flush = [&]() {
        if (cur >= 0 && cnt > 0.f) {
#pragma unroll
            for (int e = 0; e < EPL; ++e) unsafeAtomicAdd(dW + cur * DH + lane * EPL + e, sc * acc[e]);
            if (db && lane == 0) unsafeAtomicAdd(db + cur, sc * cnt);
        }
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc[e] = 0.f;
        cnt = 0.f;
    };
    for (int m = m0; m < min(M, m0 + 32); ++m) {
        const long long t = target[m];
        if (!(row_e[m] > -INFINITY) || t < 0 || t >= V) continue;       // wave-uniform
        if (t != cur) { flush(); cur = t; }
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc[e] += bf16_to_f32(hmat[(long long)m * h_pitch + lane * EPL + e]);
        cnt += 1.f;
    }
    flush();
}

struct LcePlan { int n_atiles, nsplit, tps, ntiles, Apad; };

// rows of the stationary matrix in 128-row tiles; the streamed matrix in 32-row tiles, split over blocks.  One block per CU is resident (one wave per
// SIMD), blocks cost the same, so a launch runs in ROUNDS of (number of CUs) blocks and a last round of four blocks costs as much as a full one:
// round 5 always asked for ~768 blocks -- 24576 gathered rows are 192 row tiles x 4 splits = 768 = three rounds exactly, but the 24 6xx rows a
// Bernoulli(0.75) mask actually leaves are 193 x 4 = 772 blocks = FOUR rounds (the 14.2 ms in the training step against 10.97 ms in the stand-alone
// bench).  Now the split count minimises rounds / splits (the time in units of one un-split sweep) plus a small charge per split for
// its partial (max, sum, O) tile: 193 row tiles -> 5 splits = 965 blocks = four rounds of one fifth each (0.80 against 1.00).
static int lce_cu_count() {
    static int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1) v = 256;
        return v;
    }();
    return n;
}
LcePlan lce_plan(long long n_station, long long n_stream) {
    LcePlan p;
    p.n_atiles = (int)((n_station + 127) / 128);
    p.ntiles = (int)((n_stream + 31) / 32);
    const int cus = lce_cu_count();
    int cap = p.ntiles / 8 > 0 ? p.ntiles / 8 : 1;                 // every split keeps at least 8 tiles
    if (cap > 16) cap = 16;
    int want = 1;
    double best = 1e30;
    for (int ns = 1; ns <= cap; ++ns) {
        const long long blocks = (long long)p.n_atiles * ns;
        const double cost = (double)((blocks + cus - 1) / cus) / ns + 0.004 * ns;
        if (cost < best - 1e-9) { best = cost; want = ns; }
    }
    p.tps = (p.ntiles + want - 1) / want;
    p.nsplit = (p.ntiles + p.tps - 1) / p.tps;
    p.Apad = p.n_atiles * 128;
    return p;
}

template <int DH, int MODE, bool ACC>
int lce_launch(const LceArgs& a, hipStream_t s) {
    constexpr int LDS = 4 * (32 * DH * 2 + 1024);
    auto k = lce_kernel<DH, MODE, ACC>;
#ifdef LCE_ABLATE                      // timing probe (wrong results): GENIE_LCE_ABL = bit mask, see lce_kernel's ABL
    if constexpr (DH == 512 && MODE == 1) {
        const char* e = getenv("GENIE_LCE_ABL");
        const int abl = e ? atoi(e) : 0;
        constexpr int LDSA = 4 * (32 * DH * 2 + 1024);
#define LCE_ABL_CASE(N) if (abl == N) { auto ka = lce_kernel<DH, MODE, ACC, N>; hipFuncSetAttribute((const void*)ka, hipFuncAttributeMaxDynamicSharedMemorySize, LDSA); \
            ka<<<dim3((unsigned)(a.n_atiles * a.nsplit)), 256, LDSA, s>>>(a); return GENIE_OK; }
        LCE_ABL_CASE(1) LCE_ABL_CASE(2) LCE_ABL_CASE(4) LCE_ABL_CASE(6) LCE_ABL_CASE(8) LCE_ABL_CASE(9) LCE_ABL_CASE(3)
#undef LCE_ABL_CASE
    }
#endif
    static bool attr_done = false;                               // per instantiation
    if (!attr_done) {
        const hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) {
            genie_set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return GENIE_ERR_HIP;
        }
        attr_done = true;
    }
    k<<<dim3((unsigned)(a.n_atiles * a.nsplit)), 256, LDS, s>>>(a);
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}

template <int MODE, bool ACC>
int lce_dispatch(int D, const LceArgs& a, hipStream_t s) {
    switch (D) {
        case 512: return lce_launch<512, MODE, ACC>(a, s);
        case 256: return lce_launch<256, MODE, ACC>(a, s);
        case 128: return lce_launch<128, MODE, ACC>(a, s);
        case 64: return lce_launch<64, MODE, ACC>(a, s);
    }
    genie_set_error("genie_linear_ce: D = %d (supported: 64, 128, 256, 512)", D);
    return GENIE_ERR_ARG;
}

bool lce_shapes_ok(int64_t M, int D, int64_t V, int64_t h_pitch, int64_t w_pitch) {
    return (D == 64 || D == 128 || D == 256 || D == 512) && M >= 1 && V >= 1 && h_pitch >= D && w_pitch >= D && h_pitch % 8 == 0 && w_pitch % 8 == 0 &&
           (M + 192) * h_pitch * 2 < (1ll << 31) && (V + 192) * w_pitch * 2 < (1ll << 31);
}

}  // namespace

extern "C" int genie_linear_ce_supported(int64_t M, int D, int64_t V, int64_t h_pitch, int64_t w_pitch) {
    return lce_shapes_ok(M, D, V, h_pitch, w_pitch) ? 1 : 0;
}

extern "C" int64_t genie_linear_ce_ws_floats(int64_t M, int D, int64_t V, int with_grad) {
    const LcePlan p = lce_plan(M, V);
    return (int64_t)p.nsplit * p.Apad * (2 + (with_grad ? D : 0));
}

extern "C" int genie_linear_ce_fwd(const void* h_bf16, int64_t h_pitch, int64_t M, int D, const void* w_bf16, int64_t w_pitch, int64_t V,
                                   const float* bias, const int64_t* target, const unsigned char* valid, float* ws, int64_t ws_floats,
                                   float* row_lse, float* row_e, float* loss_sum, float* dh_f32, void* stream) {
    GENIE_CHECK_ARG(h_bf16 && w_bf16 && target && ws && row_lse && row_e && loss_sum, "genie_linear_ce_fwd: null pointer");
    GENIE_CHECK_ARG(lce_shapes_ok(M, D, V, h_pitch, w_pitch),
                    "genie_linear_ce_fwd: unsupported shape M=%lld D=%d V=%lld pitches %lld / %lld (D in {64,128,256,512}, pitches multiples of 8, "
                    "(rows + 192) * pitch * 2 < 2^31)", (long long)M, D, (long long)V, (long long)h_pitch, (long long)w_pitch);
    GENIE_CHECK_ARG(((uintptr_t)h_bf16 & 15) == 0 && ((uintptr_t)w_bf16 & 15) == 0, "genie_linear_ce_fwd: h / W must be 16-byte aligned");
    const LcePlan p = lce_plan(M, V);
    const bool grad = dh_f32 != nullptr;
    GENIE_CHECK_ARG(ws_floats >= (int64_t)p.nsplit * p.Apad * (2 + (grad ? D : 0)), "genie_linear_ce_fwd: workspace too small (genie_linear_ce_ws_floats)");
    LceArgs a{};
    a.A = (const bf16_t*)h_bf16; a.a_pitch = h_pitch; a.NA = (int)M;
    a.T = (const bf16_t*)w_bf16; a.t_pitch = w_pitch; a.NT = (int)V;
    a.tvec = bias; a.tvec_len = (int)V; a.avec = nullptr;
    a.n_atiles = p.n_atiles; a.nsplit = p.nsplit; a.tps = p.tps; a.ntiles = p.ntiles; a.Apad = p.Apad;
    a.part_ml = ws; a.part_o = ws + (int64_t)p.nsplit * p.Apad * 2;
    const hipStream_t s = (hipStream_t)stream;
    const int rc = grad ? lce_dispatch<0, true>(D, a, s) : lce_dispatch<0, false>(D, a, s);
    if (rc != GENIE_OK) return rc;
    const int Mpad64 = (int)((M + 63) / 64 * 64);
    const unsigned cgrid = (unsigned)((Mpad64 + 4 * LCE_CROWS - 1) / (4 * LCE_CROWS));
#define LCE_COMBINE(DH_)                                                                                                                    \
    lce_combine_kernel<DH_><<<cgrid, 256, 0, s>>>(a.part_ml, grad ? a.part_o : nullptr, p.nsplit, p.Apad, a.A, h_pitch, (int)M, Mpad64, a.T, \
                                                 w_pitch, (int)V, bias, (const long long*)target, valid, row_lse, row_e, loss_sum, dh_f32)
    switch (D) {
        case 512: LCE_COMBINE(512); break;
        case 256: LCE_COMBINE(256); break;
        case 128: LCE_COMBINE(128); break;
        default: LCE_COMBINE(64); break;
    }
#undef LCE_COMBINE
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}

extern "C" int genie_linear_ce_bwd(const void* h_bf16, int64_t h_pitch, int64_t M, int D, const void* w_bf16, int64_t w_pitch, int64_t V,
                                   const float* bias, const int64_t* target, const float* row_e, const float* scale, const float* dh_f32,
                                   void* dh_bf16, int64_t dh_pitch, float* dW, float* dbias, void* stream) {
    GENIE_CHECK_ARG(h_bf16 && w_bf16 && target && row_e && scale, "genie_linear_ce_bwd: null pointer");
    GENIE_CHECK_ARG(lce_shapes_ok(M, D, V, h_pitch, w_pitch), "genie_linear_ce_bwd: unsupported shape M=%lld D=%d V=%lld", (long long)M, D, (long long)V);
    GENIE_CHECK_ARG((dh_bf16 == nullptr) == (dh_f32 == nullptr) && (!dh_bf16 || (dh_pitch >= D && dh_pitch % 4 == 0)), "genie_linear_ce_bwd: dh_f32 / dh_bf16 / dh_pitch");
    const hipStream_t s = (hipStream_t)stream;
    if (dh_bf16) {
        const long long n = M * (D / 4);
        const unsigned grid = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
        lce_scale_cast_kernel<<<grid, 256, 0, s>>>(dh_f32, scale, (bf16_t*)dh_bf16, dh_pitch, M, D);
        GENIE_CHECK_LAUNCH();
    }
    if (!dW) return GENIE_OK;
    const LcePlan p = lce_plan(V, M);
    LceArgs a{};
    a.A = (const bf16_t*)w_bf16; a.a_pitch = w_pitch; a.NA = (int)V;
    a.T = (const bf16_t*)h_bf16; a.t_pitch = h_pitch; a.NT = (int)M;
    a.tvec = row_e; a.tvec_len = (int)((M + 63) / 64 * 64); a.avec = bias;
    a.n_atiles = p.n_atiles; a.nsplit = p.nsplit; a.tps = p.tps; a.ntiles = p.ntiles; a.Apad = p.Apad;
    a.dW = dW; a.db = dbias; a.scale = scale;
    const int rc = lce_dispatch<1, true>(D, a, s);
    if (rc != GENIE_OK) return rc;
    const unsigned ogrid = (unsigned)((M + 127) / 128);
#define LCE_ONEHOT(DH_) lce_onehot_kernel<DH_><<<ogrid, 256, 0, s>>>(a.T, h_pitch, (int)M, (int)V, (const long long*)target, row_e, scale, dW, dbias)
    switch (D) {
        case 512: LCE_ONEHOT(512); break;
        case 256: LCE_ONEHOT(256); break;
        case 128: LCE_ONEHOT(128); break;
        default: LCE_ONEHOT(64); break;
    }
#undef LCE_ONEHOT
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}
