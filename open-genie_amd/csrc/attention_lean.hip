// Register-lean MFMA attention kernels for d_head = 64 (reference genie/module/attention.py:199-239; every SpaceTimeAttention of the
// shipped blueprints has 64-wide heads).  Same tiling, LDS ring, swizzle and arithmetic as attention.hip's general kernels -- 32 rows per
// wave, 64-row tiles DMA'd two ahead into a ring of three stages, exp2-domain softmax, transposing LDS reads -- re-cut for OCCUPANCY:
//
//   forward   165 -> 128 VGPRs : four waves per SIMD instead of three
//   backward  238 / 228 -> <= 168 VGPRs : three waves per SIMD instead of two
//
// which is also what the grid asks for: a 4096-token frame of the LatentAction model is 2048 blocks of four waves on 256 CUs --
// with three resident blocks per CU that is 3 + 3 + 2 rounds, with four it is exactly two (backward: four rounds -> three).
// What pays for the registers:
//   * fragment offsets are kept once per k-step / d-tile; the (32-row half, 16-row step, ring slot) part is an instruction
//     immediate (the swizzle keys only use row bits 0..3, so rows r, r + 16, r + 32 share one lane offset);
//   * K / V / Q / dO tiles are staged by BUFFER-addressed LDS-DMA: the lane's byte offset inside a tile is loop-invariant, the
//     tile's base and the bytes left behind it live in a descriptor rebuilt on the scalar unit per tile, rows past the end read
//     zeros from the range check -- no per-lane predicate, no 64-bit per-lane address arithmetic in the loop;
//   * the 64-row tile is consumed as two 32-row halves whose fragments are live one half at a time; P / dS are packed to bf16 as
//     they leave the exponential;
//   * lane ^ 32 exchanges are v_permlane32_swap_b32 (no ds_bpermute round trip, no address register);
//   * this file is compiled with -fno-slp-vectorize: hipcc's SLP pass pairs the fp32 row-sum / scale operations into v_pk_*_f32
//     on 64-bit register pairs, which cost the forward kernel 18 spilled registers at this budget (and v_pk_add_f32 is the
//     slower form beside MFMAs, /opt/skills/guides/MI355X_MICROARCH.md "price of one filler").
// Spills are not an option here: a scratch reload is a VMEM load, and the s_waitcnt vmcnt(0) hipcc puts behind it drains the
// K / V prefetch ring every tile.
#include "common.h"
#include "genie_hip.h"
#include "attn_args.h"
#include "attn_common.h"

static __device__ __attribute__((aligned(256))) uint32_t g_zero_page_l[64];      // source of the lse / D DMA of query rows past the end

// Forward, register-lean form (<= 128 VGPRs -> FOUR waves per SIMD instead of three).  Same tiling, ring and arithmetic as
// attn_fwd_kernel; what changed is what a lane keeps alive: (1) fragment offsets are held once per k-step / per d-tile and the
// (32-key half, 16-key step, ring slot) part rides in the instruction's immediate offset -- the swizzle keys only use row bits
// 0..3, so rows r, r + 16 and r + 32 share one lane offset; (2) P is packed to bf16 as it leaves the exponential (16 registers
// instead of 32 by the time the second product starts); (3) the V^T fragments of the second 32-key half are requested only
// when the first half's products are being issued.  With 2048 blocks of 4 waves on 256 CUs (S = 4096, 16 frames x 4 heads)
// four resident blocks per CU also turn 2.67 rounds (3 + 3 + 2) into exactly two.
// value of the partner lane (lane ^ 32) combined with this lane's by max / add: one v_permlane32_swap_b32 (VALU, no LDS round trip,
// no address register) instead of ds_bpermute_b32.  Inline asm: given two copies of ONE value the builtin's result is folded by
// hipcc as if the swap were the identity on equal operands.  `s_nop 1` covers the VALU-write -> permlane-read hazard.
__device__ __forceinline__ void attn_swap32(float& a, float& b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float attn_xmax32(float x) { float a = x, b = x; attn_swap32(a, b); return fmaxf(a, b); }
__device__ __forceinline__ float attn_xsum32(float x) { float a = x, b = x; attn_swap32(a, b); return a + b; }
// One 16-B-per-lane LDS-DMA piece through a buffer descriptor {base, bytes behind base}: lanes whose offset is >= `bytes_left` get zeros.
// (A __device__ function of its own: the builtins do not exist for the host pass, and used inside a lambda of a __global__ template they
// make hipcc drop the kernel's host stub without a diagnostic.)
__device__ __forceinline__ void attn_dma16(const void* base, int bytes_left, char* lds, uint32_t voff) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, (short)0, bytes_left, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDS_PTR(lds), 16, voff, 0u, 0, 0);
}
__device__ __forceinline__ void attn_dma4(const float* src, char* lds) {      // 4 B per lane, flat address
    __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(lds), 4, 0, 0);
}
// Workgroup -> (sequence, head, tile).  The hardware deals consecutive workgroup ids round-robin over the 8 XCDs, each with its own 4-MB L2.
// With the plain (x = sequence * tiles + tile, y = head) grid the 32 query tiles of one (sequence, head) -- which all stream the SAME
// 512 KB of K / V (S = 4096) -- land on all 8 XCDs, every L2 holds pieces of every sequence at once (16 MB for the blocks resident on
// one XCD) and the tiles come from the Infinity Cache.  Swizzled: XCD c owns the c-th contiguous eighth of the (sequence, head, tile)
// list, so the ~128 blocks resident on it belong to about four (sequence, head) pairs: 2 MB of K / V, L2-resident.
// Launch: 1-D grid of 8 * ceil(N / 8) blocks; blocks whose list position is past N return at once.
struct AttnBlock { int seq, head, tile; bool live; };
__device__ __forceinline__ AttnBlock attn_block_of(int swizzle, int nseq, int nhead, int tiles) {
    AttnBlock b;
    if (!swizzle) {
        b.seq = blockIdx.x / tiles; b.tile = blockIdx.x % tiles; b.head = blockIdx.y; b.live = true;
        return b;
    }
    const int n = nseq * nhead * tiles, per = (n + 7) >> 3;
    const int pos = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    b.live = (int)(blockIdx.x >> 3) < per && pos < n;
    b.tile = pos % tiles;
    const int sh = pos / tiles;
    b.head = sh % nhead;
    b.seq = sh / nhead;
    return b;
}
template <int I, int N, class F>
__device__ __forceinline__ void attn_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        attn_static_for<I + 1, N>(f);
    }
}
// DEFER: the running maximum only moves -- and O, l are only
// rescaled -- when some lane's tile maximum exceeds it by more than 2^8 in the exp2 domain (defer-max, cdna_hip_programming.md T13):
// P is then bounded by 2^8 instead of 1, which neither fp32 sums nor bf16 P mind; on i.i.d. scores the exact rule rescales on two
// tiles out of three (any of 32 queries meeting a new maximum), 35 VALU instructions each, in a loop that is VALU-bound.
template <int DH, int NW, bool KVSAME, int DEFER>
__global__ void __attribute__((amdgpu_waves_per_eu(4, 4))) __launch_bounds__(64 * NW) attn_fwd4_kernel(const AttnArgs a) {
    constexpr int KT = 64, ROWB = DH * 2, CPR = DH / 8, TILE = KT * ROWB, KS = DH / 16, DT = DH / 32;
    // SUP: 64-key tiles per ring stage.  2 for the sum-triggered self-attention form (round 6): a stage is two tiles = 128 keys, so the barrier, the counted
    // wait and the DMA issue block come once per 32 MFMAs instead of once per 16 (LDS: 3 x 16 KB per workgroup); 1 elsewhere
    constexpr int SUP = (DEFER == 2 && KVSAME && NW == 8) ? 2 : 1;      // (DEFER == 3: the sum-triggered rule with one tile per stage, for A/B)
    // (four-wave workgroups would stage twice the pieces per wave: two more offset registers, which spill at this budget)
    constexpr int SLABS = TILE / 1024, LPW = SLABS / NW, LPT = SUP * (KVSAME ? LPW : 2 * LPW);
    constexpr int STAGE = SUP * (KVSAME ? TILE : 2 * TILE);
    static_assert(SLABS % NW == 0, "every wave stages the same number of pieces (counted vmcnt)");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qtiles = (a.Sq + 32 * NW - 1) / (32 * NW);
    const AttnBlock blk = attn_block_of(a.xcd_swizzle, a.nseq, a.nhead, qtiles);
    if (!blk.live) return;
    const int seq = blk.seq, qtile = blk.tile, head = blk.head;
    const int q0 = qtile * (32 * NW) + wave * 32;
    const int qi = q0 + (lane & 31);
    const int h = lane >> 5;

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
    float m_run = -1e30f, l_run = 0.f;
    const float c2 = a.scale * 1.4426950408889634f;

    const int blk_q_max = qtile * (32 * NW) + 32 * NW - 1;
    int k_end = a.Sk;
    if (a.causal && blk_q_max + 1 < k_end) k_end = blk_q_max + 1;
    const int ntile = (k_end + KT - 1) / KT;

    // staging: buffer-addressed LDS-DMA.  A lane's byte offset inside a 64-key tile is loop-invariant; the tile's base address and the
    // bytes that remain behind it are wave-uniform and live in the descriptor (rebuilt on the scalar unit per tile), so rows past
    // the last key -- and whole tiles past the last tile -- read zeros from the descriptor's range check: no per-lane predicate, no
    // 64-bit per-lane address arithmetic in the loop
    const bf16_t* kseq = a.k + seq_base(a.km, seq) + head * DH;
    const bf16_t* vseq = a.v + seq_base(a.km, seq) + head * DH;
    // (host side guarantees (Sk + 192) * pos_stride * 2 < 2^31, so the byte counts below fit an int)
    const int tile_bytes = (int)(KT * a.km.pos_stride * 2);
    const int seq_bytes = (int)((a.Sk - 1) * a.km.pos_stride * 2) + DH * 2;                // first byte behind the last key's head slice
    uint32_t st_voff[SUP * LPW];
#pragma unroll
    for (int i = 0; i < SUP * LPW; ++i) {
        const int idx = (wave + i * NW) * 64 + lane;          // (rows 64 .. 127 of a two-tile stage: the second tile, same swizzle -- its keys use row bits 0 .. 3)
        const int row = idx / CPR;
        st_voff[i] = (uint32_t)((long long)row * a.km.pos_stride * 2) + (uint32_t)(attn_swz<CPR>(row, idx % CPR) * 16);
    }
    const int nstage = (ntile + SUP - 1) / SUP;
    auto stage = [&](int t, int buf) {                       // t: stage index (SUP tiles of 64 keys)
        char* kt_ = smem + buf * STAGE;
        int left = seq_bytes - t * (SUP * tile_bytes);
        left = (t < nstage && left > 0) ? left : 0;
        const bf16_t* kt_base = kseq + (long long)t * (SUP * tile_bytes / 2);
        const bf16_t* vt_base = vseq + (long long)t * (SUP * tile_bytes / 2);
#pragma unroll
        for (int i = 0; i < SUP * LPW; ++i) {
            const int slab = wave + i * NW;
            attn_dma16(kt_base, left, kt_ + slab * 1024, st_voff[i]);
            if (!KVSAME) attn_dma16(vt_base, left, kt_ + SUP * TILE + slab * 1024, st_voff[i]);
        }
    };

    if (ntile > 0) {
        stage(0, 0);
        stage(1, 1);
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LPT) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    // lane offsets of the fragment reads, valid for rows r, r + 16, r + 32, r + 48 alike (swizzle keys use row bits 0..3 only)
    uint32_t k_off[KS], v_off[DT][2];
    {
        const int row = lane & 31;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) k_off[ks] = (uint32_t)(row * ROWB + (attn_swz<CPR>(row, ks * 2 + h) << 4));
        const int g16 = lane >> 4, rr = (lane >> 2) & 3, qq = lane & 3;
        const int r0 = 4 * (g16 >> 1) + rr, r1 = r0 + 8;
        const uint32_t smem_off = attn_lds_offset(smem);
#pragma unroll
        for (int d = 0; d < DT; ++d) {
            const int col = d * 32 + 16 * (g16 & 1) + 4 * qq;
            v_off[d][0] = smem_off + (uint32_t)(r0 * ROWB + (attn_swz<CPR>(r0, col >> 3) << 4) + (col & 7) * 2);
            v_off[d][1] = smem_off + (uint32_t)(r1 * ROWB + (attn_swz<CPR>(r1, col >> 3) << 4) + (col & 7) * 2);
        }
    }

    auto tile_body = [&](auto slot_c, int t) {
        constexpr int SLOT = decltype(slot_c)::value;
        constexpr int KBASE = SLOT * STAGE, VBASE = KVSAME ? KBASE : KBASE + TILE;
        const int k0 = t * KT;
        stage(t + 2, SLOT == 0 ? 2 : SLOT - 1);

        // S^T = K Q^T: all K fragments of the tile requested up front (32 registers that are free in this phase), the two 32-key
        // accumulators alternate so that no MFMA waits on the one before it
        f32x16_t sacc[2];
        bf16x8_t kfr[2][KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) kfr[kt][ks] = *reinterpret_cast<const bf16x8_t*>(smem + KBASE + kt * 32 * ROWB + k_off[ks]);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[kt][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[kt][ks], qf[ks], sacc[kt], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        // V^T fragments of the first 32 keys: requested now, they land under the maximum / first-half exponentials
        bf16x4_t vlo[2][2][DT], vhi[2][2][DT];
        attn_static_for<0, 2>([&](auto s2c) {
            constexpr int s2 = decltype(s2c)::value;
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                vlo[0][s2][d] = attn_tr16i<VBASE + (16 * s2) * ROWB>(v_off[d][0]);
                vhi[0][s2][d] = attn_tr16i<VBASE + (16 * s2) * ROWB>(v_off[d][1]);
            }
        });
        // masks: only where the tile crosses the end of the keys or (causal) the diagonal of this wave's queries.  Key (kt, r) of lane
        // half h is kt * 32 + (r & 3) + 8 (r >> 2) + 4 h: one per-lane threshold, compile-time constants on the other side
        if ((k0 + KT > a.Sk) || (a.causal && k0 + KT - 1 > q0)) {
            int lim = a.Sk - k0;
            if (a.causal) lim = min(lim, qi - k0 + 1);
            lim -= 4 * h;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[kt][r] = (kt * 32 + (r & 3) + 8 * (r >> 2)) < lim ? sacc[kt][r] : -INFINITY;
        }
        float tmax;
        {
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
        tmax = attn_xmax32(tmax);
        const float m_new = fmaxf(m_run, tmax);
        // exact rule: any lane's maximum moved.  Deferred rule: some lane's moved by more than 8 / c2 (the first tile always does: m = -1e30)
        if (__builtin_amdgcn_ballot_w64(DEFER == 1 ? (m_new - m_run) * c2 > 8.f : m_new > m_run) != 0) {
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c2);
            l_run *= alpha;
#pragma unroll
            for (int d = 0; d < DT; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
            m_run = m_new;
        }
        const float mc = m_run * c2;
        // exp2, row sums on four independent chains, P packed to bf16 on the way out (the fp32 scores die pair by pair).  The first
        // half's four P V products are issued between the two halves' exponentials, so that they run in the matrix pipe while this
        // wave's own VALU works through the second half.  (Measured neutral against issuing all eight at the end -- 932 vs 936 TFLOP/s
        // at S = 4096 -- like the other schedule changes tried on this loop: the SQ counters put the VALU at 74 % and the matrix pipe
        // at 43 % of the kernel's cycles with only a quarter of the latter overlapped, profiles/r04_pmc_attn.txt.)
        uint32_t pw[2][8];
        float ps0 = 0.f, ps1 = 0.f, ps2 = 0.f, ps3 = 0.f;
        attn_static_for<0, 2>([&](auto ktc) {
            constexpr int kt = decltype(ktc)::value;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[kt][4 * e + 0], c2, -mc));
                const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[kt][4 * e + 1], c2, -mc));
                const float p2 = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[kt][4 * e + 2], c2, -mc));
                const float p3 = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[kt][4 * e + 3], c2, -mc));
                ps0 += p0; ps1 += p1; ps2 += p2; ps3 += p3;
                pw[kt][2 * e] = pack_bf16x2(p0, p1);
                pw[kt][2 * e + 1] = pack_bf16x2(p2, p3);
            }
            if constexpr (kt == 1) {
                float psum = (ps0 + ps1) + (ps2 + ps3);
                psum = attn_xsum32(psum);
                l_run += psum;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                u32x4_t pv4;
#pragma unroll
                for (int e = 0; e < 4; ++e) pv4[e] = pw[kt][4 * s2 + e];
                const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, pv4);
#pragma unroll
                for (int d = 0; d < DT; ++d) {
                    asm volatile("" : "+v"(vlo[kt][s2][d]), "+v"(vhi[kt][s2][d]));
                    const bf16x8_t vf = __builtin_shufflevector(vlo[kt][s2][d], vhi[kt][s2][d], 0, 1, 2, 3, 4, 5, 6, 7);
                    oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oacc[d], 0, 0, 0);
                }
            }
            __builtin_amdgcn_s_setprio(0);
            if constexpr (kt == 0) {
                // second half's V^T fragments: requested behind the first half's products, they land under the second half's exponentials
                __builtin_amdgcn_sched_barrier(0);
                attn_static_for<0, 2>([&](auto s2c) {
                    constexpr int s2 = decltype(s2c)::value;
#pragma unroll
                    for (int d = 0; d < DT; ++d) {
                        vlo[1][s2][d] = attn_tr16i<VBASE + (32 + 16 * s2) * ROWB>(v_off[d][0]);
                        vhi[1][s2][d] = attn_tr16i<VBASE + (32 + 16 * s2) * ROWB>(v_off[d][1]);
                    }
                });
                __builtin_amdgcn_sched_barrier(0);
            }
        });
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LPT) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    // DEFER == 2, "sum-triggered" running maximum (round 6).  The deferred rule above still pays the maximum of every tile (24 VALU instructions
    // of ~150 per tile and lane in a loop that is VALU-issue-bound: 9.9 VALU per MFMA) only to find, on all but the first tiles, that nothing
    // has to move.  Here a tile is exponentiated against the running maximum AS IT IS; the row sums, which are needed anyway, tell whether that
    // was legitimate: every p >= 0, so a lane's partial sum <= 2^8 bounds each of its p by 2^8 -- the deferred rule's bound.  Only when some
    // lane's sum exceeds it (or is inf / NaN: the first tile, m = -1e30) is the tile redone with its exact maximum: scores recomputed (the K
    // tile is still in its ring slot), maximum, rescale of O and l, exponentials again.  No P V product is issued before the test, so a redo
    // finds O untouched.  Steady state: no maximum tree, no compare chain -- 113 VALU instructions per tile instead of 150.
    auto tile_body_s = [&](auto slot_c, int st) {
        constexpr int SLOT = decltype(slot_c)::value;
        stage(st + 2, SLOT == 0 ? 2 : SLOT - 1);
        attn_static_for<0, SUP>([&](auto half_c) {
        constexpr int HALF = decltype(half_c)::value;
        constexpr int KBASE = SLOT * STAGE + HALF * TILE, VBASE = KVSAME ? KBASE : SLOT * STAGE + SUP * TILE + HALF * TILE;
        const int t = st * SUP + HALF;
        if (HALF > 0 && t >= ntile) return;                        // odd tile count: the stage's second tile does not exist (wave-uniform)
        const int k0 = t * KT;
        bool with_max = t == 0;                                    // wave-uniform
        uint32_t pw[2][8];
        bf16x4_t vlo[2][2][DT], vhi[2][2][DT];
        float psum;
        for (;;) {
            f32x16_t sacc[2];
            bf16x8_t kfr[2][KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) kfr[kt][ks] = *reinterpret_cast<const bf16x8_t*>(smem + KBASE + kt * 32 * ROWB + k_off[ks]);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[kt][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[kt][ks], qf[ks], sacc[kt], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            attn_static_for<0, 2>([&](auto s2c) {
                constexpr int s2 = decltype(s2c)::value;
#pragma unroll
                for (int d = 0; d < DT; ++d) {
                    vlo[0][s2][d] = attn_tr16i<VBASE + (16 * s2) * ROWB>(v_off[d][0]);
                    vhi[0][s2][d] = attn_tr16i<VBASE + (16 * s2) * ROWB>(v_off[d][1]);
                }
            });
            if ((k0 + KT > a.Sk) || (a.causal && k0 + KT - 1 > q0)) {
                int lim = a.Sk - k0;
                if (a.causal) lim = min(lim, qi - k0 + 1);
                lim -= 4 * h;
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sacc[kt][r] = (kt * 32 + (r & 3) + 8 * (r >> 2)) < lim ? sacc[kt][r] : -INFINITY;
            }
            if (with_max) {
                float tmax;
                {
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
                tmax = attn_xmax32(tmax);
                const float m_new = fmaxf(m_run, tmax);
                const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c2);
                l_run *= alpha;
#pragma unroll
                for (int d = 0; d < DT; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
                m_run = m_new;
            }
            const float mc = m_run * c2;
            float ps0, ps1, ps2, ps3;
            attn_static_for<0, 2>([&](auto ktc) {
                constexpr int kt = decltype(ktc)::value;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[kt][4 * e + 0], c2, -mc));
                    const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[kt][4 * e + 1], c2, -mc));
                    const float p2 = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[kt][4 * e + 2], c2, -mc));
                    const float p3 = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[kt][4 * e + 3], c2, -mc));
                    if (kt == 0 && e == 0) { ps0 = p0; ps1 = p1; ps2 = p2; ps3 = p3; }
                    else { ps0 += p0; ps1 += p1; ps2 += p2; ps3 += p3; }
                    pw[kt][2 * e] = pack_bf16x2(p0, p1);
                    pw[kt][2 * e + 1] = pack_bf16x2(p2, p3);
                }
            });
            psum = (ps0 + ps1) + (ps2 + ps3);
            // (!(x <= 2^8): true for inf and NaN as well)
            if (with_max || __builtin_amdgcn_ballot_w64(!(psum <= 256.f)) == 0) break;
            with_max = true;
        }
        l_run += attn_xsum32(psum);
        // second half's V^T fragments: requested in front of the first half's products, they land under them
        __builtin_amdgcn_sched_barrier(0);
        attn_static_for<0, 2>([&](auto s2c) {
            constexpr int s2 = decltype(s2c)::value;
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                vlo[1][s2][d] = attn_tr16i<VBASE + (32 + 16 * s2) * ROWB>(v_off[d][0]);
                vhi[1][s2][d] = attn_tr16i<VBASE + (32 + 16 * s2) * ROWB>(v_off[d][1]);
            }
        });
        attn_static_for<0, 2>([&](auto ktc) {
            constexpr int kt = decltype(ktc)::value;
            if constexpr (kt == 0) asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(4 * DT) : "memory");     // the first half's fragments (older requests) have landed
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                u32x4_t pv4;
#pragma unroll
                for (int e = 0; e < 4; ++e) pv4[e] = pw[kt][4 * s2 + e];
                const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, pv4);
#pragma unroll
                for (int d = 0; d < DT; ++d) {
                    asm volatile("" : "+v"(vlo[kt][s2][d]), "+v"(vhi[kt][s2][d]));
                    const bf16x8_t vf = __builtin_shufflevector(vlo[kt][s2][d], vhi[kt][s2][d], 0, 1, 2, 3, 4, 5, 6, 7);
                    oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oacc[d], 0, 0, 0);
                }
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
        });
        });                                                        // tiles of the stage
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LPT) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    auto tile_any = [&](auto slot_c, int t) {
        if constexpr (DEFER >= 2) tile_body_s(slot_c, t); else tile_body(slot_c, t);
    };
    for (int t = 0; t < nstage; t += 3) {                          // t: stage index (= tile index where a stage is one tile)
        tile_any(std::integral_constant<int, 0>{}, t);
        if (t + 1 < nstage) tile_any(std::integral_constant<int, 1>{}, t + 1);
        if (t + 2 < nstage) tile_any(std::integral_constant<int, 2>{}, t + 2);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    __syncthreads();
    {
        float* fl = reinterpret_cast<float*>(smem) + wave * 32 * DH;
        const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
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
        if (a.lse && h == 0 && qi < a.Sq) a.lse[((obase_s - head * DH + (long long)qi * a.om.pos_stride) / a.C) * a.nhead + head] = m_run * a.scale + __logf(l_run);
#pragma unroll
        for (int i = 0; i < CPR / 2; ++i) {
            const int idx = i * 64 + lane, row = idx / CPR, c = idx % CPR;
            if (q0 + row >= a.Sq) continue;
            float f[8];
            rows_get_f32<DH>(fl, row, c, f);
            const long long o = obase_s + (long long)(q0 + row) * a.om.pos_stride + c * 8;
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

// ------------------------------------------------------------------------------------------------------------------------------
// Forward with the WHOLE key / value sequence resident in LDS (round 6): self-attention (q == k == v, the reference's default,
// attention.py:219-223), non-causal, 64 < S <= 1024 -- spatial attention of the tokenizers and of the half-resolution LatentAction blocks.
// attn_fwd4_kernel streams K / V through a ring for every block of 128 queries: at S = 1024 that is eight blocks per (sequence, head), each
// DMA-ing the same 128 KB, each paying a prologue, an epilogue and a barrier + counted wait per key tile -- 4.4 tile-times of fixed cost against
// 16 tiles (profiles/r05_attention_vs_length.txt).  Here ONE workgroup of up to 16 waves owns a (sequence, head): the sequence is DMA'd once
// (128 KB at S = 1024; the CU's whole register file and 128 of its 160 KB of LDS), the waves then walk their 32-query tiles on their own --
// no barrier, no vmcnt, no DMA in the key loop; Q fragments come from the same LDS image (q == k); O leaves through registers: a
// v_permlane32_swap per value gives each lane 8 consecutive channels of its row, so the residual is read and the results are stored as
// 16-byte chunks, 32 contiguous bytes per row and instruction, without an LDS staging tile that would not fit.
// Arithmetic: that of attn_fwd4_kernel<.., DEFER = 2> (sum-triggered running maximum).
// MEASURED (profiles/r06_attention_resident_ab.log): parity-green (tests/test_gpu_attention.py::test_attention_sequence_resident_forward) and
// SLOWER than the ring kernel -- 762 vs 816 TFLOP/s at S = 1024 / C = 256, 739 vs 764 at C = 512, 456 vs 464 at S = 256: with 128 KB of LDS a CU holds
// ONE workgroup, so its 128-KB load (8-10 us of a ~95-us workgroup) is covered by nothing, and the ring kernel's per-tile barrier + counted wait --
// what this form removes -- turn out not to be what S = 1024 pays for (four ring workgroups per CU already overlap each other's prologues).
// Kept behind mode bit 8, OFF by default.
// ------------------------------------------------------------------------------------------------------------------------------
template <int DH>
__global__ void __attribute__((amdgpu_waves_per_eu(4, 4))) __launch_bounds__(1024) attn_fwdr_kernel(const AttnArgs a) {
    constexpr int KT = 64, ROWB = DH * 2, CPR = DH / 8, TILE = KT * ROWB, KS = DH / 16, DT = DH / 32;
    static_assert(DH == 64, "register-direct epilogue is written for two 32-channel accumulator tiles");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = (int)(blockDim.x >> 6);
    const int seq = blockIdx.x / a.nhead, head = blockIdx.x % a.nhead;
    const int h = lane >> 5;
    const int ntile = (a.Sk + KT - 1) / KT;

    {   // the whole sequence: 1-KiB pieces (8 key rows each), wave w takes pieces w, w + nw, ...; rows past the end read zeros (range check)
        const bf16_t* kseq = a.k + seq_base(a.km, seq) + head * DH;
        const int seq_bytes = (int)((a.Sk - 1) * a.km.pos_stride * 2) + DH * 2;
        const int nslab = ntile * (TILE / 1024);
        const int r8 = lane / CPR, ch = lane % CPR;
        for (int slab = wave; slab < nslab; slab += nw) {
            const int row = slab * (64 / CPR) + r8;
            const uint32_t voff = (uint32_t)((long long)row * a.km.pos_stride * 2) + (uint32_t)(attn_swz<CPR>(row, ch) * 16);
            attn_dma16(kseq, seq_bytes, smem + slab * 1024, voff);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    const float c2 = a.scale * 1.4426950408889634f;
    auto ldk = [&](uint32_t addr) -> bf16x8_t {              // ds_read_b128 (the caller waits)
        bf16x8_t v;
        asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
        return v;
    };

    for (int qt = wave; qt * 32 < a.Sq; qt += nw) {
        const int q0 = qt * 32, qi = q0 + (lane & 31);
        // lane offsets of the fragment reads for key tile 0 (rebuilt per query tile -- a dozen instructions -- instead of kept: 8 registers);
        // the key loop advances them by four tiles per group, the tile inside a group is an instruction immediate
        uint32_t kb[KS], vb[DT][2];
        {
            int ln = lane;
            asm volatile("" : "+v"(ln));                    // laundered: hipcc would hoist these lane constants out of the query loop and keep BOTH copies
            const int row = ln & 31, hh = ln >> 5;
            const uint32_t smem_off = attn_lds_offset(smem);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) kb[ks] = smem_off + (uint32_t)(row * ROWB + (attn_swz<CPR>(row, ks * 2 + hh) << 4));
            const int g16 = ln >> 4, rr = (ln >> 2) & 3, qq = ln & 3;
            const int r0 = 4 * (g16 >> 1) + rr, r1 = r0 + 8;
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                const int col = d * 32 + 16 * (g16 & 1) + 4 * qq;
                vb[d][0] = smem_off + (uint32_t)(r0 * ROWB + (attn_swz<CPR>(r0, col >> 3) << 4) + (col & 7) * 2);
                vb[d][1] = smem_off + (uint32_t)(r1 * ROWB + (attn_swz<CPR>(r1, col >> 3) << 4) + (col & 7) * 2);
            }
        }
        bf16x8_t qf[KS];
        {
            const uint32_t qb = (uint32_t)((q0 >> 6) * TILE + (q0 & 32) * ROWB);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) qf[ks] = ldk(kb[ks] + qb);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]));
        }
        f32x16_t oacc[DT];
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
        float m_run = -1e30f, l_run = 0.f;

        auto tile_body_r = [&](auto imm_c, int t) {
            constexpr int IMM = decltype(imm_c)::value * TILE;
            const int k0 = t * KT;
            bool with_max = t == 0;
            uint32_t pw[2][8];
            bf16x4_t vlo[2][2][DT], vhi[2][2][DT];
            float psum;
            for (;;) {
                // S^T = K Q^T.  The K fragments come in two batches of four (k-steps 0-1, then 2-3), the second requested behind the first batch's
                // products: 16 registers of fragments in flight instead of 32 -- with all eight the kernel spilled two Q fragments to scratch, and a
                // scratch reload is a vmcnt(0) in the middle of the MFMA cluster.  The second batch's LDS latency is covered by the other three
                // waves of the SIMD.
                f32x16_t sacc[2];
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sacc[kt][r] = 0.f;
                attn_static_for<0, 2>([&](auto bc) {
                    constexpr int b2 = decltype(bc)::value;
                    bf16x8_t kf[2][2];
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int kt = 0; kt < 2; ++kt)
                            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kf[kt][j]) : "v"(kb[2 * b2 + j]), "i"(IMM + kt * 32 * ROWB));
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int kt = 0; kt < 2; ++kt) asm volatile("" : "+v"(kf[kt][j]));
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int kt = 0; kt < 2; ++kt) sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kt][j], qf[2 * b2 + j], sacc[kt], 0, 0, 0);
                    __builtin_amdgcn_s_setprio(0);
                    __builtin_amdgcn_sched_barrier(0);
                });
                attn_static_for<0, 2>([&](auto s2c) {
                    constexpr int s2 = decltype(s2c)::value;
#pragma unroll
                    for (int d = 0; d < DT; ++d) {
                        vlo[0][s2][d] = attn_tr16i<IMM + (16 * s2) * ROWB>(vb[d][0]);
                        vhi[0][s2][d] = attn_tr16i<IMM + (16 * s2) * ROWB>(vb[d][1]);
                    }
                });
                if (k0 + KT > a.Sk) {
                    const int lim = a.Sk - k0 - 4 * h;
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) sacc[kt][r] = (kt * 32 + (r & 3) + 8 * (r >> 2)) < lim ? sacc[kt][r] : -INFINITY;
                }
                if (with_max) {
                    float tmax;
                    {
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
                    tmax = attn_xmax32(tmax);
                    const float m_new = fmaxf(m_run, tmax);
                    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c2);
                    l_run *= alpha;
#pragma unroll
                    for (int d = 0; d < DT; ++d)
#pragma unroll
                        for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
                    m_run = m_new;
                }
                const float mc = m_run * c2;
                float ps0, ps1, ps2, ps3;
                attn_static_for<0, 2>([&](auto ktc) {
                    constexpr int kt = decltype(ktc)::value;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[kt][4 * e + 0], c2, -mc));
                        const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[kt][4 * e + 1], c2, -mc));
                        const float p2 = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[kt][4 * e + 2], c2, -mc));
                        const float p3 = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[kt][4 * e + 3], c2, -mc));
                        if (kt == 0 && e == 0) { ps0 = p0; ps1 = p1; ps2 = p2; ps3 = p3; }
                        else { ps0 += p0; ps1 += p1; ps2 += p2; ps3 += p3; }
                        pw[kt][2 * e] = pack_bf16x2(p0, p1);
                        pw[kt][2 * e + 1] = pack_bf16x2(p2, p3);
                    }
                });
                psum = (ps0 + ps1) + (ps2 + ps3);
                if (with_max || __builtin_amdgcn_ballot_w64(!(psum <= 256.f)) == 0) break;
                with_max = true;
            }
            l_run += attn_xsum32(psum);
            __builtin_amdgcn_sched_barrier(0);
            attn_static_for<0, 2>([&](auto s2c) {
                constexpr int s2 = decltype(s2c)::value;
#pragma unroll
                for (int d = 0; d < DT; ++d) {
                    vlo[1][s2][d] = attn_tr16i<IMM + (32 + 16 * s2) * ROWB>(vb[d][0]);
                    vhi[1][s2][d] = attn_tr16i<IMM + (32 + 16 * s2) * ROWB>(vb[d][1]);
                }
            });
            attn_static_for<0, 2>([&](auto ktc) {
                constexpr int kt = decltype(ktc)::value;
                if constexpr (kt == 0) asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(4 * DT) : "memory");
                else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    u32x4_t pv4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) pv4[e] = pw[kt][4 * s2 + e];
                    const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, pv4);
#pragma unroll
                    for (int d = 0; d < DT; ++d) {
                        asm volatile("" : "+v"(vlo[kt][s2][d]), "+v"(vhi[kt][s2][d]));
                        const bf16x8_t vf = __builtin_shufflevector(vlo[kt][s2][d], vhi[kt][s2][d], 0, 1, 2, 3, 4, 5, 6, 7);
                        oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oacc[d], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        // key tiles in groups of four: the group's base goes into the lane offsets once, the tile inside the group is an instruction immediate
        for (int t0 = 0; t0 < ntile; t0 += 4) {
            tile_body_r(std::integral_constant<int, 0>{}, t0);
            if (t0 + 1 < ntile) tile_body_r(std::integral_constant<int, 1>{}, t0 + 1);
            if (t0 + 2 < ntile) tile_body_r(std::integral_constant<int, 2>{}, t0 + 2);
            if (t0 + 3 < ntile) tile_body_r(std::integral_constant<int, 3>{}, t0 + 3);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) kb[ks] += 4 * TILE;
#pragma unroll
            for (int d = 0; d < DT; ++d) { vb[d][0] += 4 * TILE; vb[d][1] += 4 * TILE; }
        }
        // epilogue, through registers: accumulator (d, 4 g + e) of lane (query, h) is channel 32 d + 8 g + 4 h + e.  For a pair of groups (g, g + 1)
        // one v_permlane32_swap per value hands lane h = 0 the whole group g (its own four channels and the partner's) and lane h = 1 the
        // whole group g + 1: eight consecutive channels = one 16-byte chunk of the row
        const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
        const long long obase_s = seq_base(a.om, seq) + head * DH;
        if (a.lse && h == 0 && qi < a.Sq) a.lse[((obase_s - head * DH + (long long)qi * a.om.pos_stride) / a.C) * a.nhead + head] = m_run * a.scale + __logf(l_run);
        const long long orow = obase_s + (long long)qi * a.om.pos_stride;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                float x[4], y[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) { x[e] = oacc[d][8 * gp + e] * inv; y[e] = oacc[d][8 * gp + 4 + e] * inv; }
#pragma unroll
                for (int e = 0; e < 4; ++e) attn_swap32(x[e], y[e]);
                // lanes < 32: (x, y) = group 2 gp, channels 0-3 | 4-7; lanes >= 32: group 2 gp + 1
                float f[8] = {x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
                if (qi < a.Sq) {
                    const long long o = orow + d * 32 + (2 * gp + h) * 8;
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
}

// ------------------------------------------------------------------------------------------------------------------------------
// backward, dQ: per query tile, loop over key tiles (lane = query).  Per 32-key half: S^T and dP^T (8 MFMAs on two independent
// accumulators), p = exp2(c s - lse), dS = p (dP - D) packed to bf16, dQ^T += K^T dS^T (4 MFMAs).  Keys past the end need no mask
// here: their K rows are zeros (descriptor range check), so whatever p they get multiplies a zero row; only the causal diagonal does.
// ------------------------------------------------------------------------------------------------------------------------------
template <int DH, int NW, bool KVSAME>
__global__ void __attribute__((amdgpu_waves_per_eu(3, 3))) __launch_bounds__(64 * NW) attn_bwd_dq3_kernel(const AttnBwdArgs a) {
    constexpr int KT = 64, ROWB = DH * 2, CPR = DH / 8, TILE = KT * ROWB, KS = DH / 16, DT = DH / 32;
    constexpr int SLABS = TILE / 1024, LPW = SLABS / NW, LPT = KVSAME ? LPW : 2 * LPW;
    constexpr int STAGE = KVSAME ? TILE : 2 * TILE;
    static_assert(SLABS % NW == 0, "every wave stages the same number of pieces (counted vmcnt)");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qtiles = (a.Sq + 32 * NW - 1) / (32 * NW);
    const AttnBlock blk = attn_block_of(a.xcd_swizzle, a.nseq, a.nhead, qtiles);
    if (!blk.live) return;
    const int seq = blk.seq, qtile = blk.tile, head = blk.head;
    const int q0 = qtile * (32 * NW) + wave * 32;
    const int qi = q0 + (lane & 31), h = lane >> 5;
    bf16x8_t qf[KS], dof[KS];
    float lse2 = 0.f, nD_q = 0.f;                 // lse * log2(e); -D
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
            lse2 = a.lse2[tok * a.nhead + head];
            nD_q = -a.D[tok * a.nhead + head];
        }
    }
    f32x16_t dq[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[d][r] = 0.f;
    // dS = p (dP - D): the subtraction rides in the dP^T product as its accumulator INPUT -- a lane's 16 accumulator rows all belong to
    // its one query, so a block of 16 registers holding -D feeds the first MFMA of every half (16 registers for 32 v_sub_f32 per tile)
    f32x16_t negD;
#pragma unroll
    for (int r = 0; r < 16; ++r) negD[r] = nD_q;
    const float c2 = a.scale * 1.4426950408889634f;
    const int blk_q_max = qtile * (32 * NW) + 32 * NW - 1;
    int k_end = a.Sk;
    if (a.causal && blk_q_max + 1 < k_end) k_end = blk_q_max + 1;
    const int ntile = (k_end + KT - 1) / KT;

    const bf16_t* kseq = a.k + seq_base(a.km, seq) + head * DH;
    const bf16_t* vseq = a.v + seq_base(a.km, seq) + head * DH;
    const int tile_bytes = (int)(KT * a.km.pos_stride * 2);
    const int seq_bytes = (int)((a.Sk - 1) * a.km.pos_stride * 2) + DH * 2;
    uint32_t st_voff[LPW];
#pragma unroll
    for (int i = 0; i < LPW; ++i) {
        const int idx = (wave + i * NW) * 64 + lane;
        const int row = idx / CPR;
        st_voff[i] = (uint32_t)((long long)row * a.km.pos_stride * 2) + (uint32_t)(attn_swz<CPR>(row, idx % CPR) * 16);
    }
    auto stage = [&](int t, int buf) {
        char* kt_ = smem + buf * STAGE;
        int left = seq_bytes - t * tile_bytes;
        left = (t < ntile && left > 0) ? left : 0;
        const bf16_t* kt_base = kseq + (long long)t * (tile_bytes / 2);
        const bf16_t* vt_base = vseq + (long long)t * (tile_bytes / 2);
#pragma unroll
        for (int i = 0; i < LPW; ++i) {
            const int slab = wave + i * NW;
            attn_dma16(kt_base, left, kt_ + slab * 1024, st_voff[i]);
            if (!KVSAME) attn_dma16(vt_base, left, kt_ + TILE + slab * 1024, st_voff[i]);
        }
    };
    if (ntile > 0) {
        stage(0, 0);
        stage(1, 1);
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LPT) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    uint32_t k_off[KS], t_off[DT][2];
    {
        const int row = lane & 31;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) k_off[ks] = (uint32_t)(row * ROWB + (attn_swz<CPR>(row, ks * 2 + h) << 4));
        const int g16 = lane >> 4, rr = (lane >> 2) & 3, qq = lane & 3;
        const int r0 = 4 * (g16 >> 1) + rr, r1 = r0 + 8;
        const uint32_t smem_off = attn_lds_offset(smem);
#pragma unroll
        for (int d = 0; d < DT; ++d) {
            const int col = d * 32 + 16 * (g16 & 1) + 4 * qq;
            t_off[d][0] = smem_off + (uint32_t)(r0 * ROWB + (attn_swz<CPR>(r0, col >> 3) << 4) + (col & 7) * 2);
            t_off[d][1] = smem_off + (uint32_t)(r1 * ROWB + (attn_swz<CPR>(r1, col >> 3) << 4) + (col & 7) * 2);
        }
    }
    auto tile_body = [&](auto slot_c, int t) {
        constexpr int SLOT = decltype(slot_c)::value;
        constexpr int KBASE = SLOT * STAGE, VBASE = KVSAME ? KBASE : KBASE + TILE;
        const int k0 = t * KT;
        stage(t + 2, SLOT == 0 ? 2 : SLOT - 1);
        const bool diag = a.causal && k0 + KT - 1 > q0;
        const int lim = qi - k0 + 1 - 4 * h;             // causal: local keys below lim are visible to this lane's query
        attn_static_for<0, 2>([&](auto ktc) {
            constexpr int kt = decltype(ktc)::value;
            bf16x8_t kfr[KS], vfr[KVSAME ? 1 : KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                kfr[ks] = *reinterpret_cast<const bf16x8_t*>(smem + KBASE + kt * 32 * ROWB + k_off[ks]);
                if (!KVSAME) vfr[ks] = *reinterpret_cast<const bf16x8_t*>(smem + VBASE + kt * 32 * ROWB + k_off[ks]);
            }
            __builtin_amdgcn_sched_barrier(0);
            f32x16_t sacc, pacc = negD;
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[ks], qf[ks], sacc, 0, 0, 0);                                   // S^T
                pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(KVSAME ? kfr[ks] : vfr[KVSAME ? 0 : ks], dof[ks], pacc, 0, 0, 0);  // dP^T - D
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            // K^T fragments of this half for dQ^T += K^T dS^T: requested now, they land under the element-wise phase
            bf16x4_t klo[2][DT], khi[2][DT];
            attn_static_for<0, 2>([&](auto s2c) {
                constexpr int s2 = decltype(s2c)::value;
#pragma unroll
                for (int d = 0; d < DT; ++d) {
                    klo[s2][d] = attn_tr16i<KBASE + (kt * 32 + 16 * s2) * ROWB>(t_off[d][0]);
                    khi[s2][d] = attn_tr16i<KBASE + (kt * 32 + 16 * s2) * ROWB>(t_off[d][1]);
                }
            });
            if (diag) {                                   // (a block of its own: hipcc turns a per-element `if` into selects on EVERY tile)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[r] = (kt * 32 + (r & 3) + 8 * (r >> 2)) < lim ? sacc[r] : -INFINITY;     // exp2(-inf) = 0
            }
            uint32_t dw[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[2 * e], c2, -lse2));
                const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[2 * e + 1], c2, -lse2));
                dw[e] = pack_bf16x2(p0 * pacc[2 * e], p1 * pacc[2 * e + 1]);                                     // dS^T / scale
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                u32x4_t d4;
#pragma unroll
                for (int e = 0; e < 4; ++e) d4[e] = dw[4 * s2 + e];
                const bf16x8_t df = __builtin_bit_cast(bf16x8_t, d4);
#pragma unroll
                for (int d = 0; d < DT; ++d) {
                    asm volatile("" : "+v"(klo[s2][d]), "+v"(khi[s2][d]));
                    const bf16x8_t kT = __builtin_shufflevector(klo[s2][d], khi[s2][d], 0, 1, 2, 3, 4, 5, 6, 7);
                    dq[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kT, df, dq[d], 0, 0, 0);                                     // dQ^T += K^T dS^T
                }
            }
            __builtin_amdgcn_s_setprio(0);
        });
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
                *reinterpret_cast<u32x2_t*>(wl + lr * ROWB + (attn_swz<CPR>(lr, d * 4 + g) << 4) + 8 * h) = ov;
            }
        const long long obase_s = seq_base(a.qm, seq) + head * DH;
#pragma unroll
        for (int i = 0; i < CPR / 2; ++i) {
            const int idx = i * 64 + lane, row = idx / CPR;
            if (q0 + row >= a.Sq) continue;
            *reinterpret_cast<u32x4_t*>(a.dq + obase_s + (long long)(q0 + row) * a.qm.pos_stride + attn_swz<CPR>(row, idx % CPR) * 8) =
                *reinterpret_cast<const u32x4_t*>(wl + idx * 16);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// backward, dK / dV: per key tile, loop over query tiles (lane = key).  The LDS stage holds a 64-query Q tile, its dO tile and the
// rows' lse / D.  Per 32-query half: S and dP (8 MFMAs), p = exp2(c s - lse), dS = p (dP - D), then dV^T += dO^T P and
// dK^T += Q^T dS -- one after the other, so only ONE set of transposed fragments (16 registers) is live at a time.  Self-attention
// (K == V) keeps a single copy of the lane's key row.  Queries past the end are zero rows (descriptor range check) with lse = D = 0:
// p = 1 multiplies dO = 0 / dP - D = 0, so only the causal diagonal needs a mask.
// ------------------------------------------------------------------------------------------------------------------------------
template <int DH, int NW, bool KVSAME>
__global__ void __attribute__((amdgpu_waves_per_eu(3, 3))) __launch_bounds__(64 * NW) attn_bwd_dkv3_kernel(const AttnBwdArgs a) {
    constexpr int QT = 64, ROWB = DH * 2, CPR = DH / 8, TILE = QT * ROWB, KS = DH / 16, DT = DH / 32;
    constexpr int SLABS = TILE / 1024, LPW = SLABS / NW, LPT = 2 * LPW + 2;
    constexpr int STAGE = 2 * TILE + 512;            // Q tile | dO tile | lse * log2e [64] | -D [64]
    static_assert(SLABS % NW == 0, "every wave stages the same number of pieces (counted vmcnt)");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ktiles = (a.Sk + 32 * NW - 1) / (32 * NW);
    const AttnBlock blk = attn_block_of(a.xcd_swizzle, a.nseq, a.nhead, ktiles);
    if (!blk.live) return;
    const int seq = blk.seq, ktile_i = blk.tile, head = blk.head;
    const int key0 = ktile_i * (32 * NW) + wave * 32;
    const int ki = key0 + (lane & 31), h = lane >> 5;
    bf16x8_t kf[KS], vf[KVSAME ? 1 : KS];
    {
        const bool ok = ki < a.Sk;
        const long long off = seq_base(a.km, seq) + (long long)(ok ? ki : 0) * a.km.pos_stride + head * DH;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ok) {
                kf[ks] = *reinterpret_cast<const bf16x8_t*>(a.k + off + ks * 16 + h * 8);
                if (!KVSAME) vf[ks] = *reinterpret_cast<const bf16x8_t*>(a.v + off + ks * 16 + h * 8);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    kf[ks][e] = 0;
                    if (!KVSAME) vf[ks][e] = 0;
                }
            }
        }
    }
    f32x16_t dk[DT], dv[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[d][r] = 0.f; dv[d][r] = 0.f; }
    const float c2 = a.scale * 1.4426950408889634f;
    const int blk_key_min = ktile_i * (32 * NW);
    const int q_begin = a.causal ? (blk_key_min / QT) * QT : 0;      // queries before the first key of the block see none of its keys
    const int ntile = a.Sq > q_begin ? (a.Sq - q_begin + QT - 1) / QT : 0;

    // staging: Q and dO tiles by buffer-addressed LDS-DMA (rows past Sq read zeros); lse / D of the tile's 64 queries by a 4-B DMA per lane
    const bf16_t* qseq = a.q + seq_base(a.qm, seq) + head * DH + (long long)q_begin * a.qm.pos_stride;
    const bf16_t* oseq = a.dO + seq_base(a.om, seq) + head * DH + (long long)q_begin * a.om.pos_stride;
    const int q_tile_bytes = (int)(QT * a.qm.pos_stride * 2), o_tile_bytes = (int)(QT * a.om.pos_stride * 2);
    const int q_seq_bytes = (int)((a.Sq - q_begin - 1) * a.qm.pos_stride * 2) + DH * 2;
    const int o_seq_bytes = (int)((a.Sq - q_begin - 1) * a.om.pos_stride * 2) + DH * 2;
    // lse / D rows: token = element offset / C.  The host sends only maps whose strides are multiples of C here, so the division
    // happens once per block and a tile's rows are tok_seq + qrow * tok_step (the general kernel divides 64-bit per lane per tile)
    const long long tok_seq = seq_base(a.om, seq) / a.C;
    const int tok_step = (int)(a.om.pos_stride / a.C);
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page_l);
    uint32_t st_qoff[LPW], st_ooff[LPW];
#pragma unroll
    for (int i = 0; i < LPW; ++i) {
        const int idx = (wave + i * NW) * 64 + lane;
        const int row = idx / CPR;
        const uint32_t cb = (uint32_t)(attn_swz<CPR>(row, idx % CPR) * 16);
        st_qoff[i] = (uint32_t)((long long)row * a.qm.pos_stride * 2) + cb;
        st_ooff[i] = (uint32_t)((long long)row * a.om.pos_stride * 2) + cb;
    }
    auto stage = [&](int t, int buf) {
        char* qt_ = smem + buf * STAGE;
        int lq = q_seq_bytes - t * q_tile_bytes, lo = o_seq_bytes - t * o_tile_bytes;
        lq = (t < ntile && lq > 0) ? lq : 0;
        lo = (t < ntile && lo > 0) ? lo : 0;
        const bf16_t* qt_base = qseq + (long long)t * (q_tile_bytes / 2);
        const bf16_t* ot_base = oseq + (long long)t * (o_tile_bytes / 2);
#pragma unroll
        for (int i = 0; i < LPW; ++i) {
            const int slab = wave + i * NW;
            attn_dma16(qt_base, lq, qt_ + slab * 1024, st_qoff[i]);
            attn_dma16(ot_base, lo, qt_ + TILE + slab * 1024, st_ooff[i]);
        }
        {   // lse / D: 4 B per lane, every wave writes the same 256 B (same data)
            const int qrow = q_begin + t * QT + lane;
            const float* pl = reinterpret_cast<const float*>(zero);
            const float* pd = reinterpret_cast<const float*>(zero);
            if (t < ntile && qrow < a.Sq) {
                const long long tok = tok_seq + (long long)qrow * tok_step;
                pl = a.lse2 + tok * a.nhead + head;
                pd = a.negD + tok * a.nhead + head;
            }
            attn_dma4(pl, qt_ + 2 * TILE);
            attn_dma4(pd, qt_ + 2 * TILE + 256);
        }
    };
    if (ntile > 0) {
        stage(0, 0);
        stage(1, 1);
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LPT) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    uint32_t r_off[KS], t_off[DT][2];
    {
        const int row = lane & 31;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) r_off[ks] = (uint32_t)(row * ROWB + (attn_swz<CPR>(row, ks * 2 + h) << 4));
        const int g16 = lane >> 4, rr = (lane >> 2) & 3, qq = lane & 3;
        const int r0 = 4 * (g16 >> 1) + rr, r1 = r0 + 8;
        const uint32_t smem_off = attn_lds_offset(smem);
#pragma unroll
        for (int d = 0; d < DT; ++d) {
            const int col = d * 32 + 16 * (g16 & 1) + 4 * qq;
            t_off[d][0] = smem_off + (uint32_t)(r0 * ROWB + (attn_swz<CPR>(r0, col >> 3) << 4) + (col & 7) * 2);
            t_off[d][1] = smem_off + (uint32_t)(r1 * ROWB + (attn_swz<CPR>(r1, col >> 3) << 4) + (col & 7) * 2);
        }
    }
    auto tile_body = [&](auto slot_c, int t) {
        constexpr int SLOT = decltype(slot_c)::value;
        constexpr int QBASE = SLOT * STAGE, OBASE = QBASE + TILE;
        const int qs = q_begin + t * QT;
        stage(t + 2, SLOT == 0 ? 2 : SLOT - 1);
        const float* lse_l = reinterpret_cast<const float*>(smem + QBASE + 2 * TILE);
        const float* D_l = lse_l + 64;
        const bool diag = a.causal && key0 + 31 > qs;
        const int lo_q = ki - qs - 4 * h;                    // causal: local query rows >= lo_q see this lane's key
        attn_static_for<0, 2>([&](auto qtc) {
            constexpr int qt = decltype(qtc)::value;
            // A fragments two k-steps ahead of the MFMAs that use them (at most 3 x 8 registers in flight: this phase is the kernel's
            // register peak -- 80 persistent + 32 accumulator + fragments)
            bf16x8_t qfr[KS], dofr[KS];
            auto load_a = [&](int ks) {
                qfr[ks] = *reinterpret_cast<const bf16x8_t*>(smem + QBASE + qt * 32 * ROWB + r_off[ks]);
                dofr[ks] = *reinterpret_cast<const bf16x8_t*>(smem + OBASE + qt * 32 * ROWB + r_off[ks]);
            };
            load_a(0);
            if (KS > 1) load_a(1);
            __builtin_amdgcn_sched_barrier(0);
            // dS = p (dP - D): -D of the half's 32 queries (staged next to the tile by the DMA) is the accumulator INPUT of the dP product --
            // accumulator row r of a lane is query 8 (r >> 2) + 4 h + (r & 3): four 16-B LDS reads instead of 32 v_sub_f32
            f32x16_t sacc, pacc;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4_t d4 = *reinterpret_cast<const f32x4_t*>(D_l + qt * 32 + 8 * g + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) pacc[4 * g + e] = d4[e];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qfr[ks], kf[ks], sacc, 0, 0, 0);                          // S[q][key]
                pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dofr[ks], KVSAME ? kf[ks] : vf[KVSAME ? 0 : ks], pacc, 0, 0, 0);   // dP[q][key] - D[q]
                if (ks + 2 < KS) load_a(ks + 2);
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            // dO^T fragments of this half for dV^T += dO^T P: requested now, they land under the element-wise phase
            bf16x4_t tlo[2][DT], thi[2][DT];
            attn_static_for<0, 2>([&](auto s2c) {
                constexpr int s2 = decltype(s2c)::value;
#pragma unroll
                for (int d = 0; d < DT; ++d) {
                    tlo[s2][d] = attn_tr16i<OBASE + (qt * 32 + 16 * s2) * ROWB>(t_off[d][0]);
                    thi[s2][d] = attn_tr16i<OBASE + (qt * 32 + 16 * s2) * ROWB>(t_off[d][1]);
                }
            });
            if (diag) {                                   // (a block of its own: hipcc turns a per-element `if` into selects on EVERY tile)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[r] = (qt * 32 + 8 * (r >> 2) + (r & 3)) >= lo_q ? sacc[r] : -INFINITY;   // exp2(-inf) = 0
            }
            uint32_t pw[8], dw[8];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ql0 = qt * 32 + 8 * g + 4 * h;                                               // rows 4 g .. 4 g + 3 of this lane
                const f32x4_t l4 = *reinterpret_cast<const f32x4_t*>(lse_l + ql0);
                float p[4], ds[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g + e;
                    p[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[r], c2, -l4[e]));
                    ds[e] = p[e] * pacc[r];                                                             // dS / scale
                }
                pw[2 * g] = pack_bf16x2(p[0], p[1]); pw[2 * g + 1] = pack_bf16x2(p[2], p[3]);
                dw[2 * g] = pack_bf16x2(ds[0], ds[1]); dw[2 * g + 1] = pack_bf16x2(ds[2], ds[3]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                u32x4_t p4;
#pragma unroll
                for (int e = 0; e < 4; ++e) p4[e] = pw[4 * s2 + e];
                const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, p4);
#pragma unroll
                for (int d = 0; d < DT; ++d) {
                    asm volatile("" : "+v"(tlo[s2][d]), "+v"(thi[s2][d]));
                    const bf16x8_t doT = __builtin_shufflevector(tlo[s2][d], thi[s2][d], 0, 1, 2, 3, 4, 5, 6, 7);
                    dv[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(doT, pf, dv[d], 0, 0, 0);           // dV^T += dO^T P
                }
            }
            __builtin_amdgcn_s_setprio(0);
            // Q^T fragments into the registers the dO^T fragments just left; they land while the matrix pipe drains the four products above
            __builtin_amdgcn_sched_barrier(0);
            attn_static_for<0, 2>([&](auto s2c) {
                constexpr int s2 = decltype(s2c)::value;
#pragma unroll
                for (int d = 0; d < DT; ++d) {
                    tlo[s2][d] = attn_tr16i<QBASE + (qt * 32 + 16 * s2) * ROWB>(t_off[d][0]);
                    thi[s2][d] = attn_tr16i<QBASE + (qt * 32 + 16 * s2) * ROWB>(t_off[d][1]);
                }
            });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                u32x4_t d4;
#pragma unroll
                for (int e = 0; e < 4; ++e) d4[e] = dw[4 * s2 + e];
                const bf16x8_t df = __builtin_bit_cast(bf16x8_t, d4);
#pragma unroll
                for (int d = 0; d < DT; ++d) {
                    asm volatile("" : "+v"(tlo[s2][d]), "+v"(thi[s2][d]));
                    const bf16x8_t qT = __builtin_shufflevector(tlo[s2][d], thi[s2][d], 0, 1, 2, 3, 4, 5, 6, 7);
                    dk[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qT, df, dk[d], 0, 0, 0);            // dK^T += Q^T dS
                }
            }
            __builtin_amdgcn_s_setprio(0);
        });
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

// ------------------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------------------
// GENIE_ATTN_LEAN / genie_attention_lean_mode: bit 0 forward, bit 1 backward dQ, bit 2 backward dK / dV on the lean kernels (0 sends
// everything through attention.hip's general kernels -- A/B timing and the tests that compare the two families); bit 3: unused (was:
// no s_setprio around the MFMA clusters -- measured neutral, profiles/r04_attention_lean_ab.log); bit 6: forward blocks of four waves at every
// length (default: eight from 2048 queries on); bit 4: deferred running maximum in the forward (see attn_fwd4_kernel); bit 7 (with bit 4):
// the sum-triggered form of it (no per-tile maximum at all in the steady state); bit 8: self-attention sequences of 65..1024 positions run with the
// whole K / V sequence resident in LDS (attn_fwdr_kernel)
#define LEAN_DEFAULT 151     // lean forward + dQ + dK-dV, deferred running maximum in its sum-triggered form (bits 0, 1, 2, 4, 7); bit 8 (sequence-resident
                             // forward) is OFF: measured 5-7 % slower than the ring kernel, see attn_fwdr_kernel
static int g_lean_mode = -1;
static int lean_mode() {
    if (g_lean_mode < 0) { const char* e = getenv("GENIE_ATTN_LEAN"); g_lean_mode = e ? atoi(e) : LEAN_DEFAULT; }
    return g_lean_mode;
}
extern "C" int genie_attention_lean_mode(int mask) {
    const int old = lean_mode();
    if (mask >= 0) g_lean_mode = mask & 1023;
    return old;
}
// byte offsets inside a sequence travel as 32-bit buffer offsets / record counts
static bool fits32(long long rows, long long pos_stride) { return pos_stride > 0 && (rows + 192) * pos_stride * 2 < (1ll << 31); }

bool genie_attn_lean_fwd_ok(const AttnArgs& a, int d_head) {
    // blocks of four waves only (Sq > 64): the one- and two-wave forms stage more pieces per wave and do not fit 128 registers --
    // short sequences are traffic-bound and stay on the general kernel
    return d_head == 64 && (lean_mode() & 1) && a.Sq > 64 && fits32(a.Sk, a.km.pos_stride);
}
int genie_attn_lean_bwd_mask(const AttnBwdArgs& a, int d_head) {
    if (d_head != 64) return 0;
    int m = 0;
    if ((lean_mode() & 2) && fits32(a.Sk, a.km.pos_stride)) m |= 1;
    const bool tok_affine = a.C > 0 && a.om.pos_stride % a.C == 0 && a.om.stride_outer % a.C == 0 && a.om.stride_inner % a.C == 0;
    // dK / dV: self-attention layout (K == V: one copy of the lane's key row) and blocks of two or four waves fit 168 registers
    if ((lean_mode() & 4) && a.kv_same && a.Sk > 32 && tok_affine && fits32(a.Sq, a.qm.pos_stride) && fits32(a.Sq, a.om.pos_stride)) m |= 2;
    return m;
}

// grid of a lean launch: XCD-swizzled 1-D (default) or the general kernels' (sequence * tiles, head)
static dim3 lean_grid(int nseq, int nhead, int tiles, int* swizzle) {
    *swizzle = (lean_mode() & 32) ? 0 : 1;
    if (!*swizzle) return dim3((unsigned)(nseq * tiles), (unsigned)nhead, 1);
    const long long n = (long long)nseq * nhead * tiles;
    return dim3((unsigned)(((n + 7) / 8) * 8), 1, 1);
}

// sequence-resident forward (attn_fwdr_kernel): self-attention through ONE tensor, non-causal, 64 < S <= 1024; mode bit 8 (default on)
static bool lean_resident_ok(const AttnArgs& a) {
    return (lean_mode() & 256) && a.kv_same && a.q == a.k && !a.causal && a.Sq == a.Sk && a.Sk > 64 && a.Sk <= 1024 &&
           a.qm.pos_stride == a.km.pos_stride && a.qm.stride_outer == a.km.stride_outer && a.qm.stride_inner == a.km.stride_inner && a.qm.n_inner == a.km.n_inner;
}

int genie_attn_lean_fwd(const AttnArgs& a_in, hipStream_t s) {
    AttnArgs a = a_in;
    if (lean_resident_ok(a)) {
        const int ntile = (a.Sk + 63) / 64;
        const int lds = ntile * 64 * 64 * 2;
        int nw = (a.Sq + 31) / 32;
        if (nw > 16) nw = 16;
        GENIE_CHECK_ARG((long long)a.nseq * a.nhead < (1ll << 31), "genie_attention_fwd: grid too large");
        static bool configured = false;
        if (!configured) {
            GENIE_CHECK_ARG(hipFuncSetAttribute((const void*)attn_fwdr_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) == hipSuccess,
                            "hipFuncSetAttribute failed");
            configured = true;
        }
        attn_fwdr_kernel<64><<<(unsigned)(a.nseq * a.nhead), 64 * nw, lds, s>>>(a);
        GENIE_CHECK_LAUNCH();
        return GENIE_OK;
    }
    // blocks of four waves (128 queries); from 2048 queries on, eight (256 queries share a K / V tile: half the L2 -> LDS bytes per query; two
    // blocks of eight waves per CU are the same four waves per SIMD).  Same arithmetic, bit-identical results; measured 970 vs 951 TFLOP/s
    // at S = 4096, equal at 1024, 494 vs 520 at 256 (profiles/r04_attention_lean_ab.log).  Bit 6 of the mode keeps four everywhere (A/B).
    static const int nw8_min = [] { const char* e = getenv("GENIE_ATTN_NW8_MIN"); return e ? atoi(e) : 2048; }();      // (A/B knob)
    const int nw = (!(lean_mode() & 64) && a.Sq >= nw8_min) ? 8 : 4;
    const int qtiles = (a.Sq + 32 * nw - 1) / (32 * nw);
    GENIE_CHECK_ARG((long long)a.nseq * qtiles * a.nhead < (1ll << 31) - 8 && a.nhead <= 65535, "genie_attention_fwd: grid too large");
    const int tile = 64 * 64 * 2;
    // running-maximum rule: bit 4 deferred (round 4), bit 7 on top of it sum-triggered (round 6, the default); neither: exact
    const int defer = (lean_mode() & 16) ? ((lean_mode() & 128) ? ((lean_mode() & 512) ? 3 : 2) : 1) : 0;      // (bit 9: sum-triggered, one tile per stage)
    const int sup = (defer == 2 && a.kv_same && nw == 8) ? 2 : 1;     // 64-key tiles per ring stage (attn_fwd4_kernel: SUP)
    int lds = 3 * sup * (a.kv_same ? tile : 2 * tile);
    if (lds < nw * 32 * 64 * 4) lds = nw * 32 * 64 * 4;               // the epilogue stages NW x 32 fp32 rows in the ring's memory
    const dim3 grid = lean_grid(a.nseq, a.nhead, qtiles, &a.xcd_swizzle);
#define LEAN_FWD(NWv)                                                                                    \
    do {                                                                                                 \
        if (a.kv_same && defer == 3) attn_fwd4_kernel<64, NWv, true, 3><<<grid, 64 * NWv, lds, s>>>(a);  \
        else if (a.kv_same && defer == 2) attn_fwd4_kernel<64, NWv, true, 2><<<grid, 64 * NWv, lds, s>>>(a);  \
        else if (a.kv_same && defer) attn_fwd4_kernel<64, NWv, true, 1><<<grid, 64 * NWv, lds, s>>>(a);  \
        else if (a.kv_same) attn_fwd4_kernel<64, NWv, true, 0><<<grid, 64 * NWv, lds, s>>>(a);           \
        else if (defer >= 2) attn_fwd4_kernel<64, NWv, false, 2><<<grid, 64 * NWv, lds, s>>>(a);         \
        else if (defer) attn_fwd4_kernel<64, NWv, false, 1><<<grid, 64 * NWv, lds, s>>>(a);              \
        else attn_fwd4_kernel<64, NWv, false, 0><<<grid, 64 * NWv, lds, s>>>(a);                         \
    } while (0)
    if (nw == 8) LEAN_FWD(8); else LEAN_FWD(4);
#undef LEAN_FWD
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}

int genie_attn_lean_bwd_dq(const AttnBwdArgs& a_in, hipStream_t s) {
    AttnBwdArgs a = a_in;
    int nw = (a.Sq + 31) / 32;
    nw = nw >= 3 ? 4 : nw;
    const int qtiles = (a.Sq + 32 * nw - 1) / (32 * nw);
    // (attn_block_of forms nseq * nhead * tiles in an int; one- and two-wave blocks on short sequences make MORE tiles than the forward's)
    GENIE_CHECK_ARG((long long)a.nseq * qtiles * a.nhead < (1ll << 31) - 8 && a.nhead <= 65535, "genie_attention_bwd (dQ): grid too large");
    const int tile = 64 * 64 * 2;
    const int lds = 3 * (a.kv_same ? tile : 2 * tile);
    const dim3 grid = lean_grid(a.nseq, a.nhead, qtiles, &a.xcd_swizzle);
#define LEAN_DQ(NWv)                                                                                     \
    do {                                                                                                 \
        if (a.kv_same) attn_bwd_dq3_kernel<64, NWv, true><<<grid, 64 * NWv, lds, s>>>(a);                \
        else attn_bwd_dq3_kernel<64, NWv, false><<<grid, 64 * NWv, lds, s>>>(a);                         \
    } while (0)
    if (nw == 1) LEAN_DQ(1); else if (nw == 2) LEAN_DQ(2); else LEAN_DQ(4);
#undef LEAN_DQ
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}

int genie_attn_lean_bwd_dkv(const AttnBwdArgs& a_in, hipStream_t s) {
    AttnBwdArgs a = a_in;
    int nw = (a.Sk + 31) / 32;
    nw = nw >= 3 ? 4 : nw;
    const int ktiles = (a.Sk + 32 * nw - 1) / (32 * nw);
    GENIE_CHECK_ARG((long long)a.nseq * ktiles * a.nhead < (1ll << 31) - 8 && a.nhead <= 65535, "genie_attention_bwd (dK / dV): grid too large");
    const int tile = 64 * 64 * 2;
    const int lds = 3 * (2 * tile + 512);
    const dim3 grid = lean_grid(a.nseq, a.nhead, ktiles, &a.xcd_swizzle);
    if (nw == 2) attn_bwd_dkv3_kernel<64, 2, true><<<grid, 128, lds, s>>>(a);
    else attn_bwd_dkv3_kernel<64, 4, true><<<grid, 256, lds, s>>>(a);
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}

// Resident blocks per CU the runtime computes for a lean kernel at its launch configuration (which: 0 forward, 1 dQ, 2 dK / dV; four waves per
// block, K == V layout).  The kernels are budgeted for 4 / 3 / 3 blocks per CU; a toolchain that allocates differently shows here
// (tests/test_gpu_attention.py) instead of as a silent slowdown.
extern "C" int genie_attention_lean_occupancy(int which) {
    int n = -1;
    const int tile = 64 * 64 * 2;
    hipError_t e;
    if (which == 0) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, attn_fwd4_kernel<64, 4, true, 2>, 256, 4 * 32 * 64 * 4);
    else if (which == 1) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, attn_bwd_dq3_kernel<64, 4, true>, 256, 3 * tile);
    else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, attn_bwd_dkv3_kernel<64, 4, true>, 256, 3 * (2 * tile + 512));
    if (e != hipSuccess) { genie_set_error("hipOccupancyMaxActiveBlocksPerMultiprocessor: %s", hipGetErrorString(e)); return -1; }
    return n;
}
