// Conv3d weight gradient for stride-1 convolutions whose taps come in kw-triples (dw = -1, 0, +1), on gfx950 MFMA.
//
//   dW[co][(dt, dh, dw)][ci] += sum_pixels DY[pix][co] * X[pix + (dt, dh, dw)][ci]        (+ dbias[co] += sum DY)
//
// The generic kernel (conv_wgrad.hip) gives every tap its own block, so both operand tiles (32 KB per 64 pixels) are DMA'd
// and read from LDS once per tap.  The three taps of a kw-triple read the SAME dy tile and the SAME x rows shifted by one
// pixel, so here one block owns a whole triple:
//   - dy tile [64 pixels][128 co] and an x IMAGE [(64 / W) rows of W + 2 pixels][128 ci] with explicit zero columns left and
//     right of every image row (written once per block; only the 64 real pixels are DMA'd) are staged once per 64-pixel chunk and serve 3 x 16 = 48 MFMAs per wave-quad: a third of the
//     glds issues and LDS-DMA bytes per MFMA;
//   - the dy fragments (2 transposing LDS reads each) are shared by the three taps in registers, the x fragments are read at
//     row offsets +0 / +1 / +2: 5 fragment reads per 6 MFMAs instead of 6.
// 8 waves (2 per SIMD) as 2 (co) x 4 (ci); a wave owns 64 co x 32 ci x 3 taps = 96 accumulator registers.  Stages live in
// a ring of THREE LDS buffers, issued two chunks ahead; waits are counted (vmcnt) so a whole stage stays in flight across
// the barrier.  Both operands are pixel-major in HBM and in LDS; MFMA fragments come from ds_read_b64_tr_b16 (lane semantics
// pinned by tests/test_gpu_kernels.py::test_probe_ds_read_tr16).  Split-K over pixel ranges, fp32 atomics into the gradient.
#include "common.h"
#include "genie_hip.h"

static __device__ __attribute__((aligned(256))) uint32_t g_zero_page_w3[64];

struct FastDiv3 {
    uint32_t magic, shift, d;
};
static FastDiv3 make_fastdiv3(uint32_t d) {
    FastDiv3 f;
    f.d = d;
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;
    f.shift = 31 + l;
    f.magic = (uint32_t)(((1ull << f.shift) + d - 1) / d);
    return f;
}
__device__ __forceinline__ uint32_t fd3(uint32_t n, const FastDiv3& f) { return (uint32_t)(((uint64_t)n * f.magic) >> f.shift); }   // n < 2^31

struct Wgrad3Args {
    const bf16_t* src;
    const bf16_t* dy;
    float* dw;
    float* dbias;
    const GenieTap* taps;       // dt / dh of triple tr are read from taps[3 tr]
    int ntriples;
    int T, H, W, Cs, Cin;       // x geometry == row grid (stride 1, same size)
    int Td, Hd, Wd, Cd, Cout;   // dy geometry
    int dmt, dmh, dmw;
    int shuf_c, shuf_q, shuf_r, shuf_f;
    long long s_cout, s_tap, s_cin;
    int M, nchunks, split_k, chunks_per_split;
    int tiles_m, tiles_n;
    int img_rows, log2W;        // (64 / W) * (W + 2);  W is a power of two in [8, 64]
    int dbg;                    // timing ablations (wrong results): 1 no MFMAs, 2 no fragment reads / MFMAs, 4 no DMA
    int trim;                   // wgrad3l: skip the chunks of time-padding frames (GENIE_W3_TRIM=1; off by default)
    int row_px, px0;            // wgrad3l, W = 64: memory pixels per image row (>= W) and the window's first column (GenieWgradDesc::row_px; dense: W, 0)
    FastDiv3 dW_, dH_, dT_;
};

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// ds_read_b64_tr_b16 with an immediate byte offset, as inline asm (outside hipcc's waitcnt bookkeeping: the caller waits)
template <int IMM>
__device__ __forceinline__ bf16x4_t w3_tr16(uint32_t lds_addr) {
    static_assert(IMM >= 0 && IMM < 65536, "ds offset field is 16 bits");
    bf16x4_t v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "i"(IMM));
    return v;
}
__device__ __forceinline__ uint32_t w3_lds_offset(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}

// x-image rows that precede chunk pixel kk0 = 16 ks + 4 h2 because of the two pad pixels per image row: 2 * (kk0 >> log2 W).
// (For W = 8 the lane's own k-row adds 2 * (krow >> 3) more: a per-lane constant handled in the VGPR part of the address.)
__host__ __device__ constexpr int w3_pad_rows(int ks, int h2, int log2w) { return 2 * ((ks * 16 + 4 * h2) >> log2w); }

// All fragment reads of k-step KS: dy rows 16 KS + krow (+ 4), x-image rows of the same pixels shifted by s = 0, 1, 2.
// a_addr[i] / b_addr[p][s] hold everything that is not a compile-time constant (stage base, lane's k-row, column, swizzle).
template <int KS, int LOG2W, int TM>
__device__ __forceinline__ void w3_issue(const uint32_t (&a_addr)[TM], const uint32_t (&b_addr)[2][3], bf16x4_t (&alo)[TM],
                                         bf16x4_t (&ahi)[TM], bf16x4_t (&blo)[3], bf16x4_t (&bhi)[3]) {
    constexpr int P0 = (w3_pad_rows(KS, 0, LOG2W) >> 1) & 1, P1 = (w3_pad_rows(KS, 1, LOG2W) >> 1) & 1;   // swizzle parity
    constexpr int X0 = (KS * 16 + w3_pad_rows(KS, 0, LOG2W)) * 256, X1 = (KS * 16 + 4 + w3_pad_rows(KS, 1, LOG2W)) * 256;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        alo[i] = w3_tr16<KS * 16 * 256>(a_addr[i]);
        ahi[i] = w3_tr16<(KS * 16 + 4) * 256>(a_addr[i]);
    }
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        blo[s] = w3_tr16<X0>(b_addr[P0][s]);
        bhi[s] = w3_tr16<X1>(b_addr[P1][s]);
    }
}

#ifndef W3_NSTAGE
#define W3_NSTAGE 4        // LDS ring depth: stages are issued W3_NSTAGE - 1 chunks ahead (4 x 36 KB = 144 KB)
#endif

template <int LOG2W, bool SHUF>
__global__ void __launch_bounds__(512) wgrad3_kernel(const Wgrad3Args a) {
    constexpr int BK = 64;                               // pixels per chunk
    constexpr int W = 1 << LOG2W, WP = W + 2, HS = BK >> LOG2W;   // image rows per chunk
    constexpr int PITCH = 256;                           // bytes per LDS row (128 channels)
    constexpr int A_BYTES = BK * PITCH;                  // dy tile, 16 KB
    constexpr int X_BYTES = 80 * PITCH;                  // x image: (64 / W) * (W + 2) rows, 80 at W = 8; 20 KB
    constexpr int STAGE = A_BYTES + X_BYTES;             // 36 KB
    constexpr int NSTAGE = W3_NSTAGE;
    constexpr int TM = 2;                                // 32-row co tiles per wave; one 32-col ci tile
    constexpr int IMG_ROWS = HS * WP;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;             // 2 (co) x 4 (ci)

    // work id: triples of one (tile, pixel range) are consecutive and consecutive ids stay on ONE XCD, so the 9 triples that
    // re-read the same dy tile and shifted x rows hit that XCD's L2
    int b;
    {
        const int nb = gridDim.x, bid = blockIdx.x;
        const int q = nb >> 3, r = nb & 7, xcd = bid & 7;
        b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tr = b % a.ntriples; b /= a.ntriples;
    const int split = b % a.split_k; b /= a.split_k;
    const int tile_n = b % a.tiles_n, tile_m = b / a.tiles_n;
    const int co0 = tile_m * 128, ci0 = tile_n * 128;
    const GenieTap tp = a.taps[3 * tr];
    const int t_dt = __builtin_amdgcn_readfirstlane(tp.dt), t_dh = __builtin_amdgcn_readfirstlane(tp.dh);
    const bool do_bias = a.dbias != nullptr && tr == 0 && tile_n == 0 && wn == 0;
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page_w3);

    int c_begin = split * a.chunks_per_split, c_end = c_begin + a.chunks_per_split;
    if (c_end > a.nchunks) c_end = a.nchunks;

    // ---- staging state (per lane, fixed for the kernel): which rows / channel chunk this lane fills in every stage ----
    const int d_row = tid >> 4;                          // + 32 * round
    const int st_lc = (tid & 15) ^ ((d_row & 3) << 2);   // logical 16-B chunk of the dy tile (same in every round: round * 32 keeps row & 3)
    int d_coff;                                          // element offset of (sub-pixel, channel) inside a dy pixel, or -1
    {
        const int co = co0 + st_lc * 8;
        if (co < a.Cout) {
            const int sub = co / a.shuf_c, ch = co - sub * a.shuf_c;
            const int r = sub % a.shuf_r, q = (sub / a.shuf_r) % a.shuf_q, p = sub / (a.shuf_r * a.shuf_q);
            d_coff = ((p * a.Hd + q) * a.Wd + r) * a.Cd + ch;
        } else {
            d_coff = -1;
        }
    }
    const int tap_delta = (t_dt * a.H + t_dh) * W * a.Cs;
    // x image: only the 64 real pixels of a chunk are DMA'd (two rounds of 32 rows, like the dy tile) -- pixel q = 32 round + d_row of
    // image row hl = q >> LOG2W lands in LDS row hl * (W + 2) + (q & (W - 1)) + 1; the zero columns left and right of every image
    // row are written ONCE below and never touched again.  (They used to arrive from a zero page with every stage: a third DMA
    // round, 20 % of the stage's LDS-DMA pieces and bytes.)  A wave's 1-KiB piece = 4 consecutive pixels of one image row.
    int x_c[2], x_to[2], x_ho[2];                        // channel element offset (or -1), (t, h) of the pixel's image row
    uint32_t x_dst[2];                                   // byte offset of the wave's piece inside the image
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = i * 32 + d_row;
        const int hl = q >> LOG2W;
        const int lrow = hl * WP + (q & (W - 1)) + 1;    // this lane's LDS row
        const int lc = (tid & 15) ^ ((lrow & 3) << 2);
        x_c[i] = (ci0 + lc * 8 < a.Cs && ci0 + lc * 8 < ((a.Cin + 7) & ~7)) ? ci0 + lc * 8 : -1;
        const int q0 = (i * 8 + wave) * 4;               // first pixel of the wave's piece
        x_dst[i] = (uint32_t)(((q0 >> LOG2W) * WP + (q0 & (W - 1)) + 1) * PITCH);
        const uint32_t rowid = (uint32_t)c_begin * HS + (uint32_t)hl;          // (n, t, h) row of the first staged chunk
        const uint32_t q2 = fd3(rowid, a.dH_);
        x_ho[i] = (int)(rowid - q2 * a.dH_.d);
        x_to[i] = (int)(q2 - fd3(q2, a.dT_) * a.dT_.d);
    }
    // zero columns of every stage's image: rows hl * WP and hl * WP + W + 1 (2 * HS rows of 256 B per stage)
    for (int e = tid; e < NSTAGE * 2 * HS * 16; e += 512) {
        const int st = e / (2 * HS * 16), r2 = (e / 16) % (2 * HS), c = e & 15;
        const int row = (r2 >> 1) * WP + ((r2 & 1) ? W + 1 : 0);
        *reinterpret_cast<u32x4_t*>(smem + st * STAGE + A_BYTES + row * PITCH + c * 16) = u32x4_t{0u, 0u, 0u, 0u};
    }
    __syncthreads();

    int next_chunk = c_begin;                            // stage pieces are issued for consecutive chunks only
    // One stage = four 1-KiB pieces per wave: dy rounds 0 / 1 (PIECE 0 / 1), x rounds 0 / 1 (PIECE 2 / 3; the last one advances the
    // chunk cursor).  In the main loop piece j rides inside k-step j, between the MFMAs of the two co tiles: issued back to back at
    // the top of a chunk the four cost ~220 cycles each with nothing to hide behind (timing ablation: DMA-only 0.13 ms + reads and
    // MFMAs 0.26 ms = the whole 0.37 ms loop -- the phases added up instead of overlapping).
    auto stage_piece = [&](int buf, bool live, int piece) {
        if (a.dbg & 4) { if (piece == 3) ++next_chunk; return; }
        char* abase = smem + buf * STAGE;
        char* xbase = abase + A_BYTES;
        const uint32_t mbase = (uint32_t)next_chunk * BK;
        const int left = a.M - (int)mbase;               // pixels of this chunk that exist (<= 0: none)
        if (piece < 2) {
            const int i = piece;
            const int pl = i * 32 + d_row;
            const bf16_t* q = zero;
            if (live && pl < left && d_coff >= 0) {
                if (SHUF) {
                    const uint32_t m = mbase + pl;
                    const uint32_t q1 = m >> LOG2W, wo = m & (uint32_t)(W - 1);
                    const uint32_t q2 = fd3(q1, a.dH_), ho = q1 - q2 * a.dH_.d;
                    const uint32_t n = fd3(q2, a.dT_), to = q2 - n * a.dT_.d;
                    q = a.dy + ((((n * a.Td + to * a.dmt) * a.Hd + ho * a.dmh) * a.Wd + wo * a.dmw) * a.Cd + (uint32_t)d_coff);
                } else {
                    q = a.dy + ((mbase + pl) * (uint32_t)a.Cd + (uint32_t)d_coff);     // dy has the row grid's geometry: linear
                }
            }
            __builtin_amdgcn_global_load_lds(GLB_PTR(q), LDS_PTR(abase + (i * 8 + wave) * 1024), 16, 0, 0);
        } else {
            const int i = piece - 2;
            const int pl = i * 32 + d_row;
            const bf16_t* q = zero;
            const int t = x_to[i] + t_dt, h = x_ho[i] + t_dh;
            if (live && x_c[i] >= 0 && pl < left && (unsigned)t < (unsigned)a.T && (unsigned)h < (unsigned)a.H)
                q = a.src + (int)((mbase + (uint32_t)pl) * (uint32_t)a.Cs + (uint32_t)(tap_delta + x_c[i]));
            __builtin_amdgcn_global_load_lds(GLB_PTR(q), LDS_PTR(xbase + x_dst[i]), 16, 0, 0);
            // next chunk: HS image rows further
            x_ho[i] += HS;
            while (x_ho[i] >= a.H) {
                x_ho[i] -= a.H;
                x_to[i] = x_to[i] + 1 == a.T ? 0 : x_to[i] + 1;
            }
            if (piece == 3) ++next_chunk;
        }
    };
    auto stage = [&](int buf, bool live) {
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) stage_piece(buf, live, pc);
    };

    f32x16_t acc[3][TM];
    f32x16_t accb[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) accb[i][r] = 0.f;
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[s][i][r] = 0.f;
    }

    // ---- transposing-read addresses: lane = 16 g + 4 r + q reads k-row 8 (g >> 1) + r (+ 16 kstep, + 4 second read),
    //      channels 16 (g & 1) + 4 q .. + 3 of the 32-wide tile.  Everything but the stage base is constant per lane:
    //      dy : row = kk,                    swizzle key = kk & 3 = r
    //      x  : row = kk + 2 (kk >> LOG2W) + s, swizzle key = (r + s + 2 (parity of kk >> LOG2W)) & 3 ----
    const int g16 = lane >> 4, rr = (lane >> 2) & 3, qq = lane & 3;
    const int krow = 8 * (g16 >> 1) + rr;
    const int lane_pad = LOG2W == 3 ? 2 * (g16 >> 1) : 0;            // W = 8: k-rows 8..11 sit one image row further
    uint32_t a_const[TM], b_const[2][3];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int col = wm * 64 + i * 32 + 16 * (g16 & 1) + 4 * qq;
        a_const[i] = (uint32_t)(krow * PITCH + (((col >> 3) ^ (rr << 2)) << 4) + (col & 7) * 2);
    }
    {
        const int col = wn * 32 + 16 * (g16 & 1) + 4 * qq;
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int key = (rr + s + lane_pad + 2 * p) & 3;
                b_const[p][s] = (uint32_t)((krow + lane_pad + s) * PITCH + (((col >> 3) ^ (key << 2)) << 4) + (col & 7) * 2);
            }
    }

    bf16x8_t ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (short)0x3F80;

    const uint32_t smem_base = w3_lds_offset(smem);
    const int nch = c_end - c_begin;
    if (nch > 0) {
        // prologue: NSTAGE - 1 stages in flight, the first one landed
        stage(0, true);
        stage(1, nch > 1);
        if (NSTAGE == 4) stage(2, nch > 2);
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * (NSTAGE - 2)) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        int buf = 0;
        for (int c = 0; c < nch; ++c) {
            const int nbuf = buf >= 1 ? buf - 1 : NSTAGE - 1;            // (buf + NSTAGE - 1) % NSTAGE: slot of chunk c - 1, every wave is past
            const bool nlive = c + NSTAGE - 1 < nch;                      // the barrier behind it
            const uint32_t abase = smem_base + buf * STAGE;
            uint32_t a_addr[TM], b_addr[2][3];
#pragma unroll
            for (int i = 0; i < TM; ++i) a_addr[i] = abase + a_const[i];
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int s = 0; s < 3; ++s) b_addr[p][s] = abase + A_BYTES + b_const[p][s];
            bf16x4_t alo[2][TM], ahi[2][TM], blo[2][3], bhi[2][3];
#define W3_CONSUME(SET)                                                                                          \
            bf16x8_t af[TM], bfr[3];                                                                             \
            _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                                     \
                asm volatile("" : "+v"(alo[SET][i]), "+v"(ahi[SET][i]));                                         \
                af[i] = __builtin_shufflevector(alo[SET][i], ahi[SET][i], 0, 1, 2, 3, 4, 5, 6, 7);               \
            }                                                                                                    \
            _Pragma("unroll") for (int s = 0; s < 3; ++s) {                                                      \
                asm volatile("" : "+v"(blo[SET][s]), "+v"(bhi[SET][s]));                                         \
                bfr[s] = __builtin_shufflevector(blo[SET][s], bhi[SET][s], 0, 1, 2, 3, 4, 5, 6, 7);              \
            }
#define W3_MFMA(KS)                                                                                              \
            _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                                     \
                if (!(a.dbg & 1)) {                                                                              \
                _Pragma("unroll") for (int s = 0; s < 3; ++s)                                                    \
                    acc[s][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[s], acc[s][i], 0, 0, 0);     \
                if (do_bias) accb[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], ones, accb[i], 0, 0, 0);   \
                }                                                                                                \
                if (i == 0) {                                                                                    \
                    __builtin_amdgcn_sched_barrier(0);                                                           \
                    stage_piece(nbuf, nlive, KS);                                                                \
                    __builtin_amdgcn_sched_barrier(0);                                                           \
                }                                                                                                \
            }
#define W3_STEP(KS, SET, NEXT)                                                                                   \
            {                                                                                                    \
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                               \
                __builtin_amdgcn_sched_barrier(0);                                                               \
                W3_CONSUME(SET)                                                                                  \
                NEXT                                                                                             \
                W3_MFMA(KS)                                                                                      \
            }
            if (!(a.dbg & 2)) {
            w3_issue<0, LOG2W, TM>(a_addr, b_addr, alo[0], ahi[0], blo[0], bhi[0]);
            W3_STEP(0, 0, (w3_issue<1, LOG2W, TM>(a_addr, b_addr, alo[1], ahi[1], blo[1], bhi[1]));)
            W3_STEP(1, 1, (w3_issue<2, LOG2W, TM>(a_addr, b_addr, alo[0], ahi[0], blo[0], bhi[0]));)
            W3_STEP(2, 0, (w3_issue<3, LOG2W, TM>(a_addr, b_addr, alo[1], ahi[1], blo[1], bhi[1]));)
            W3_STEP(3, 1, ;)
            } else {
                stage(nbuf, nlive);
            }
#undef W3_STEP
#undef W3_MFMA
#undef W3_CONSUME
            // chunk c + 1 must have landed; the NSTAGE - 2 newest stages (4 glds each) stay in flight
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * (NSTAGE - 2)) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            buf = buf == NSTAGE - 1 ? 0 : buf + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }

    // ---- epilogue: fp32 atomics; D row = cout (registers), col = cin (lane & 31) ----
    const int khalf = lane >> 5;
    const int ci = ci0 + wn * 32 + (lane & 31);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r16 = 0; r16 < 16; ++r16) {
            const int co = co0 + wm * 64 + i * 32 + (r16 & 3) + 8 * (r16 >> 2) + 4 * khalf;
            if (co >= a.Cout) continue;
            const int sub = co / a.shuf_c, ch = co - sub * a.shuf_c;
            const int co_nat = a.shuf_f > 1 ? ch * a.shuf_f + sub : co;
            float* row = a.dw + co_nat * a.s_cout + (long long)(3 * tr) * a.s_tap;
            if (ci < a.Cin) {
                if (a.split_k == 1) {                           // one owner per element: plain read-modify-write (no atomics)
#pragma unroll
                    for (int s = 0; s < 3; ++s) row[s * a.s_tap + ci * a.s_cin] += acc[s][i][r16];
                } else {
#pragma unroll
                    for (int s = 0; s < 3; ++s) atomicAdd(row + s * a.s_tap + ci * a.s_cin, acc[s][i][r16]);
                }
            }
            if (do_bias && (lane & 31) == 0) atomicAdd(a.dbias + co_nat, accb[i][r16]);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Lean main loop (round 3).  Same tiling, LDS layout, ring and MFMA order as wgrad3_kernel above; what changed is everything AROUND
// the MFMAs.  SQ counters of the kernel above (256 -> 256 @16x32x32): 8.3 VALU + 3.9 SALU instructions per MFMA and 27 scalar
// branches per 64-pixel chunk -- per-lane 64-bit address arithmetic, bounds predicates and zero-page selects for every LDS-DMA piece,
// per-lane (t, h) bookkeeping with a while loop, run-time debug / bias switches inside the k-steps.  One wave's VALU issue filled the
// issue slots between its own MFMAs, so the "phases" of a wave (DMA issue, fragment reads, MFMAs) added up instead of overlapping.
// Here:
//   * LDS-DMA through BUFFER loads (`buffer_load_dwordx4 ... lds`): a wave-uniform resource descriptor (SGPRs, rebased to the block's
//     first chunk) + a per-lane byte offset that never changes + a scalar chunk offset that advances by one s_add per chunk.  What
//     used to be predicates is the descriptor's range check: a lane whose channel chunk does not exist, a piece whose image row
//     (t + dt, h + dh) lies outside the clip and a stage issued past the block's last chunk all carry offset 0x80000000 >= num_records
//     and the DMA writes ZEROS (the mechanism composable_kernel's direct loads rely on) -- one v_or per x piece, nothing per dy piece.
//   * (t, h) of a piece's image row is wave-uniform: scalar registers, scalar compare / select.
//   * the bias gradient (an MFMA against ones) is a template parameter of the loop -- WHICH of the chunk's four k-steps this wave also sums --
//     chosen once per wave by a scalar branch in front of it (the four ci-waves of a co-half hold the same dy fragments and take one k-step
//     each, in the block of the first triple / first ci tile); no debug switches.
// Eligibility (host): dy not shuffled, H * W a multiple of 64 (chunks are whole image rows, M a multiple of 64), a block's byte range
// below 2 GiB.  Everything else stays on wgrad3_kernel.
// ------------------------------------------------------------------------------------------------------------------------------
#define W3L_OOB 0x80000000u

template <int LOG2W, int BIAS_KS>       // BIAS_KS: the k-step (0..3) in which this wave also sums the bias gradient, -1: none
__device__ __forceinline__ void w3l_loop(const Wgrad3Args& a, char* smem, const int nch, const int wave,
                                          const __amdgpu_buffer_rsrc_t rs_dy, const __amdgpu_buffer_rsrc_t rs_x,
                                          const uint32_t (&voff_dy)[2], const uint32_t (&voff_x)[2], const uint32_t (&x_dst)[2],
                                          int (&x_t)[2], int (&x_h)[2], const int t_dt, const int t_dh,
                                          const uint32_t (&a_const)[2], const uint32_t (&b_const)[2][3],
                                          f32x16_t (&acc)[3][2], f32x16_t (&accb)[2], int clip_left, const int clip_chunks, const int skip_frames) {
    constexpr int BK = 64, W = 1 << LOG2W, HS = BK >> LOG2W, PITCH = 256;
    constexpr int A_BYTES = BK * PITCH, X_BYTES = 80 * PITCH, STAGE = A_BYTES + X_BYTES, NSTAGE = W3_NSTAGE, TM = 2;
    // bytes per chunk: 64 dense pixels, or -- a 64-column window of wider rows (HS == 1) -- one memory row
    const uint32_t chunk_px = (uint32_t)(a.row_px > W ? a.row_px : BK);
    const uint32_t dy_step = chunk_px * (uint32_t)(a.Cd * 2), x_step = chunk_px * (uint32_t)(a.Cs * 2);
    // zero-frame skipping (skip_frames = |dt| > 0): the block walks only the chunks of frames t with 0 <= t + dt < T -- clip_chunks
    // consecutive chunks per clip; after the last of them the cursor jumps skip_frames frames ahead (to the first valid frame of the next clip)
    const uint32_t dy_skip = (uint32_t)skip_frames * (uint32_t)(a.H * a.row_px) * (uint32_t)(a.Cd * 2);
    const uint32_t x_skip = (uint32_t)skip_frames * (uint32_t)(a.H * a.row_px) * (uint32_t)(a.Cs * 2);
    uint32_t so_dy = 0, so_x = 0;                        // scalar byte offsets of the NEXT chunk to stage, relative to the block's first
    int staged = 0;                                      // chunks staged so far

    // one 1-KiB piece of the stage for chunk `staged`: dy rounds 0 / 1 (PIECE 0 / 1), x rounds 0 / 1 (PIECE 2 / 3; 3 advances the cursor)
    auto stage_piece = [&](const int buf, const int piece) {
        char* abase = smem + buf * STAGE;
        const bool live = staged < nch;
        if (piece < 2) {
            const uint32_t v = voff_dy[piece] | (live ? 0u : W3L_OOB);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_dy, LDS_PTR(abase + (piece * 8 + wave) * 1024), 16, v, live ? so_dy : 0u, 0, 0);
        } else {
            const int i = piece - 2;
            const bool ok = (int)live & (int)((unsigned)(x_t[i] + t_dt) < (unsigned)a.T) & (int)((unsigned)(x_h[i] + t_dh) < (unsigned)a.H);   // (no short-circuit branches)
            const uint32_t v = voff_x[i] | (ok ? 0u : W3L_OOB);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, LDS_PTR(abase + A_BYTES + x_dst[i]), 16, v, ok ? so_x : 0u, 0, 0);
            x_h[i] += HS;                                // next chunk: HS image rows further (HS <= H: at most one wrap)
            if (x_h[i] >= a.H) {
                x_h[i] -= a.H;
                x_t[i] = x_t[i] + 1 == a.T ? 0 : x_t[i] + 1;
            }
            if (piece == 3) {
                ++staged;
                so_dy += dy_step;
                so_x += x_step;
                if (skip_frames > 0 && --clip_left == 0) {           // (scalar: one s_cmp per chunk)
                    clip_left = clip_chunks;
                    so_dy += dy_skip;
                    so_x += x_skip;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        x_t[q] += skip_frames;
                        if (x_t[q] >= a.T) x_t[q] -= a.T;
                    }
                }
            }
        }
    };
    auto stage = [&](const int buf) {
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) stage_piece(buf, pc);
    };

    bf16x8_t ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (short)0x3F80;

    const uint32_t smem_base = w3_lds_offset(smem);
    // prologue: NSTAGE - 1 stages in flight, the first one landed
    stage(0);
    stage(1);
    if (NSTAGE == 4) stage(2);
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * (NSTAGE - 2)) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int buf = 0;
    for (int c = 0; c < nch; ++c) {
        const int nbuf = buf >= 1 ? buf - 1 : NSTAGE - 1;                // slot of chunk c - 1: every wave is past the barrier behind it
        const uint32_t abase = smem_base + buf * STAGE;
        uint32_t a_addr[TM], b_addr[2][3];
#pragma unroll
        for (int i = 0; i < TM; ++i) a_addr[i] = abase + a_const[i];
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int s = 0; s < 3; ++s) b_addr[p][s] = abase + A_BYTES + b_const[p][s];
        bf16x4_t alo[2][TM], ahi[2][TM], blo[2][3], bhi[2][3];
#define W3L_CONSUME(SET)                                                                                         \
        bf16x8_t af[TM], bfr[3];                                                                                 \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                                         \
            asm volatile("" : "+v"(alo[SET][i]), "+v"(ahi[SET][i]));                                             \
            af[i] = __builtin_shufflevector(alo[SET][i], ahi[SET][i], 0, 1, 2, 3, 4, 5, 6, 7);                   \
        }                                                                                                        \
        _Pragma("unroll") for (int s = 0; s < 3; ++s) {                                                          \
            asm volatile("" : "+v"(blo[SET][s]), "+v"(bhi[SET][s]));                                             \
            bfr[s] = __builtin_shufflevector(blo[SET][s], bhi[SET][s], 0, 1, 2, 3, 4, 5, 6, 7);                  \
        }
#define W3L_MFMA(KS)                                                                                             \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                                         \
            _Pragma("unroll") for (int s = 0; s < 3; ++s)                                                        \
                acc[s][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[s], acc[s][i], 0, 0, 0);         \
            if (BIAS_KS == KS) accb[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], ones, accb[i], 0, 0, 0); \
            if (i == 0) {                                                                                        \
                __builtin_amdgcn_sched_barrier(0);                                                               \
                stage_piece(nbuf, KS);                                                                           \
                __builtin_amdgcn_sched_barrier(0);                                                               \
            }                                                                                                    \
        }
#define W3L_STEP(KS, SET, NEXT)                                                                                  \
        {                                                                                                        \
            __builtin_amdgcn_sched_barrier(0);      /* the wait stays BEHIND the previous k-step's MFMAs */      \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                   \
            __builtin_amdgcn_sched_barrier(0);                                                                   \
            W3L_CONSUME(SET)                                                                                     \
            NEXT                                                                                                 \
            W3L_MFMA(KS)                                                                                         \
        }
        w3_issue<0, LOG2W, TM>(a_addr, b_addr, alo[0], ahi[0], blo[0], bhi[0]);
        W3L_STEP(0, 0, (w3_issue<1, LOG2W, TM>(a_addr, b_addr, alo[1], ahi[1], blo[1], bhi[1]));)
        W3L_STEP(1, 1, (w3_issue<2, LOG2W, TM>(a_addr, b_addr, alo[0], ahi[0], blo[0], bhi[0]));)
        W3L_STEP(2, 0, (w3_issue<3, LOG2W, TM>(a_addr, b_addr, alo[1], ahi[1], blo[1], bhi[1]));)
        W3L_STEP(3, 1, ;)
#undef W3L_STEP
#undef W3L_MFMA
#undef W3L_CONSUME
        // chunk c + 1 must have landed; the NSTAGE - 2 newest stages (4 pieces each) stay in flight
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * (NSTAGE - 2)) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        buf = buf == NSTAGE - 1 ? 0 : buf + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int LOG2W>
__global__ void __launch_bounds__(512) wgrad3l_kernel(const Wgrad3Args a) {
    constexpr int BK = 64;
    constexpr int W = 1 << LOG2W, WP = W + 2, HS = BK >> LOG2W;
    constexpr int PITCH = 256, A_BYTES = BK * PITCH, X_BYTES = 80 * PITCH, STAGE = A_BYTES + X_BYTES, NSTAGE = W3_NSTAGE, TM = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;             // 2 (co) x 4 (ci)
    int b;
    {
        const int nb = gridDim.x, bid = blockIdx.x;
        const int q = nb >> 3, r = nb & 7, xcd = bid & 7;
        b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tr = b % a.ntriples; b /= a.ntriples;
    const int split = b % a.split_k; b /= a.split_k;
    const int tile_n = b % a.tiles_n, tile_m = b / a.tiles_n;
    const int co0 = tile_m * 128, ci0 = tile_n * 128;
    const GenieTap tp = a.taps[3 * tr];
    const int t_dt = __builtin_amdgcn_readfirstlane(tp.dt), t_dh = __builtin_amdgcn_readfirstlane(tp.dh);
    // bias gradient (column sums of dy = an MFMA against ones): in the block of the first triple / first ci tile.  Its four ci-waves of a
    // co-half all hold the SAME dy fragments, so each takes ONE of the chunk's four k-steps (k-step == wn, a template parameter of the
    // loop: no branch inside it) -- two extra MFMAs per wave and chunk; round 3 gave all eight to the wn == 0 waves (+33 % on that block's
    // pace, +5...8 % on the launch).  Every wave of the block adds its partial column sums in the epilogue.
    const bool do_bias = a.dbias != nullptr && tr == 0 && tile_n == 0;

    int c_begin = split * a.chunks_per_split, c_end = c_begin + a.chunks_per_split;
    if (c_end > a.nchunks) c_end = a.nchunks;
    int nch = c_end - c_begin;
    // Zero-frame skipping: for a tap with dt != 0 the chunks of the |dt| frames per clip whose source frame t + dt is time padding stage
    // zeros and multiply them.  The block then partitions the VALID chunks (per clip: (T - |dt|) * cpf consecutive ones, starting at frame
    // max(0, -dt)) instead.  Not for the block that also sums the bias gradient (that sum runs over every pixel).
    int clip_left = 0, clip_chunks = 0, skip_frames = 0;
    {
        const int adt = t_dt < 0 ? -t_dt : t_dt;
        const int cpf = (a.H * a.W) >> 6;                                    // chunks per frame (H * W is a multiple of 64 here)
        const bool bias_block = a.dbias != nullptr && tr == 0 && tile_n == 0;
        if (a.trim && adt > 0 && adt < a.T && !bias_block) {
            const int nclips = a.nchunks / (a.T * cpf);
            clip_chunks = (a.T - adt) * cpf;
            const int nvalid = nclips * clip_chunks;
            const int per = (nvalid + a.split_k - 1) / a.split_k;
            int v0 = split * per, v1 = v0 + per;
            v0 = v0 > nvalid ? nvalid : v0;
            v1 = v1 > nvalid ? nvalid : v1;
            const int n0 = v0 / clip_chunks, r0 = v0 - n0 * clip_chunks;
            c_begin = __builtin_amdgcn_readfirstlane((n0 * a.T + (t_dt < 0 ? adt : 0)) * cpf + r0);
            nch = __builtin_amdgcn_readfirstlane(v1 - v0);
            clip_left = __builtin_amdgcn_readfirstlane(clip_chunks - r0);
            skip_frames = adt;
        }
    }

    // ---- per-lane staging constants: byte offsets inside a chunk (never change), OOB for channel chunks that do not exist ----
    const int d_row = tid >> 4;                          // + 32 * round
    const int st_lc = (tid & 15) ^ ((d_row & 3) << 2);   // logical 16-B chunk of the dy tile
    uint32_t voff_dy[2], voff_x[2], x_dst[2];
    int x_t[2], x_h[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int pl = i * 32 + d_row;
        const int co = co0 + st_lc * 8;
        voff_dy[i] = co < a.Cout ? (uint32_t)((pl * a.Cd + co) * 2) : W3L_OOB;
        const int hl = pl >> LOG2W;
        const int lrow = hl * WP + (pl & (W - 1)) + 1;   // this lane's LDS row of the x image
        const int lc = (tid & 15) ^ ((lrow & 3) << 2);
        const int ci = ci0 + lc * 8;
        voff_x[i] = (ci < a.Cs && ci < ((a.Cin + 7) & ~7)) ? (uint32_t)((pl * a.Cs + ci) * 2) : W3L_OOB;
        const int q0 = (i * 8 + wave) * 4;               // first pixel of the wave's piece (4 consecutive pixels of ONE image row)
        x_dst[i] = (uint32_t)(((q0 >> LOG2W) * WP + (q0 & (W - 1)) + 1) * PITCH);
        const uint32_t rowid = (uint32_t)c_begin * HS + (uint32_t)(q0 >> LOG2W);       // (n, t, h) row of the wave's piece, first chunk
        const uint32_t q2 = fd3(rowid, a.dH_);
        x_h[i] = __builtin_amdgcn_readfirstlane((int)(rowid - q2 * a.dH_.d));
        x_t[i] = __builtin_amdgcn_readfirstlane((int)(q2 - fd3(q2, a.dT_) * a.dT_.d));
    }
    // resource descriptors rebased to the block's first chunk (x: to the tap's shifted row as well; only in-range rows are dereferenced)
    const long long chunk_px = a.row_px > W ? a.row_px : BK;
    const long long dy_base = ((long long)c_begin * chunk_px + a.px0) * a.Cd;
    const long long x_base = ((long long)c_begin * chunk_px + a.px0) * a.Cs + (long long)(t_dt * a.H + t_dh) * a.row_px * a.Cs;
    const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc((void*)(a.dy + dy_base), (short)0, (int)W3L_OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(a.src + x_base), (short)0, (int)W3L_OOB, 0x00020000);

    // zero columns of every stage's image: rows hl * WP and hl * WP + W + 1 (written once, never DMA'd)
    for (int e = tid; e < NSTAGE * 2 * HS * 16; e += 512) {
        const int st = e / (2 * HS * 16), r2 = (e / 16) % (2 * HS), c = e & 15;
        const int row = (r2 >> 1) * WP + ((r2 & 1) ? W + 1 : 0);
        *reinterpret_cast<u32x4_t*>(smem + st * STAGE + A_BYTES + row * PITCH + c * 16) = u32x4_t{0u, 0u, 0u, 0u};
    }
    __syncthreads();

    f32x16_t acc[3][TM];
    f32x16_t accb[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) accb[i][r] = 0.f;
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[s][i][r] = 0.f;
    }

    // transposing-read addresses (see wgrad3_kernel)
    const int g16 = lane >> 4, rr = (lane >> 2) & 3, qq = lane & 3;
    const int krow = 8 * (g16 >> 1) + rr;
    const int lane_pad = LOG2W == 3 ? 2 * (g16 >> 1) : 0;
    uint32_t a_const[TM], b_const[2][3];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int col = wm * 64 + i * 32 + 16 * (g16 & 1) + 4 * qq;
        a_const[i] = (uint32_t)(krow * PITCH + (((col >> 3) ^ (rr << 2)) << 4) + (col & 7) * 2);
    }
    {
        const int col = wn * 32 + 16 * (g16 & 1) + 4 * qq;
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int key = (rr + s + lane_pad + 2 * p) & 3;
                b_const[p][s] = (uint32_t)((krow + lane_pad + s) * PITCH + (((col >> 3) ^ (key << 2)) << 4) + (col & 7) * 2);
            }
    }

    if (nch > 0) {
#define W3L_RUN(BKS, CL, CC, SF) w3l_loop<LOG2W, BKS>(a, smem, nch, wave, rs_dy, rs_x, voff_dy, voff_x, x_dst, x_t, x_h, t_dt, t_dh, a_const, b_const, acc, accb, CL, CC, SF)
        if (!do_bias) W3L_RUN(-1, clip_left, clip_chunks, skip_frames);
        else if (wn == 0) W3L_RUN(0, 0, 0, 0);
        else if (wn == 1) W3L_RUN(1, 0, 0, 0);
        else if (wn == 2) W3L_RUN(2, 0, 0, 0);
        else W3L_RUN(3, 0, 0, 0);
#undef W3L_RUN
    }

    // ---- epilogue: fp32 atomics; D row = cout (registers), col = cin (lane & 31) ----
    const int khalf = lane >> 5;
    const int ci = ci0 + wn * 32 + (lane & 31);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r16 = 0; r16 < 16; ++r16) {
            const int co = co0 + wm * 64 + i * 32 + (r16 & 3) + 8 * (r16 >> 2) + 4 * khalf;
            if (co >= a.Cout) continue;
            // un-shuffled upsample gradient (dy_unshuffled): dy channel co = sub-pixel co / shuf_c, channel co % shuf_c -> natural weight row
            const int co_nat = a.shuf_f > 1 ? (co % a.shuf_c) * a.shuf_f + co / a.shuf_c : co;
            float* row = a.dw + co_nat * a.s_cout + (long long)(3 * tr) * a.s_tap;
            if (ci < a.Cin) {
                if (a.split_k == 1) {                           // one owner per element: plain read-modify-write (no atomics)
#pragma unroll
                    for (int s = 0; s < 3; ++s) row[s * a.s_tap + ci * a.s_cin] += acc[s][i][r16];
                } else {
#pragma unroll
                    for (int s = 0; s < 3; ++s) atomicAdd(row + s * a.s_tap + ci * a.s_cin, acc[s][i][r16]);
                }
            }
            if (do_bias && (lane & 31) == 0) atomicAdd(a.dbias + co_nat, accb[i][r16]);
        }
    }
}

// Returns 1 when the problem is not eligible (caller falls back to the generic kernel), 0 on launch, < 0 on error.
int genie_conv_wgrad3_try(const GenieWgradDesc* d, hipStream_t s) {
    if (d->tri_mode <= 0) return 1;
    if (d->dy_unshuffled && !(d->Td == d->To && d->Hd == d->Ho && d->Wd == d->Wo && d->dmt == 1 && d->dmh == 1 && d->dmw == 1 && d->shuf_c < d->Cout &&
                              d->Cout % d->shuf_c == 0)) {
        genie_set_error("genie_conv_wgrad: dy_unshuffled wants dy on the conv's own row grid and shuf_c | Cout");
        return GENIE_ERR_ARG;
    }
    if (d->st != 1 || d->sh != 1 || d->sw != 1 || d->To != d->Ts || d->Ho != d->Hs || d->Wo != d->Ws) return 1;
    if (d->ntaps % 3 != 0 || d->ntaps < 3) return 1;
    const int W = d->Wo;
    const bool window = d->row_px != 0;
    if (window && !(W == 64 && d->Wd == 64 && d->row_px >= 64 && d->px0 >= 0 && d->px0 + 64 <= d->row_px && !d->dy_unshuffled && d->shuf_c >= d->Cout)) {
        genie_set_error("genie_conv_wgrad: a W-window wants Ws = Wo = Wd = 64 inside row_px >= px0 + 64 pixels and a plain (unshuffled) dy");
        return GENIE_ERR_ARG;
    }
    if (!(W == 8 || W == 16 || W == 32 || W == 64)) return 1;
    if (d->Cin < 64 || d->Cout < 64) return 1;
    Wgrad3Args a;
    a.src = (const bf16_t*)d->src; a.dy = (const bf16_t*)d->dy; a.dw = d->dw; a.dbias = d->dbias; a.taps = d->taps;
    a.ntriples = d->ntaps / 3;
    a.T = d->Ts; a.H = d->Hs; a.W = W; a.Cs = d->Cs; a.Cin = d->Cin;
    a.Td = d->Td; a.Hd = d->Hd; a.Wd = d->Wd; a.Cd = d->Cd; a.Cout = d->Cout;
    a.dmt = d->dmt; a.dmh = d->dmh; a.dmw = d->dmw;
    const bool shuffled = d->shuf_c < d->Cout && !d->dy_unshuffled;          // dy addressed THROUGH the shuffle (general kernel only)
    if (d->shuf_c < d->Cout) {
        if (d->shuf_c % 8 != 0 || d->Cout % d->shuf_c != 0) return 1;
        a.shuf_c = d->shuf_c; a.shuf_q = d->shuf_q; a.shuf_r = d->shuf_r; a.shuf_f = d->Cout / d->shuf_c;
    } else {
        a.shuf_c = d->Cd > d->Cout ? d->Cd : d->Cout; a.shuf_q = 1; a.shuf_r = 1; a.shuf_f = 1;
    }
    a.s_cout = d->s_cout; a.s_tap = d->s_tap; a.s_cin = d->s_cin;
    const long long M = (long long)d->N * d->To * d->Ho * d->Wo;
    if (M <= 0 || M >= (1ll << 31)) return 1;
    a.M = (int)M;
    a.nchunks = cdiv(M, 64);
    a.tiles_m = cdiv(d->Cout, 128);
    a.tiles_n = cdiv((d->Cin + 7) & ~7, 128);
    a.img_rows = (64 / W) * (W + 2);
    a.log2W = W == 8 ? 3 : (W == 16 ? 4 : (W == 32 ? 5 : 6));
    { static const int dbg = getenv("GENIE_W3_DBG") ? atoi(getenv("GENIE_W3_DBG")) : 0; a.dbg = dbg; }
    a.dW_ = make_fastdiv3(d->Wo); a.dH_ = make_fastdiv3(d->Ho); a.dT_ = make_fastdiv3(d->To);
    const long long base = (long long)a.tiles_m * a.tiles_n * a.ntriples;
    int sk = d->split_k;
    if (sk <= 0) {
        // one block per CU (8 waves, 120 KB LDS): a whole number of 256-block rounds, as many rounds (up to 3) as leave >= 32
        // chunks per block -- the 196 KB of atomics per block and the pipeline fill need a long K loop to amortise
        sk = 0;
        static const int max_rounds = getenv("GENIE_W3_ROUNDS") ? atoi(getenv("GENIE_W3_ROUNDS")) : 3;
        for (int rounds = max_rounds; rounds >= 1 && sk == 0; --rounds) {
            const int cand = (int)((256ll * rounds) / base);
            if (cand >= 1 && a.nchunks / cand >= 32) sk = cand;
        }
        if (sk == 0) sk = 1;
        if (base * sk < 200) {                           // small problem: fill the chip even at 16 chunks per block
            const int want = (int)((256 + base - 1) / base);
            if (a.nchunks / want >= 16) sk = want;
        }
        static const int small_sk1 = getenv("GENIE_W3_SMALL_SK1") ? atoi(getenv("GENIE_W3_SMALL_SK1")) : 0;
        if (small_sk1 && base >= small_sk1) sk = 1;      // >= that many blocks without a split: one split, plain read-modify-write epilogue
        if (d->tri_mode == 1 && (base * sk < 200 || a.nchunks / sk < 16)) return 1;      // too little work: generic kernel (tri_mode 2 forces)
    }
    a.chunks_per_split = cdiv(a.nchunks, sk);
    a.split_k = cdiv(a.nchunks, a.chunks_per_split);
    constexpr int lds = W3_NSTAGE * (64 * 256 + 80 * 256);
    void (*kern)(const Wgrad3Args) = nullptr;
    switch (a.log2W * 2 + (shuffled ? 1 : 0)) {
        case 6: kern = wgrad3_kernel<3, false>; break;
        case 7: kern = wgrad3_kernel<3, true>; break;
        case 8: kern = wgrad3_kernel<4, false>; break;
        case 9: kern = wgrad3_kernel<4, true>; break;
        case 10: kern = wgrad3_kernel<5, false>; break;
        case 11: kern = wgrad3_kernel<5, true>; break;
        case 12: kern = wgrad3_kernel<6, false>; break;
        default: kern = wgrad3_kernel<6, true>; break;
    }
    static bool configured[16] = {false};
    const int slot = a.log2W * 2 + (shuffled ? 1 : 0);
    if (!configured[slot]) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            genie_set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return GENIE_ERR_HIP;
        }
        configured[slot] = true;
    }
    genie_note_variant(GENIE_VARIANT_WGRAD3);
    // lean main loop (buffer-addressed LDS-DMA, scalar bookkeeping) where its preconditions hold -- see wgrad3l_kernel
    static const int lean_on = getenv("GENIE_W3_LEAN") ? atoi(getenv("GENIE_W3_LEAN")) : 1;
    a.row_px = window ? d->row_px : W;
    a.px0 = window ? d->px0 : 0;
    const long long chunk_px = a.row_px > W ? a.row_px : 64;
    const long long blk_bytes = (long long)a.chunks_per_split * chunk_px * (d->Cs > d->Cd ? d->Cs : d->Cd) * 2 + ((long long)(d->Hs + 2) * a.row_px * d->Cs * 2);
    a.trim = 0;
    if (lean_on && !shuffled && a.dbg == 0 && (d->Ho * W) % 64 == 0 && 64 / W <= d->Ho && blk_bytes < 0x7f000000ll && d->Td == d->Ts && d->Hd == d->Hs && d->Wd == W) {
        // zero-frame skipping (GENIE_W3_TRIM=1; OFF by default): a block's address range grows by the (<= 2 per clip) padding frames it
        // jumps over.  Measured on one box at 64 clips: kernel time unchanged (a launch is one round of blocks and the dt = 0 blocks set the
        // makespan), step -0.4 %, but FETCH_SIZE per launch 7.26 -> 9.88 GB (128 -> 128 @16x64x64): the dt != 0 blocks no longer walk the
        // same dy chunks at the same time as their dt = 0 siblings and lose the L2 sharing (profiles/r03_zero_frame_trim_ab.log)
        static const int trim_on = getenv("GENIE_W3_TRIM") ? atoi(getenv("GENIE_W3_TRIM")) : 0;
        const long long cpf = (long long)d->Ho * W / 64;
        const long long span = a.chunks_per_split + (a.chunks_per_split / (cpf * (d->To > 2 ? d->To - 2 : 1)) + 2) * 2 * cpf;
        a.trim = trim_on && !window && d->To >= 3 && span * 64 * (d->Cs > d->Cd ? d->Cs : d->Cd) * 2 + ((long long)(d->Hs + 2) * W * d->Cs * 2) < 0x7f000000ll;
        void (*lk)(const Wgrad3Args) = a.log2W == 3 ? wgrad3l_kernel<3> : a.log2W == 4 ? wgrad3l_kernel<4> : a.log2W == 5 ? wgrad3l_kernel<5> : wgrad3l_kernel<6>;
        static bool lconf[8] = {false};
        if (!lconf[a.log2W]) {
            hipError_t e = hipFuncSetAttribute((const void*)lk, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (e != hipSuccess) {
                genie_set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e));
                return GENIE_ERR_HIP;
            }
            lconf[a.log2W] = true;
        }
        genie_note_variant(GENIE_VARIANT_WGRAD3_LEAN);
        hipLaunchKernelGGL(lk, dim3((unsigned)(base * a.split_k)), dim3(512), lds, s, a);
        GENIE_CHECK_LAUNCH();
        return GENIE_OK;
    }
    if (window) return 1;                                 // (the caller reports it: only the lean kernel knows windows)
    if (d->dy_unshuffled) {
        genie_set_error("genie_conv_wgrad: dy_unshuffled is served by the lean kw-triple kernel only (H * W %% 64 == 0, 64 / W <= H, block range < 2 GiB)");
        return GENIE_ERR_ARG;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(base * a.split_k)), dim3(512), lds, s, a);
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}
