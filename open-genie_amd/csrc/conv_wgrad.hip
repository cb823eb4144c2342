// Conv3d weight gradient on gfx950 MFMA.
//
//   dW[co][tap][ci] += sum_m  DY[dpix(m)][co] * X[pix(m) * stride + off_tap][ci]          (+ dbias[co] += sum_m DY)
//
// GEMM view per tap: M = cout, N = cin, K = output pixels (huge) -> split-K over pixel ranges with fp32
// atomics into the (zero-initialised / accumulated) gradient arena.  Both operands are stored
// pixel-major in HBM ([pixel][channel], channel contiguous) while MFMA wants 8 consecutive k (= pixels)
// per lane, i.e. both need a transpose.  The tiles are DMA'd as they lie ([64 pixels][BM|BN channels])
// and the fragments are read with ds_read_b64_tr_b16, gfx950's transposing LDS read: inside a 16-lane
// group the lanes' 8-byte rows form a 4(k) x 16(channel) block and lane i receives column i, so two reads
// give a lane its 8 k-values for one channel.  (Lane semantics pinned by tests/test_gpu_kernels.py::
// test_probe_ds_read_tr16.)  Rows are 256 B apart, so the four k-rows of a block would hit the same
// banks: 16-B chunk c of row r is stored at chunk c ^ ((r & 3) << 2) (source-side swizzle, the DMA
// writes lane-linear).
//
// Zero padding / causal padding / stride / the depth-to-space shuffle of upsample convs are all address
// arithmetic on the gather (out-of-range lanes read a zero page).
#include "common.h"
#include "genie_hip.h"

static __device__ __attribute__((aligned(256))) uint32_t g_zero_page_w[64];

struct FastDiv {
    uint32_t magic, shift, d;
};

static FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    f.d = d;
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;
    f.shift = 31 + l;
    f.magic = (uint32_t)(((1ull << f.shift) + d - 1) / d);
    return f;
}

// exact for n < 2^31
__device__ __forceinline__ uint32_t fd_div(uint32_t n, const FastDiv& f) {
    return (uint32_t)(((uint64_t)n * f.magic) >> f.shift);
}

struct WgradArgs {
    const bf16_t* src;
    const bf16_t* dy;
    float* dw;
    float* dbias;
    const GenieTap* taps;
    int ntaps;
    int N, Ts, Hs, Ws, Cs, Cin;
    int To, Ho, Wo, st, sh, sw;
    int Td, Hd, Wd, Cd, Cout;
    int dmt, dmh, dmw;
    int shuf_c, shuf_q, shuf_r, shuf_f;   // shuf_f = P*Q*R (1: no shuffle)
    long long s_cout, s_tap, s_cin;
    int M, nchunks, split_k, chunks_per_split;
    int tiles_m, tiles_n;
    FastDiv dWo, dHo, dTo;
};

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

__device__ __forceinline__ bf16x4_t lds_tr16(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4_t*)(p));
}

// The same read as inline asm.  hipcc's waitcnt pass treats the builtin as "may alias the in-flight LDS-DMA" and puts
// s_waitcnt vmcnt(0) in front of it, which serialises the next chunk's loads with this chunk's MFMAs; an asm statement
// is outside its bookkeeping, so the loads stay in flight.  The caller waits (lgkmcnt) before consuming.
__device__ __forceinline__ bf16x4_t lds_tr16_asm(uint32_t lds_addr) {
    bf16x4_t v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(lds_addr));
    return v;
}
__device__ __forceinline__ uint32_t lds_offset(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}

template <int BM, int BN, int WM, int WN>
__global__ void __launch_bounds__(256) wgrad_kernel(const WgradArgs a) {
    constexpr int BK = 64;                                  // pixels per chunk
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int A_PITCH = BM * 2, B_PITCH = BN * 2;       // bytes per pixel row
    constexpr int A_BYTES = BK * A_PITCH, B_BYTES = BK * B_PITCH, STAGE = A_BYTES + B_BYTES;
    constexpr int A_LOADS = BM / 32, B_LOADS = BN / 32;     // 16-B lane loads per thread per chunk
    constexpr int A_CPR = BM / 8, B_CPR = BN / 8;           // chunks per row
    constexpr int A_RPS = 64 / A_CPR, B_RPS = 64 / B_CPR;   // rows per 1-KiB DMA slab
    static_assert(WM * WN == 4, "4 waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // Work order: the taps of one (tile, pixel range) are consecutive, and consecutive work ids stay on ONE XCD
    // (blocks are dealt round-robin to the 8 XCDs), so the 27 taps that re-read the same dy tile and shifted x tiles
    // hit that XCD's L2 instead of streaming both tensors from HBM once per tap.
    int b;
    {
        const int nb = gridDim.x, bid = blockIdx.x;
        const int q = nb >> 3, r = nb & 7, xcd = bid & 7;
        b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tap_i = b % a.ntaps; b /= a.ntaps;
    const int split = b % a.split_k; b /= a.split_k;
    const int tile_n = b % a.tiles_n, tile_m = b / a.tiles_n;
    const int co0 = tile_m * BM, ci0 = tile_n * BN;
    const GenieTap tp = a.taps[tap_i];
    const bool do_bias = a.dbias != nullptr && tap_i == 0 && tile_n == 0;
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page_w);

    int c_begin = split * a.chunks_per_split, c_end = c_begin + a.chunks_per_split;
    if (c_end > a.nchunks) c_end = a.nchunks;

    // per-lane constant parts of the gather
    int a_row[A_LOADS], a_coff[A_LOADS];        // dy: row in chunk, element offset (sub-pixel + channel) or -1
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
        const int slab = i * 4 + wave;
        const int row = slab * A_RPS + lane / A_CPR;
        const int pc = lane % A_CPR;
        const int lc = (A_CPR >= 16) ? (pc ^ ((row & 3) << 2)) : pc;
        a_row[i] = row;
        const int co = co0 + lc * 8;             // position in the (sub-pixel-major) column order
        if (co < a.Cout) {
            const int sub = co / a.shuf_c, ch = co - sub * a.shuf_c;
            const int r = sub % a.shuf_r, q = (sub / a.shuf_r) % a.shuf_q, p = sub / (a.shuf_r * a.shuf_q);
            a_coff[i] = ((p * a.Hd + q) * a.Wd + r) * a.Cd + ch;
        } else {
            a_coff[i] = -1;
        }
    }
    int b_row[B_LOADS], b_c[B_LOADS];
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) {
        const int slab = i * 4 + wave;
        const int row = slab * B_RPS + lane / B_CPR;
        const int pc = lane % B_CPR;
        const int lc = (B_CPR >= 16) ? (pc ^ ((row & 3) << 2)) : pc;
        b_row[i] = row;
        const int ci = ci0 + lc * 8;
        b_c[i] = ci < a.Cs && ci < ((a.Cin + 7) & ~7) ? ci : -1;
    }

    const int tap_delta = ((tp.dt * a.Hs + tp.dh) * a.Ws + tp.dw) * a.Cs;

    auto stage = [&](int chunk, int buf) {
        char* abase = smem + buf * STAGE;
        char* bbase = abase + A_BYTES;
        const int mbase = chunk * BK;
        if constexpr (BM == 128 && BN == 128) {
            // The 4 A loads and 4 B loads of a lane touch the same 4 pixel rows, and 16 lanes share each row: decode the
            // wave's 16 rows ONCE (lane r & 15 owns row r) and broadcast the results with ds_bpermute.
            const int r = lane & 15;
            const uint32_t m = mbase + ((r >> 2) * 4 + wave) * 4 + (r & 3);
            uint32_t dyoff = 0, xoff = 0, thw = 0x80000000u;          // bit 31: row invalid
            if (m < (uint32_t)a.M) {
                uint32_t q1 = fd_div(m, a.dWo); const uint32_t wo = m - q1 * a.dWo.d;
                uint32_t q2 = fd_div(q1, a.dHo); const uint32_t ho = q1 - q2 * a.dHo.d;
                uint32_t n = fd_div(q2, a.dTo); const uint32_t to = q2 - n * a.dTo.d;
                dyoff = (((n * a.Td + to * a.dmt) * a.Hd + ho * a.dmh) * a.Wd + wo * a.dmw) * a.Cd;
                const uint32_t t0 = to * a.st, h0 = ho * a.sh, w0 = wo * a.sw;
                xoff = (((n * a.Ts + t0) * a.Hs + h0) * a.Ws + w0) * a.Cs;
                thw = (t0 << 20) | (h0 << 10) | w0;                   // 10 bits each (host checks the ranges)
            }
#pragma unroll
            for (int i = 0; i < A_LOADS; ++i) {
                const int srcl = (i * 4 + (lane >> 4)) << 2;          // byte index for ds_bpermute
                const uint32_t v_dy = (uint32_t)__builtin_amdgcn_ds_bpermute(srcl, (int)dyoff);
                const uint32_t v_x = (uint32_t)__builtin_amdgcn_ds_bpermute(srcl, (int)xoff);
                const uint32_t v_p = (uint32_t)__builtin_amdgcn_ds_bpermute(srcl, (int)thw);
                const bool rowok = (v_p >> 31) == 0;
                const bf16_t* pa = a.dy + (v_dy + (uint32_t)a_coff[i]);
                pa = (rowok & (a_coff[i] >= 0)) ? pa : zero;
                __builtin_amdgcn_global_load_lds(GLB_PTR(pa), LDS_PTR(abase + (i * 4 + wave) * 1024), 16, 0, 0);
                const int t = (int)((v_p >> 20) & 1023) + tp.dt, h = (int)((v_p >> 10) & 1023) + tp.dh, w = (int)(v_p & 1023) + tp.dw;
                const bool ok = rowok & (b_c[i] >= 0) & ((unsigned)t < (unsigned)a.Ts) & ((unsigned)h < (unsigned)a.Hs) & ((unsigned)w < (unsigned)a.Ws);
                const bf16_t* pb = a.src + (v_x + (uint32_t)(tap_delta + b_c[i]));
                pb = ok ? pb : zero;
                __builtin_amdgcn_global_load_lds(GLB_PTR(pb), LDS_PTR(bbase + (i * 4 + wave) * 1024), 16, 0, 0);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            const uint32_t m = mbase + a_row[i];
            const bf16_t* p = zero;
            if (m < (uint32_t)a.M && a_coff[i] >= 0) {
                uint32_t q1 = fd_div(m, a.dWo); const uint32_t wo = m - q1 * a.dWo.d;
                uint32_t q2 = fd_div(q1, a.dHo); const uint32_t ho = q1 - q2 * a.dHo.d;
                uint32_t n = fd_div(q2, a.dTo); const uint32_t to = q2 - n * a.dTo.d;
                const uint32_t off = (((n * a.Td + to * a.dmt) * a.Hd + ho * a.dmh) * a.Wd + wo * a.dmw) * a.Cd + a_coff[i];
                p = a.dy + off;
            }
            __builtin_amdgcn_global_load_lds(GLB_PTR(p), LDS_PTR(abase + (i * 4 + wave) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i) {
            const uint32_t m = mbase + b_row[i];
            const bf16_t* p = zero;
            if (m < (uint32_t)a.M && b_c[i] >= 0) {
                uint32_t q1 = fd_div(m, a.dWo); const uint32_t wo = m - q1 * a.dWo.d;
                uint32_t q2 = fd_div(q1, a.dHo); const uint32_t ho = q1 - q2 * a.dHo.d;
                uint32_t n = fd_div(q2, a.dTo); const uint32_t to = q2 - n * a.dTo.d;
                const int t = (int)to * a.st + tp.dt, h = (int)ho * a.sh + tp.dh, w = (int)wo * a.sw + tp.dw;
                if ((unsigned)t < (unsigned)a.Ts && (unsigned)h < (unsigned)a.Hs && (unsigned)w < (unsigned)a.Ws)
                    p = a.src + (((n * a.Ts + t) * a.Hs + h) * a.Ws + w) * a.Cs + b_c[i];
            }
            __builtin_amdgcn_global_load_lds(GLB_PTR(p), LDS_PTR(bbase + (i * 4 + wave) * 1024), 16, 0, 0);
        }
    };

    f32x16_t acc[TM][TN];
    f32x16_t accb[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) accb[i][r] = 0.f;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }

    // transposing-read addresses: lane = 16 g + 4 r + q  ->  k row (8 * (g >> 1) + r), channel 16 * (g & 1) + 4 q
    const int g16 = lane >> 4, rr = (lane >> 2) & 3, qq = lane & 3;
    const int krow = 8 * (g16 >> 1) + rr;              // + 16 * kstep (+ 4 for the second read)
    int a_col[TM], b_col[TN];                          // channel (element) index of this lane's 4-wide piece inside the tile
#pragma unroll
    for (int i = 0; i < TM; ++i) a_col[i] = wm * (TM * 32) + i * 32 + 16 * (g16 & 1) + 4 * qq;
#pragma unroll
    for (int j = 0; j < TN; ++j) b_col[j] = wn * (TN * 32) + j * 32 + 16 * (g16 & 1) + 4 * qq;

    auto frag_addr = [&](int pitch, int cpr, int row, int col) -> int {
        int chunk = col >> 3;
        if (cpr >= 16) chunk ^= (row & 3) << 2;
        return row * pitch + chunk * 16 + (col & 7) * 2;
    };

    bf16x8_t ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (short)0x3F80;

    const uint32_t smem_base = lds_offset(smem);
    if (c_begin < c_end) {
        stage(c_begin, 0);
        __syncthreads();
        for (int c = c_begin; c < c_end; ++c) {
            const int cur = (c - c_begin) & 1;
            if (c + 1 < c_end) stage(c + 1, cur ^ 1);
            const uint32_t abase = smem_base + cur * STAGE;
            const uint32_t bbase = abase + A_BYTES;
            // fragments of k-step ks+1 are requested before the MFMAs of k-step ks (two register sets, fully unrolled)
            bf16x4_t alo[2][TM], ahi[2][TM], blo[2][TN], bhi[2][TN];
            auto issue = [&](int ks, int set) {
                const int r0 = ks * 16 + krow;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    alo[set][i] = lds_tr16_asm(abase + frag_addr(A_PITCH, A_CPR, r0, a_col[i]));
                    ahi[set][i] = lds_tr16_asm(abase + frag_addr(A_PITCH, A_CPR, r0 + 4, a_col[i]));
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    blo[set][j] = lds_tr16_asm(bbase + frag_addr(B_PITCH, B_CPR, r0, b_col[j]));
                    bhi[set][j] = lds_tr16_asm(bbase + frag_addr(B_PITCH, B_CPR, r0 + 4, b_col[j]));
                }
            };
            issue(0, 0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int set = ks & 1;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                bf16x8_t af[TM], bfr[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    asm volatile("" : "+v"(alo[set][i]), "+v"(ahi[set][i]));
                    af[i] = __builtin_shufflevector(alo[set][i], ahi[set][i], 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    asm volatile("" : "+v"(blo[set][j]), "+v"(bhi[set][j]));
                    bfr[j] = __builtin_shufflevector(blo[set][j], bhi[set][j], 0, 1, 2, 3, 4, 5, 6, 7);
                }
                if (ks < 3) issue(ks + 1, set ^ 1);
#pragma unroll
                for (int i = 0; i < TM; ++i) {
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
                    if (do_bias && wn == 0) accb[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], ones, accb[i], 0, 0, 0);
                }
            }
            __syncthreads();
        }
    }

    // ---- epilogue: fp32 atomics; D row = cout (regs), col = cin (lane & 31) ----
    const int khalf = lane >> 5;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r16 = 0; r16 < 16; ++r16) {
            const int co = co0 + wm * (TM * 32) + i * 32 + (r16 & 3) + 8 * (r16 >> 2) + 4 * khalf;
            if (co >= a.Cout) continue;
            const int sub = co / a.shuf_c, ch = co - sub * a.shuf_c;
            const int co_nat = a.shuf_f > 1 ? ch * a.shuf_f + sub : co;
            float* row = a.dw + co_nat * a.s_cout + tap_i * a.s_tap;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int ci = ci0 + wn * (TN * 32) + j * 32 + (lane & 31);
                if (ci < a.Cin) atomicAdd(row + ci * a.s_cin, acc[i][j][r16]);
            }
            if (do_bias && wn == 0 && (lane & 31) == 0) atomicAdd(a.dbias + co_nat, accb[i][r16]);
        }
    }
}

template <int BM, int BN, int WM, int WN>
static int launch_wgrad(WgradArgs& a, hipStream_t s) {
    constexpr int STAGE = 64 * (BM + BN) * 2;
    const int lds = 2 * STAGE;
    a.tiles_m = cdiv(a.Cout, BM);
    a.tiles_n = cdiv((a.Cin + 7) & ~7, BN);
    const long long base = (long long)a.tiles_m * a.tiles_n * a.ntaps;
    int sk = a.split_k;
    if (sk <= 0) {
        // 2 blocks fit per CU (64 KiB LDS, ~250 VGPRs) -> 512 resident blocks.  Pick the split so that the grid is a whole
        // number of 512-block rounds (a 3.01-round grid wastes a quarter of the chip on its last round).
        const int max_sk = (a.nchunks + 7) / 8;          // keep >= 8 chunks (512 pixels) per block
        sk = 1;
        if (a.ntaps == 1) {
            // 1x1x1 convs stream x and dy exactly once (HBM-bound): every extra split adds a BM x BN fp32 tile of atomics (128->128
            // @16x64x64, B = 8: 0.094 ms at 1024 splits, 0.065 ms at 256); at 64 clips 512 blocks win (step 517.2 -> 515.1 ms)
            static const int pw_blocks = getenv("GENIE_WGRAD_PW1_BLOCKS") ? atoi(getenv("GENIE_WGRAD_PW1_BLOCKS")) : 512;
            const int cand = (int)(pw_blocks / base);
            sk = cand >= 1 ? (cand < max_sk ? cand : max_sk) : 1;
        } else
        for (int rounds = 3; rounds >= 1; --rounds) {
            const int cand = (int)((512ll * rounds) / base);
            if (cand >= 1) { sk = cand < max_sk ? cand : max_sk; break; }
        }
        if (sk > max_sk) sk = max_sk;
        if (sk < 1) sk = 1;
    }
    a.chunks_per_split = cdiv(a.nchunks, sk);
    a.split_k = cdiv(a.nchunks, a.chunks_per_split);
    auto k = wgrad_kernel<BM, BN, WM, WN>;
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            genie_set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return GENIE_ERR_HIP;
        }
        configured = true;
    }
    hipLaunchKernelGGL(k, dim3((unsigned)(base * a.split_k)), dim3(256), lds, s, a);
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}

int genie_conv_wgrad3_try(const GenieWgradDesc* d, hipStream_t s);   // conv_wgrad3.hip
int genie_conv_wgrad_pw_try(const GenieWgradDesc* d, hipStream_t s);  // conv_wgrad_pw.hip

extern "C" int genie_conv_wgrad(const GenieWgradDesc* d, void* stream) {
    GENIE_CHECK_ARG(d, "genie_conv_wgrad: null descriptor");
    GENIE_CHECK_ARG(d->src && d->dy && d->dw && d->taps, "genie_conv_wgrad: null tensor pointer");
    GENIE_CHECK_ARG(d->Cs % 8 == 0 && d->Cd % 8 == 0, "genie_conv_wgrad: channel pitches must be multiples of 8");
    GENIE_CHECK_ARG(d->shuf_c >= 1 && d->shuf_q >= 1 && d->shuf_r >= 1, "genie_conv_wgrad: bad shuffle spec");
    GENIE_CHECK_ARG((long long)d->N * d->Ts * d->Hs * d->Ws * d->Cs < (1ll << 31) && (long long)d->N * d->Td * d->Hd * d->Wd * d->Cd < (1ll << 31),
                    "genie_conv_wgrad: tensor exceeds 2^31 elements");
    {
        int rc = genie_conv_wgrad3_try(d, (hipStream_t)stream);
        if (rc <= 0) return rc;
        GENIE_CHECK_ARG(d->row_px == 0, "genie_conv_wgrad: a W-window (row_px = %d) needs the lean kw-triple kernel (tri_mode, Ws = Wo = Wd = 64, "
                                        "channels >= 64, block range < 2 GiB)", d->row_px);
        GENIE_CHECK_ARG(!d->dy_unshuffled, "genie_conv_wgrad: dy_unshuffled needs the lean kw-triple kernel (tri_mode, W in {8,16,32,64}, H*W %% 64 == 0, "
                                           "channels >= 64); got W=%d H=%d Cin=%d Cout=%d tri_mode=%d", d->Wo, d->Ho, d->Cin, d->Cout, d->tri_mode);
        rc = genie_conv_wgrad_pw_try(d, (hipStream_t)stream);
        if (rc <= 0) return rc;
    }
    WgradArgs a;
    a.src = (const bf16_t*)d->src; a.dy = (const bf16_t*)d->dy; a.dw = d->dw; a.dbias = d->dbias; a.taps = d->taps; a.ntaps = d->ntaps;
    a.N = d->N; a.Ts = d->Ts; a.Hs = d->Hs; a.Ws = d->Ws; a.Cs = d->Cs; a.Cin = d->Cin;
    a.To = d->To; a.Ho = d->Ho; a.Wo = d->Wo; a.st = d->st; a.sh = d->sh; a.sw = d->sw;
    a.Td = d->Td; a.Hd = d->Hd; a.Wd = d->Wd; a.Cd = d->Cd; a.Cout = d->Cout;
    a.dmt = d->dmt; a.dmh = d->dmh; a.dmw = d->dmw;
    const bool shuffled = d->shuf_c < d->Cout;
    if (shuffled) {
        GENIE_CHECK_ARG(d->shuf_c % 8 == 0 && d->Cout % d->shuf_c == 0, "genie_conv_wgrad: shuffled convs need out_channels %% 8 == 0 (got %d)", d->shuf_c);
        a.shuf_c = d->shuf_c; a.shuf_q = d->shuf_q; a.shuf_r = d->shuf_r; a.shuf_f = d->Cout / d->shuf_c;
    } else {
        a.shuf_c = d->Cd > d->Cout ? d->Cd : d->Cout; a.shuf_q = 1; a.shuf_r = 1; a.shuf_f = 1;
    }
    a.s_cout = d->s_cout; a.s_tap = d->s_tap; a.s_cin = d->s_cin;
    const long long M = (long long)d->N * d->To * d->Ho * d->Wo;
    GENIE_CHECK_ARG(M > 0 && M < (1ll << 31), "genie_conv_wgrad: bad row count %lld", M);
    a.M = (int)M;
    a.nchunks = cdiv(M, 64);
    a.split_k = d->split_k;
    a.dWo = make_fastdiv(d->Wo); a.dHo = make_fastdiv(d->Ho); a.dTo = make_fastdiv(d->To);
    GENIE_CHECK_ARG((long long)d->To * d->st < 1024 && (long long)d->Ho * d->sh < 1024 && (long long)d->Wo * d->sw < 1024,
                    "genie_conv_wgrad: output extent * stride must stay below 1024 per axis");
    hipStream_t s = (hipStream_t)stream;
    if (d->Cin <= 32) { genie_note_variant(GENIE_VARIANT_WGRAD_128x32); return launch_wgrad<128, 32, 4, 1>(a, s); }
    if (d->Cout <= 32) { genie_note_variant(GENIE_VARIANT_WGRAD_32x128); return launch_wgrad<32, 128, 1, 4>(a, s); }
    genie_note_variant(GENIE_VARIANT_WGRAD_128);
    return launch_wgrad<128, 128, 2, 2>(a, s);
}
