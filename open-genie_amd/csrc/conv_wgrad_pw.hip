// Weight gradient of pointwise (1x1x1, stride 1) convolutions and Linear layers with >= 256 channels on both sides, on gfx950 MFMA:
//
//   dW[co][ci] += sum_m DY[m][co] * X[m][ci]          (+ dbias[co] += sum_m DY[m][co])
//
// the "TN" product of the vocabulary head (reference genie/dynamics.py:44: Linear(512 -> 2^18), whose weight gradient is 0.8 TFLOP per
// training step over the masked rows) and of the 256 / 512-channel shortcut convolutions of the tokenizer (module/video.py:588-648).
// Both operands are row-major over the reduction index m ("pixel-major"), so neither has the 8 consecutive k per lane an MFMA fragment
// wants.  The generic kernel (conv_wgrad.hip, 128 x 128 tile, two LDS stages) re-reads the DY tile once per 128-column tile of the
// inputs and ran the head at 445 TFLOP/s.  Here:
//   * 256 (co) x 256 (ci) tile, 8 waves as 2 (co) x 4 (ci), a wave owns 128 x 64 (TM = 4, TN = 2: 12 transposing reads per 8 MFMAs);
//   * K chunks of 32 pixels: DY[32][256] + X[32][256] = 32 KB per stage, a ring of FOUR stages issued three chunks ahead by LDS-DMA
//     (each wave: 2 + 2 one-KiB pieces = two 512-byte rows each), counted vmcnt + one raw barrier per chunk, as in gemm_pw256_kernel;
//   * fragments by ds_read_b64_tr_b16 (lane semantics: tests/test_gpu_kernels.py::test_probe_ds_read_tr16); rows are 512 B apart, so
//     the four k-rows of a transposing read would share banks: 16-B chunk c of row r sits at chunk c ^ ((r & 3) << 2) (source-side
//     swizzle of the DMA, which writes lane-linear);
//   * split-K over pixel ranges only when the output has too few tiles to fill the chip (fp32 atomics); a single split adds its tile
//     with plain read-modify-write (every element has one owner);
//   * the bias gradient is summed on the VALU from the DY fragments a wave holds anyway, spread over the four ci-waves by k-step.
#include "common.h"
#include "genie_hip.h"

namespace {

__device__ __attribute__((aligned(256))) uint32_t g_zero_page_wp[64];

struct WgradPwArgs {
    const bf16_t* x;        // [M][Cs]
    const bf16_t* dy;       // [M][Cd]
    float* dw;              // element (co, ci) at co * s_cout + ci * s_cin
    float* dbias;           // [Cout] or null
    int M, Cs, Cd, Cin, Cout;
    long long s_cout, s_cin;
    int tiles_m, tiles_n;   // co tiles, ci tiles
    int nchunks, split_k, chunks_per_split;
};

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

__device__ __forceinline__ bf16x4_t wp_tr16(uint32_t lds_addr) {      // inline asm: outside hipcc's waitcnt bookkeeping (it would drain the DMA ring)
    bf16x4_t v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(lds_addr));
    return v;
}
__device__ __forceinline__ uint32_t wp_lds_offset(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}

__global__ void __launch_bounds__(512) wgrad_pw_kernel(const WgradPwArgs a) {
    constexpr int BM = 256, BN = 256, BK = 32, WN = 4, TM = 4, TN = 2, NSTAGE = 4;
    constexpr int PITCH = 512;                              // bytes per LDS row (256 channels)
    constexpr int A_BYTES = BK * PITCH, STAGE = 2 * A_BYTES;   // 16 KB + 16 KB
    constexpr int NLOAD = 4;                                // DMA pieces per wave and stage: 2 (dy) + 2 (x)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // work id -> (co tile, split, ci tile): the ci tiles of one (co tile, pixel range) are consecutive ids = one XCD at the same time,
    // they share the DY tile through its L2; X is small and stays resident everywhere
    int b;
    {
        const int nb = gridDim.x, bid = blockIdx.x;
        const int q = nb >> 3, r = nb & 7, xcd = bid & 7;
        b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tile_n = b % a.tiles_n; b /= a.tiles_n;
    const int split = b % a.split_k;
    const int tile_m = b / a.split_k;
    const int co0 = tile_m * BM, ci0 = tile_n * BN;
    const bool do_bias = a.dbias != nullptr && tile_n == 0;
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page_wp);
    int c_begin = split * a.chunks_per_split, c_end = c_begin + a.chunks_per_split;
    if (c_end > a.nchunks) c_end = a.nchunks;
    const int nch = c_end - c_begin;

    // ---- staging: piece p = i * 8 + wave (i = 0, 1) holds rows 2 p, 2 p + 1 of the 32-row tile; lane -> row 2 p + (lane >> 5),
    //      LDS chunk lane & 31 = logical chunk (lane & 31) ^ ((row & 3) << 2) ----
    int s_row[2], a_c[2], b_c[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 2 * (i * 8 + wave) + (lane >> 5);
        const int lc = (lane & 31) ^ ((row & 3) << 2);
        s_row[i] = row;
        a_c[i] = co0 + lc * 8 < a.Cout ? co0 + lc * 8 : -1;
        b_c[i] = (ci0 + lc * 8 < a.Cs && ci0 + lc * 8 < ((a.Cin + 7) & ~7)) ? ci0 + lc * 8 : -1;
    }
    int next_chunk = c_begin;
    auto issue = [&](int slot, bool live) {
        char* abuf = smem + slot * STAGE;
        char* bbuf = abuf + A_BYTES;
        const long long mbase = (long long)next_chunk * BK;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const long long m = mbase + s_row[i];
            const bool rowok = live && m < a.M;
            const bf16_t* q = (rowok && a_c[i] >= 0) ? a.dy + m * a.Cd + a_c[i] : zero;
            __builtin_amdgcn_global_load_lds(GLB_PTR(q), LDS_PTR(abuf + (i * 8 + wave) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const long long m = mbase + s_row[i];
            const bool rowok = live && m < a.M;
            const bf16_t* q = (rowok && b_c[i] >= 0) ? a.x + m * a.Cs + b_c[i] : zero;
            __builtin_amdgcn_global_load_lds(GLB_PTR(q), LDS_PTR(bbuf + (i * 8 + wave) * 1024), 16, 0, 0);
        }
        ++next_chunk;
    };

    // ---- transposing-read addresses: lane = 16 g + 4 r + q reads k-row 8 (g >> 1) + r (+ 16 kstep, + 4 for the second read),
    //      channels 16 (g & 1) + 4 q .. + 3 of a 32-wide MFMA tile; swizzle key = k-row & 3 = r for both reads ----
    const int g16 = lane >> 4, rr = (lane >> 2) & 3, qq = lane & 3;
    const int krow = 8 * (g16 >> 1) + rr;
    uint32_t a_off[TM], b_off[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int col = wm * (TM * 32) + i * 32 + 16 * (g16 & 1) + 4 * qq;
        a_off[i] = (uint32_t)(krow * PITCH + (((col >> 3) ^ (rr << 2)) << 4) + (col & 7) * 2);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = wn * (TN * 32) + j * 32 + 16 * (g16 & 1) + 4 * qq;
        b_off[j] = (uint32_t)(A_BYTES + krow * PITCH + (((col >> 3) ^ (rr << 2)) << 4) + (col & 7) * 2);
    }

    f32x16_t acc[TM][TN];
    float bsum[TM];                                         // bias gradient: this lane's co = 32 i + (lane & 31), its 8 k of the k-steps it takes
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        bsum[i] = 0.f;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }

    const uint32_t smem_base = wp_lds_offset(smem);
    if (nch > 0) {
        issue(0, true);
        issue(1, nch > 1);
        issue(2, nch > 2);
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NLOAD) : "memory");     // chunk 0 landed, two stages in flight
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        int slot = 0;
        for (int c = 0; c < nch; ++c) {
            issue((slot + 3) & 3, c + 3 < nch);              // the stage consumed one chunk ago (every wave is past that barrier)
            __builtin_amdgcn_sched_barrier(0);
            const uint32_t st = smem_base + slot * STAGE;
            bf16x4_t alo[2][TM], ahi[2][TM], blo[2][TN], bhi[2][TN];
            auto reads = [&](int ks, int set) {
                const uint32_t ko = (uint32_t)(ks * 16 * PITCH);
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    alo[set][i] = wp_tr16(st + a_off[i] + ko);
                    ahi[set][i] = wp_tr16(st + a_off[i] + ko + 4 * PITCH);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    blo[set][j] = wp_tr16(st + b_off[j] + ko);
                    bhi[set][j] = wp_tr16(st + b_off[j] + ko + 4 * PITCH);
                }
            };
            reads(0, 0);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
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
                if (ks == 0) reads(1, 1);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
                // bias gradient: the four ci-waves of a co row hold the same DY fragments; k-step (2 c + ks) is summed by ci-wave
                // (2 c + ks) & 3 on the VALU (an MFMA against ones would cost 64 more accumulator registers)
                if (do_bias && ((2 * c + ks) & 3) == wn) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const u32x4_t w = __builtin_bit_cast(u32x4_t, af[i]);
                        float t = 0.f;
#pragma unroll
                        for (int e = 0; e < 4; ++e) t += __uint_as_float(w[e] << 16) + __uint_as_float(w[e] & 0xffff0000u);
                        bsum[i] += t;
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NLOAD) : "memory");   // chunk c + 1 landed; the two newest stages stay in flight
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            slot = (slot + 1) & 3;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }

    // ---- epilogue: D row = co (registers), col = ci (lane & 31): 128 contiguous bytes per register and 32-lane half ----
    const int khalf = lane >> 5;
    const bool atomic = a.split_k > 1;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r16 = 0; r16 < 16; ++r16) {
            const int co = co0 + wm * (TM * 32) + i * 32 + (r16 & 3) + 8 * (r16 >> 2) + 4 * khalf;
            if (co >= a.Cout) continue;
            float* row = a.dw + co * a.s_cout;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int ci = ci0 + wn * (TN * 32) + j * 32 + (lane & 31);
                if (ci < a.Cin) {
                    float* o = row + ci * a.s_cin;
                    if (atomic) atomicAdd(o, acc[i][j][r16]);
                    else *o += acc[i][j][r16];
                }
            }
        }
    }
    if (do_bias) {                                          // lanes l and l + 32 hold the two k halves of the same co
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int co = co0 + wm * (TM * 32) + i * 32 + (lane & 31);
            if (co < a.Cout) atomicAdd(a.dbias + co, bsum[i]);
        }
    }
}

}  // namespace

// Returns 1 when the problem is not eligible (the caller falls back to the generic kernel), 0 on launch, < 0 on error.
int genie_conv_wgrad_pw_try(const GenieWgradDesc* d, hipStream_t s) {
    static const int on = getenv("GENIE_WGRAD_PW") ? atoi(getenv("GENIE_WGRAD_PW")) : 1;
    if (!on || !d->pointwise || d->ntaps != 1 || d->Cin < 256 || d->Cout < 256) return 1;
    if (d->st != 1 || d->sh != 1 || d->sw != 1 || d->To != d->Ts || d->Ho != d->Hs || d->Wo != d->Ws) return 1;
    if (d->Td != d->To || d->Hd != d->Ho || d->Wd != d->Wo || d->dmt != 1 || d->dmh != 1 || d->dmw != 1) return 1;
    if (d->shuf_c < d->Cout) return 1;                                       // depth-to-space outputs: generic kernel
    const long long M = (long long)d->N * d->To * d->Ho * d->Wo;
    if (M < 256 || M >= (1ll << 31)) return 1;
    WgradPwArgs a;
    a.x = (const bf16_t*)d->src; a.dy = (const bf16_t*)d->dy; a.dw = d->dw; a.dbias = d->dbias;
    a.M = (int)M; a.Cs = d->Cs; a.Cd = d->Cd; a.Cin = d->Cin; a.Cout = d->Cout;
    a.s_cout = d->s_cout; a.s_cin = d->s_cin;
    a.tiles_m = cdiv(d->Cout, 256);
    a.tiles_n = cdiv((d->Cin + 7) & ~7, 256);
    a.nchunks = cdiv(M, 32);
    const long long tiles = (long long)a.tiles_m * a.tiles_n;
    if (tiles >= (1ll << 24)) return 1;
    // Few output tiles (the 256 / 512-channel shortcut convolutions of the tokenizer: 1 - 4 tiles, K = half a million pixels) are pure
    // streaming with split-K atomics; there the generic kernel's two blocks per CU keep more loads in flight (same-box A/B of the
    // tokenizer step: 277.9 ms generic, 279.3 ms here).  pointwise = 2 forces this kernel (tests).
    if (tiles < 192 && d->pointwise != 2) return 1;
    int sk = d->split_k;
    if (sk <= 0) {
        sk = 1;
        if (tiles < 192) {                                                   // few output tiles: split the pixels, >= 16 chunks (512 pixels) each
            sk = (int)(256 / tiles);
            const int max_sk = a.nchunks / 16 > 1 ? a.nchunks / 16 : 1;
            if (sk > max_sk) sk = max_sk;
            if (sk < 1) sk = 1;
        }
    }
    a.chunks_per_split = cdiv(a.nchunks, sk);
    a.split_k = cdiv(a.nchunks, a.chunks_per_split);
    constexpr int lds = 4 * 2 * 32 * 512;
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute((const void*)wgrad_pw_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            genie_set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return GENIE_ERR_HIP;
        }
        configured = true;
    }
    genie_note_variant(GENIE_VARIANT_WGRAD_PW);
    hipLaunchKernelGGL(wgrad_pw_kernel, dim3((unsigned)(tiles * a.split_k)), dim3(512), lds, s, a);
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}
