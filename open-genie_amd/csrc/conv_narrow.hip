// HBM-bound CausalConv3d: the stem of the tokenizer, Conv3d(3 -> 128, k = 3) -- reference genie/module/video.py:154-192 as used by
// MAGVIT2_ENC_DESC[0] (genie/tokenizer.py:25) -- and the backward-data pass of the decoder's head conv (128 -> 3), which is the same
// computation with the weights flipped: a <= 4-channel input, 27 taps, 128 output channels, one 256-B row written per pixel.
// 0.39 MB in, 16.8 MB out per 16x64x64 clip: the roofline is the output stream (17.19 MB per clip), the arithmetic
// (1.36 GFLOP per clip) only has to keep out of its way.
//
// The generic gather-GEMM pads every tap to 8 channels (K = 216) and stores the MFMA C layout with 8-byte pieces into 32 different
// rows per instruction: 17 % of the HBM peak.  Here
//   * K is packed to 27 taps x 4 channels = 108 (+ 2 slots that carry the bias as a bf16 hi/lo pair against a constant 1.0,
//     + 2 zero slots) = 7 MFMA k-steps of 16;
//   * the product is formed TRANSPOSED, C[cout][pixel] = W[cout][k] . X^T[k][pixel]: the weights are the A operand and live in
//     registers for the whole launch (7 x 4 fragments), a lane owns a pixel, and its B fragment is two 8-byte LDS reads from a
//     (3 frames x 6 rows x (W + 2) pixels x 4 channels) image of the input tile with explicit zero borders -- no masks;
//   * in that layout a lane holds 4 consecutive output channels of its pixel per accumulator quad, so the 32 x 128 wave tile goes
//     through LDS with 8-byte writes and comes back as whole 16-byte chunks, 16 lanes per 256-byte pixel row: every store
//     instruction writes 1 KiB of contiguous HBM.
// One workgroup = 4 waves = 4 image rows of one frame; workgroups are persistent (weights are fetched once per workgroup).
#include "common.h"
#include "genie_hip.h"

namespace {

constexpr int NIN_HB = 4;                  // image rows per workgroup (one per wave)
constexpr int NIN_KSTEPS = 7;              // 112 = 27 taps x 4 channels + bias hi/lo + 2 zero slots
constexpr int NIN_KP = 16 * NIN_KSTEPS;    // row pitch of the weight pack [128][112]
constexpr int NIN_OPITCH = 272;            // bytes per staged output pixel row (256 + 16: spreads the 8-byte writes over the banks)

struct NarrowInArgs {
    const bf16_t* src;      // CL [N][T][H][W][cs], cs >= 4 channels per pixel (only the first 4 are read)
    const bf16_t* wpack;    // [128][112]
    bf16_t* dst;            // CL [N][T][H][W][cd], cd >= 128
    int N, T, H, W, cs, cd;
    int t_lo;               // frame offset of the first tap plane: -2 for the causal forward, 0 for its backward-data pass
    int ntiles;             // N * T * ceil(H / 4)
    int hblocks;            // ceil(H / 4)
};

template <int W>
__global__ void __launch_bounds__(256, 2) conv_narrow_in_kernel(const NarrowInArgs a) {
    constexpr int WP = W + 2, ROWS = NIN_HB + 2;
    constexpr int IMG_PIX = 3 * ROWS * WP;                       // staged input pixels, 8 bytes each
    constexpr int IMG_BYTES = (IMG_PIX * 8 + 15) & ~15;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const img = smem;
    char* const stage = smem + IMG_BYTES;                        // 4 waves x 32 pixels x NIN_OPITCH

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = lane >> 5, px = lane & 31;

    // ---- weights: A fragments for (k-step j, cout tile ct): row = ct * 32 + px, k = 16 j + 8 kh .. + 7 ----
    bf16x8_t wf[NIN_KSTEPS][4];
#pragma unroll
    for (int j = 0; j < NIN_KSTEPS; ++j)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
            wf[j][ct] = *reinterpret_cast<const bf16x8_t*>(a.wpack + (ct * 32 + px) * NIN_KP + 16 * j + 8 * kh);

    // ---- per-lane LDS offsets of the two taps of every k-step: tap = 4 j + 2 kh (+ 1), tap = (f * 3 + r) * 3 + c ----
    int off_a[NIN_KSTEPS], off_b[NIN_KSTEPS];
#pragma unroll
    for (int j = 0; j < NIN_KSTEPS; ++j) {
        const int ta = 4 * j + 2 * kh, tb = ta + 1;
        const int fa = ta / 9, ra = (ta / 3) % 3, ca = ta % 3;
        const int fb = tb / 9, rb = (tb / 3) % 3, cb = tb % 3;
        off_a[j] = ((fa * ROWS + ra) * WP + ca) * 8;
        off_b[j] = tb < 27 ? ((fb * ROWS + rb) * WP + cb) * 8 : -1;          // tap 27 is the bias slot (a constant fragment)
    }
    // bias slot: k = 108, 109 multiply 1.0 (bf16 0x3F80), k = 110, 111 are zero
    bf16x4_t ones;
    ones[0] = 0x3F80; ones[1] = 0x3F80; ones[2] = 0; ones[3] = 0;

    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const int hb = tile % a.hblocks;
        const int t = (tile / a.hblocks) % a.T;
        const int n = tile / (a.hblocks * a.T);
        const int h0 = hb * NIN_HB;
        __syncthreads();                                           // the previous tile's image is no longer read
        // ---- stage the input image: frames t + t_lo .. + 2, rows h0 - 1 .. h0 + 4, columns -1 .. W ----
        for (int i = tid; i < IMG_PIX; i += 256) {
            const int c = i % WP, r = (i / WP) % ROWS, f = i / (WP * ROWS);
            const int tt = t + a.t_lo + f, hh = h0 - 1 + r, ww = c - 1;
            u32x2_t v = {0u, 0u};
            if ((unsigned)tt < (unsigned)a.T && (unsigned)hh < (unsigned)a.H && (unsigned)ww < (unsigned)W)
                v = *reinterpret_cast<const u32x2_t*>(a.src + ((((long long)n * a.T + tt) * a.H + hh) * W + ww) * a.cs);
            *reinterpret_cast<u32x2_t*>(img + i * 8) = v;
        }
        __syncthreads();
        const int h = h0 + wave;
        if (h >= a.H) continue;                                    // wave-uniform (partial last row block); barriers are at the loop top
        char* const st = stage + wave * 32 * NIN_OPITCH;
        const long long orow = (((long long)n * a.T + t) * a.H + h) * W;
#pragma unroll 1
        for (int mt = 0; mt < W / 32; ++mt) {
            const char* base = img + ((wave * WP) + mt * 32 + px) * 8;        // image position of (row h - 1, column w - 1) of tap (f = 0, r = 0, c = 0)
            f32x16_t acc[4];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
#pragma unroll
            for (int j = 0; j < NIN_KSTEPS; ++j) {
                const bf16x4_t lo = *reinterpret_cast<const bf16x4_t*>(base + off_a[j]);
                const bf16x4_t hi = off_b[j] >= 0 ? *reinterpret_cast<const bf16x4_t*>(base + off_b[j]) : ones;
                bf16x8_t xf;
                xf[0] = lo[0]; xf[1] = lo[1]; xf[2] = lo[2]; xf[3] = lo[3];
                xf[4] = hi[0]; xf[5] = hi[1]; xf[6] = hi[2]; xf[7] = hi[3];
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j][ct], xf, acc[ct], 0, 0, 0);
            }
            // ---- C[cout][pixel] -> bf16 -> LDS [pixel][cout] (8-byte writes) -> 16-byte chunks, 16 lanes per pixel row ----
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    u32x2_t v;
                    v[0] = pack_bf16x2(acc[ct][4 * g], acc[ct][4 * g + 1]);
                    v[1] = pack_bf16x2(acc[ct][4 * g + 2], acc[ct][4 * g + 3]);
                    *reinterpret_cast<u32x2_t*>(st + px * NIN_OPITCH + (ct * 32 + g * 8 + kh * 4) * 2) = v;
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the wave's own LDS writes have landed
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int idx = it * 64 + lane, p = idx >> 4, ch = idx & 15;
                const u32x4_t v = *reinterpret_cast<const u32x4_t*>(st + p * NIN_OPITCH + ch * 16);
                *reinterpret_cast<u32x4_t*>(a.dst + (orow + mt * 32 + p) * a.cd + ch * 8) = v;
            }
            __builtin_amdgcn_wave_barrier();                       // the next m-tile overwrites the staging rows
        }
    }
}

}  // namespace

extern "C" int genie_conv_narrow_in(const void* src_cl, int src_pitch, const void* wpack, void* dst_cl, int dst_pitch, int N, int T, int H, int W,
                                    int t_lo, void* stream) {
    GENIE_CHECK_ARG(src_cl && wpack && dst_cl, "genie_conv_narrow_in: null pointer");
    GENIE_CHECK_ARG(W == 16 || W == 32 || W == 64 || W == 128, "genie_conv_narrow_in: image width %d not in {16, 32, 64, 128}", W);
    GENIE_CHECK_ARG(src_pitch >= 4 && src_pitch % 4 == 0 && dst_pitch >= 128 && dst_pitch % 8 == 0, "genie_conv_narrow_in: bad channel pitch (%d in, %d out)", src_pitch, dst_pitch);
    GENIE_CHECK_ARG(N >= 1 && T >= 1 && H >= 1 && (long long)N * T * H * W * (long long)dst_pitch < (1ll << 40), "genie_conv_narrow_in: bad geometry");
    NarrowInArgs a;
    a.src = (const bf16_t*)src_cl; a.wpack = (const bf16_t*)wpack; a.dst = (bf16_t*)dst_cl;
    a.N = N; a.T = T; a.H = H; a.W = W; a.cs = src_pitch; a.cd = dst_pitch; a.t_lo = t_lo;
    a.hblocks = (H + NIN_HB - 1) / NIN_HB;
    const long long tiles = (long long)N * T * a.hblocks;
    GENIE_CHECK_ARG(tiles < (1ll << 31), "genie_conv_narrow_in: too many tiles");
    a.ntiles = (int)tiles;
    const int grid = a.ntiles < 512 ? a.ntiles : 512;            // two workgroups per CU, each walks its share of the tiles
    hipStream_t s = (hipStream_t)stream;
#define GENIE_NIN(Wv)                                                                                          \
    do {                                                                                                       \
        const int lds = ((3 * (NIN_HB + 2) * (Wv + 2) * 8 + 15) & ~15) + 4 * 32 * NIN_OPITCH;                   \
        conv_narrow_in_kernel<Wv><<<grid, 256, lds, s>>>(a);                                                   \
    } while (0)
    if (W == 16) { GENIE_CHECK_ARG(false, "genie_conv_narrow_in: W = 16 needs 32-pixel row tiles"); }
    else if (W == 32) GENIE_NIN(32);
    else if (W == 64) GENIE_NIN(64);
    else GENIE_NIN(128);
#undef GENIE_NIN
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}
