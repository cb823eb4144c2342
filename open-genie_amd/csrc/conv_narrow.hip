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

__device__ __attribute__((aligned(256))) uint32_t g_zero_page_n[64];

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

    // The input image of a tile (frames t + t_lo .. + 2, rows h0 - 1 .. h0 + 4, columns -1 .. W; 8 B per pixel) is REGISTER-staged one tile
    // ahead: its loads are issued before the current tile's MFMAs / stores and written to LDS at the top of the next iteration.  (Round 2
    // loaded it at the top of its own tile: every tile began with an exposed global round trip, ~40 % of a tile's time at 64 clips, which
    // only the second co-resident workgroup covered -- 3.97 TB/s on the 1.1-GB output stream.)
    constexpr int IMG_LD = (IMG_PIX + 255) / 256;
    constexpr bool PREFETCH = W <= 64;                             // W = 128: 10 loads per thread would spill (weights hold 112 VGPRs); load at the top
    u32x2_t ireg[IMG_LD];
    auto load_image = [&](int tile) {
        const int hb = tile % a.hblocks;
        const int t = (tile / a.hblocks) % a.T;
        const int n = tile / (a.hblocks * a.T);
        const int h0 = hb * NIN_HB;
#pragma unroll
        for (int k = 0; k < IMG_LD; ++k) {
            const int i = k * 256 + tid;
            const int c = i % WP, r = (i / WP) % ROWS, f = i / (WP * ROWS);
            const int tt = t + a.t_lo + f, hh = h0 - 1 + r, ww = c - 1;
            u32x2_t v = {0u, 0u};
            if (i < IMG_PIX && (unsigned)tt < (unsigned)a.T && (unsigned)hh < (unsigned)a.H && (unsigned)ww < (unsigned)W)
                v = *reinterpret_cast<const u32x2_t*>(a.src + ((((long long)n * a.T + tt) * a.H + hh) * W + ww) * a.cs);
            ireg[k] = v;
        }
    };
    if (PREFETCH && (int)blockIdx.x < a.ntiles) load_image(blockIdx.x);
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const int hb = tile % a.hblocks;
        const int t = (tile / a.hblocks) % a.T;
        const int n = tile / (a.hblocks * a.T);
        const int h0 = hb * NIN_HB;
        __syncthreads();                                           // the previous tile's image is no longer read
        if (!PREFETCH) load_image(tile);
#pragma unroll
        for (int k = 0; k < IMG_LD; ++k) {
            const int i = k * 256 + tid;
            if (i < IMG_PIX) *reinterpret_cast<u32x2_t*>(img + i * 8) = ireg[k];
        }
        __syncthreads();
        if (PREFETCH && tile + (int)gridDim.x < a.ntiles) load_image(tile + gridDim.x);      // in flight under this tile's MFMAs and stores
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


// ------------------------------------------------------------------------------------------------------------------------------
// The other direction: 128 input channels -> <= 3 output channels, 3x3x3, stride 1 -- the forward of the decoder's head conv
// CausalConv3d(128 -> 3) (tokenizer.py:172).  16.8 MB in, 0.39 MB out per clip: the roofline is the INPUT stream.
//
// A 3-column GEMM wastes a 32-wide MFMA tile ten times over.  Here the product is transposed and only the frame taps stay apart:
//     C[(dt, co)][q] = sum_{dh, dw, ci} W[co][(dt, dh, dw)][ci] * X[tin][q + (dh - 1, dw - 1)][ci]      (K = 9 * 128 = 1152, 36 k-steps of 32)
// is formed with v_mfma_f32_16x16x32_bf16 for 16 output pixels q of one image row and ONE input frame tin: 9 of the 16 rows are
// used (row = 4 dt + co), the weights are the A operand and live in registers for the whole launch (36 x 4 VGPRs).  out[t] is the
// sum of C_{t + t_lo + dt}[dt] over three consecutive input frames.  In the 16x16 C layout row block dt is lane group dt (lanes
// 16 dt .. 16 dt + 15, registers = co), so that sum is carried in the accumulator itself: after each input frame lane group 2
// holds a finished output frame (converted, bias added and stored as whole 16-byte pixels straight from registers), and the
// partial sums move one lane group up (ds_bpermute, 8 per frame) to be the C input of the next frame's MFMAs.  No scatter, no
// atomics (an earlier input-stationary version spent 3/4 of its time in ds_add_f32), no fp32 staging.
// One workgroup = 4 waves = 4 image rows x 32 columns of one sample, marching over the frames of its segment; the input tile of a
// frame (6 rows x 34 pixels x 256 B, 16-byte chunks XOR-swizzled by pixel so the 256-B-strided fragment reads are conflict-free)
// is shared by the waves and arrives by LDS-DMA two frames ahead (3 buffers, counted waits, one s_barrier per frame); pixels
// outside the image or the clip are fetched from a zero page, so there are no masks in the inner loop.
// ------------------------------------------------------------------------------------------------------------------------------
constexpr int NOUT_KSTEPS = 36;
constexpr int NOUT_ROWPIX = 34;                         // 32 columns + 2 halo
constexpr int NOUT_PIECES = 13;                         // DMA pieces (4 pixels x 256 B) per wave and frame: 4 x 13 x 4 = 208 >= 6 x 34 pixels
constexpr int NOUT_BUF_BYTES = 4 * NOUT_PIECES * 1024;  // 53248
constexpr int NOUT_NBUF = 3;
constexpr int NOUT_WROW = 1152;                         // weight pack [16][1152]

__device__ __forceinline__ unsigned nout_lds_offset(const void* p) {
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
__device__ __forceinline__ bf16x8_t nout_read128(unsigned lds_addr) {
    bf16x8_t v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(lds_addr));
    return v;
}
__device__ __forceinline__ bf16x8_t nout_read128_hi(unsigned lds_addr) {      // the second 16-column half: + 16 pixels
    bf16x8_t v;
    asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(v) : "v"(lds_addr));
    return v;
}

struct NarrowOutArgs {
    const bf16_t* src;      // CL [N][T][H][W][128]
    const bf16_t* wpack;    // [16][1152]: row = 4 dt + co (other rows zero), k = (dh * 3 + dw) * 128 + ci
    const float* bias;      // [3] or null
    bf16_t* dst;            // CL [N][T][H][W][8]
    int N, T, H, W, cout;
    int t_lo;               // out[t] reads input frames t + t_lo .. t + t_lo + 2
    int hblocks, wblocks, tsegs, tseg_len;
};

__global__ void __launch_bounds__(256) conv_narrow_out_kernel(const NarrowOutArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, kg = lane >> 4;
    const int W = a.W, H = a.H;
    const unsigned lds0 = nout_lds_offset(smem);

    int b = blockIdx.x;
    const int seg = b % a.tsegs; b /= a.tsegs;
    const int wb = b % a.wblocks; b /= a.wblocks;
    const int hb = b % a.hblocks;
    const int n = b / a.hblocks;
    const int h0 = hb * 4, w0 = wb * 32;
    const int h = h0 + wave;
    const bool row_ok = h < H;                                                 // wave-uniform
    const int ts0 = seg * a.tseg_len, ts1 = (ts0 + a.tseg_len < a.T) ? ts0 + a.tseg_len : a.T;      // output frames of this workgroup
    const int nf = ts1 - ts0 + 2;                                              // input frames tin = ts0 + t_lo + f, f = 0 .. nf - 1
    const int tin_first = ts0 + a.t_lo;

    // weights: A fragments, row = lane & 15, k = 32 ks + 8 kg .. + 7
    bf16x8_t wf[NOUT_KSTEPS];
#pragma unroll
    for (int ks = 0; ks < NOUT_KSTEPS; ++ks) wf[ks] = *reinterpret_cast<const bf16x8_t*>(a.wpack + col * NOUT_WROW + 32 * ks + 8 * kg);
    float bias_r[3] = {0.f, 0.f, 0.f};
    if (a.bias) {
#pragma unroll
        for (int c = 0; c < 3; ++c) if (c < a.cout) bias_r[c] = a.bias[c];
    }
    // pin the completion of these loads HERE: hipcc would otherwise wait for them at their first uses inside the frame loop with
    // counted vmcnt(N) that also count (and drain) the inline-asm DMA stream
#pragma unroll
    for (int c = 0; c < 3; ++c) asm volatile("" : "+v"(bias_r[c]));
#pragma unroll
    for (int ks = 0; ks < NOUT_KSTEPS; ++ks) asm volatile("" : "+v"(wf[ks]));

    // ---- lane constants ----
    // DMA piece P = wave + 4 i covers tile pixels 4 P .. 4 P + 3 (pixel q = row * 34 + column, 6 rows: image rows h0 - 1 .. h0 + 4,
    // columns w0 - 1 .. w0 + 32); lane -> pixel q = 4 P + kg, LDS slot col holds the source chunk col ^ (q & 15)
    unsigned dma_off[NOUT_PIECES];                                             // byte offset inside the input frame, or ~0u: zero page
#pragma unroll
    for (int i = 0; i < NOUT_PIECES; ++i) {
        const int q = 4 * (wave + 4 * i) + kg;
        const int rr = q / NOUT_ROWPIX, cc = q - rr * NOUT_ROWPIX;
        const int hin = h0 - 1 + rr, w = w0 - 1 + cc;
        const bool ok = rr < 6 && hin >= 0 && hin < H && w >= 0 && w < W;
        dma_off[i] = ok ? (unsigned)((hin * W + w) * 256 + ((col ^ (q & 15)) << 4)) : 0xffffffffu;
    }
    const char* const zero_ptr = reinterpret_cast<const char*>(g_zero_page_n) + col * 16;
    // fragment reads: B operand column = output pixel col (+ 16 for the second half), k-group kg; tap (dh, dw) reads tile pixel
    // q = (wave + dh) * 34 + col + dw, chunk (4 j + kg) ^ (q & 15) for the k-step j = 0..3 inside the tap
    unsigned rd_base[9];
#pragma unroll
    for (int dh = 0; dh < 3; ++dh)
#pragma unroll
        for (int dw = 0; dw < 3; ++dw) {
            const int q = (wave + dh) * NOUT_ROWPIX + col + dw;
            rd_base[dh * 3 + dw] = (unsigned)(q * 256 + ((kg ^ (q & 15)) << 4));
        }
    const unsigned perm_addr = (unsigned)(((lane - 16) & 63) * 4);

    auto issue_frame = [&](int f) {
        const int tin = tin_first + f;
        const bool fv = f < nf && tin >= 0 && tin < a.T;                       // wave-uniform
        const char* base = reinterpret_cast<const char*>(a.src) + (((long long)n * a.T + (fv ? tin : 0)) * H * W) * 256;
        const unsigned buf = lds0 + (unsigned)(f % NOUT_NBUF) * NOUT_BUF_BYTES + wave * 1024;
#pragma unroll
        for (int i = 0; i < NOUT_PIECES; ++i) {
            const char* p = (fv && dma_off[i] != 0xffffffffu) ? base + dma_off[i] : zero_ptr;
            const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(buf + i * 4096));
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(p), "s"(dst) : "memory", "m0");
        }
    };

    issue_frame(0);
    issue_frame(1);
    f32x4_t acc[2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) acc[hf] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    u32x4_t pend[2] = {u32x4_t{0u, 0u, 0u, 0u}, u32x4_t{0u, 0u, 0u, 0u}};
    int pend_t = -1;
    auto flush = [&]() {
        if (pend_t >= 0 && kg == 2) {
            bf16_t* o = a.dst + ((((long long)n * a.T + pend_t) * H + h) * W + w0 + col) * 8;
            *reinterpret_cast<u32x4_t*>(o) = pend[0];
            *reinterpret_cast<u32x4_t*>(o + 128) = pend[1];
        }
        pend_t = -1;
    };

    for (int f = 0; f < nf; ++f) {
        asm volatile("s_waitcnt vmcnt(13)" ::: "memory");                     // this wave's pieces of frame f landed (frame f + 1 stays in flight)
        __builtin_amdgcn_s_barrier();                                          // ... and everyone's; everyone is done with frame f - 1
        asm volatile("" ::: "memory");
        flush();                                                               // stores go out BEFORE the next DMA batch: the counted wait above
        issue_frame(f + 2);                                                    // then covers exactly one batch (into the buffer frame f - 1 left)
        const int tin = tin_first + f;
        if (row_ok && tin >= 0 && tin < a.T) {
            const unsigned buf = lds0 + (unsigned)(f % NOUT_NBUF) * NOUT_BUF_BYTES;
            // Fragment reads are inline asm: hipcc orders a compiler-visible LDS access behind EVERY outstanding LDS-DMA (vmcnt(0)),
            // which would drain the prefetch once per frame.  Groups of four reads (two k-steps x two halves), one group ahead.
            bf16x8_t xg[2][4];
            auto read_group = [&](int g, bf16x8_t* x) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int ks = 2 * g + q;
                    const unsigned ad = buf + (rd_base[ks >> 2] ^ (unsigned)((ks & 3) << 6));
                    x[2 * q] = nout_read128(ad);
                    x[2 * q + 1] = nout_read128_hi(ad);
                }
            };
            read_group(0, xg[0]);
#pragma unroll
            for (int g = 0; g < NOUT_KSTEPS / 2; ++g) {
                if (g + 1 < NOUT_KSTEPS / 2) {
                    read_group(g + 1, xg[(g + 1) & 1]);
                    asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                asm volatile("" : "+v"(xg[g & 1][0]), "+v"(xg[g & 1][1]), "+v"(xg[g & 1][2]), "+v"(xg[g & 1][3]));
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[2 * g + q], xg[g & 1][2 * q], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[2 * g + q], xg[g & 1][2 * q + 1], acc[1], 0, 0, 0);
                }
            }
        }
        // lane group 2 now holds out[ts0 + f - 2] (its dt = 2 term was the last one): park it for the store after the next barrier
        if (f >= 2 && row_ok) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                float o[8] = {acc[hf][0] + bias_r[0], acc[hf][1] + bias_r[1], acc[hf][2] + bias_r[2], 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < 3; ++c) if (c >= a.cout) o[c] = 0.f;
                pend[hf] = pack8(o);
            }
            pend_t = ts0 + f - 2;
        }
        // partial sums move one lane group up: dt -> dt + 1 for the next input frame; group 0 starts a new output frame
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");                      // MFMA results -> inline-asm reader: explicit wait states
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v;
                asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(v) : "v"(perm_addr), "v"(acc[hf][r]));
                acc[hf][r] = v;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                asm volatile("" : "+v"(acc[hf][r]));
                if (kg == 0) acc[hf][r] = 0.f;
            }
    }
    flush();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}


// ------------------------------------------------------------------------------------------------------------------------------
// Head conv forward, second cut (round 5) -- images up to 64 pixels wide.
// The kernel above is bound by the LDS port: per 32 pixels and input frame a wave issues 144 ds_read_b128 (one per 16x16x32 MFMA
// and column half, 9 of 16 MFMA rows used) -- 2304 LDS cycles per 32-KB slab and CU, which is the time HBM needs to deliver it.
// Here the column tap joins the frame tap in the M dimension:
//     C[(dt, dw, co)][x] = sum_{dh, ci} W[co][(dt, dh, dw)][ci] * X[tin][h + dh - 1][x][ci]          (27 of 32 rows, K = 3 * 128 = 384)
// with x the INPUT column, one v_mfma_f32_32x32x16_bf16 per 16 channels, row tap and 32 columns: 24 MFMAs and 24 fragment reads
// per 32 pixels and frame (was 72 and 144).  The output is out[t][h][p] = sum_dw C_(dt = 2, accumulated)[dw][p + dw - 1]: a wave
// owns a whole image row (two 32-column tiles for W = 64, one for W = 32), so the dw shift is a LANE shift of the finished
// accumulator rows with zeros entering at both ends = the conv's zero padding; no halo columns.
//   * rows: row = 8 dt + s for s = 3 dw + co < 8, row = 24 + dt for (dw, co) = (2, 2).  In the 32x32 C layout (row = 8 (i / 4) + 4 (lane / 32)
//     + i % 4) the frame tap dt is then the register quad i / 4: carrying partial sums to the next input frame (dt -> dt + 1) is a
//     register move, and the finished quad dt = 2 holds slots s = 0..3 in lanes 0-31 and s = 4..7 in lanes 32-63.  One
//     v_permlane32_swap per register between the two column tiles turns that into "lane = column 0..63" for every slot, and
//     the shift is one DPP wave_shr / wave_shl (bound_ctrl: zero fill) per outer column tap and channel.
//   * the input tile of a frame (6 rows x W pixels x 256 B) is consumed in four channel quarters: a stage is 6 x W x 64 B (24 KB),
//     three stages in LDS (72 KB: TWO workgroups per CU -- one's barriers and epilogues are covered by the other), DMA'd two stages
//     ahead, 16-byte chunks XOR-swizzled with bits 2-3 of the pixel (conflict-free ds_read_b128 at a 64-byte pixel stride);
//   * weights: the [16][1152] pack of the kernel above, read with this kernel's row mapping into 24 A fragments per lane.
// Measured (64 clips, 1.07 GB in): 0.236-0.26 ms = 4.4-4.9 TB/s by run (the kernel above: 0.31).  With every DMA redirected to the zero page
// the same instruction stream takes 0.129 ms (0.078 without the MFMAs): what is left is the memory side.  Tried and not kept: a ring
// of whole image rows instead of channel quarters (16-KB contiguous DMAs, no reliance on the second half of an L2 line still being
// there: same 4.4-4.7, and 0.41 instead of 0.47-0.58 at 8 clips: six barriers per frame with at most three waves working); rings of
// 3 / 5 / 6-9 slots (3.2 / 4.3 / 4.0 TB/s); non-temporal DMAs (2.5 TB/s with quarters, 3.8 with rows: the halo rows and the other line
// half come from L2 or not at all); the XCD remap below is worth nothing measurable (FETCH_SIZE says the halo is an L2 hit either way).
// ------------------------------------------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float nout2_shift(float v) {             // 0x138: lane l <- lane l - 1; 0x130: lane l <- lane l + 1; zero fill
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int IMM>
__device__ __forceinline__ bf16x8_t nout2_read128(unsigned lds_addr) {
    bf16x8_t v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "n"(IMM));
    return v;
}

template <int NT>
__global__ void __launch_bounds__(256, 2) conv_narrow_out2_kernel(const NarrowOutArgs a) {
    constexpr int PXR = 32 * NT;                        // pixels per tile row = image width
    constexpr int ROWB = PXR * 64;                      // bytes per tile row and stage
    constexpr int STAGE = 6 * ROWB;                     // 24576 / 12288
    constexpr int DPS = STAGE / 1024;                   // DMA instructions per stage (16 pixels x 64 B each)
    constexpr int DPW = DPS / 4;                        // ... per wave: 6 / 3
    constexpr int DPR = PXR / 16;                       // ... per tile row
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n32 = lane & 31, kh = lane >> 5;
    const int W = a.W, H = a.H;
    const unsigned lds0 = nout_lds_offset(smem);

    // consecutive logical blocks (row blocks of one sample: shared halo rows) on ONE XCD: hardware block b runs on XCD b % 8
    int b = blockIdx.x;
    if ((gridDim.x & 7) == 0) b = (b & 7) * (int)(gridDim.x >> 3) + (b >> 3);
    const int seg = b % a.tsegs; b /= a.tsegs;
    const int hb = b % a.hblocks;
    const int n = b / a.hblocks;
    const int h0 = hb * 4;
    const int h = h0 + wave;
    const bool row_ok = h < H;                                                 // wave-uniform
    const int ts0 = seg * a.tseg_len, ts1 = (ts0 + a.tseg_len < a.T) ? ts0 + a.tseg_len : a.T;
    const int nf = ts1 - ts0 + 2;
    const int tin_first = ts0 + a.t_lo;

    // weights: A fragments.  MFMA row = lane & 31 -> (dt, dw, co); k-step ks = (quarter * 3 + dh) * 2 + j covers channels quarter * 32 + j * 16 ..
    bf16x8_t wf[24];
    {
        int dt, sl;
        if (n32 < 24) { dt = n32 >> 3; sl = n32 & 7; } else { dt = n32 - 24; sl = 8; }
        const int dw = sl / 3, co = sl - 3 * dw;
        const bool used = n32 < 27;
        const bf16_t* wrow = a.wpack + (4 * (used ? dt : 0) + co) * NOUT_WROW + dw * 128 + kh * 8;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int dh = 0; dh < 3; ++dh)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(wrow + dh * 384 + q * 32 + j * 16);
                    if (!used) v = bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
                    wf[(q * 3 + dh) * 2 + j] = v;
                }
    }
    float bias_r[3] = {0.f, 0.f, 0.f};
    if (a.bias) {
#pragma unroll
        for (int c = 0; c < 3; ++c) if (c < a.cout) bias_r[c] = a.bias[c];
    }
    // pin the completion of these loads HERE (see the kernel above: counted vmcnt and the inline-asm DMA stream)
#pragma unroll
    for (int c = 0; c < 3; ++c) asm volatile("" : "+v"(bias_r[c]));
#pragma unroll
    for (int ks = 0; ks < 24; ++ks) asm volatile("" : "+v"(wf[ks]));

    // DMA piece d = wave + 4 i: tile row d / DPR, pixels 16 (d % DPR) .. + 15; lane -> pixel + (lane >> 2), LDS slot lane & 3 holds chunk slot ^ key(pixel)
    unsigned dma_off[DPW];
#pragma unroll
    for (int i = 0; i < DPW; ++i) {
        const int d = wave + 4 * i;
        const int rr = d / DPR, p = (d % DPR) * 16 + (lane >> 2);
        const int hin = h0 - 1 + rr;
        const bool ok = hin >= 0 && hin < H && p < W;
        dma_off[i] = ok ? (unsigned)((hin * W + p) * 256 + (((lane & 3) ^ ((p >> 2) & 3)) << 4)) : 0xffffffffu;
    }
    const char* const zero_ptr = reinterpret_cast<const char*>(g_zero_page_n) + (lane & 3) * 16;
    // fragment reads: B column = pixel n32 (+ 32 for the second tile), 8 channels kh of k-step j: chunk (2 j + kh) ^ key
    const unsigned key = (unsigned)((n32 >> 2) & 3);
    const unsigned rd0 = (unsigned)((wave * PXR + n32) * 64) + (((unsigned)kh ^ key) << 4);
    const unsigned rd1 = (unsigned)((wave * PXR + n32) * 64) + (((unsigned)(2 + kh) ^ key) << 4);

    auto issue_stage = [&](int f, int q, int buf) {
        const int tin = tin_first + f;
        const bool fv = f < nf && tin >= 0 && tin < a.T && !(a.wblocks & 8);                       // wave-uniform
        const char* base = reinterpret_cast<const char*>(a.src) + (((long long)n * a.T + (fv ? tin : 0)) * H * W) * 256 + q * 64;
        const unsigned dst0 = lds0 + (unsigned)buf * STAGE + wave * 1024;
#pragma unroll
        for (int i = 0; i < DPW; ++i) {
            const char* p = (fv && dma_off[i] != 0xffffffffu) ? base + dma_off[i] : zero_ptr;
            const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(dst0 + i * 4096));
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(p), "s"(dst) : "memory", "m0");
        }
    };

    f32x16_t acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    u32x4_t pend = u32x4_t{0u, 0u, 0u, 0u};
    int pend_t = -1;
    const int pcol = lane;                                                     // output column of this lane after the half swap
    auto flush = [&]() {
        if (pend_t >= 0 && pcol < W)
            *reinterpret_cast<u32x4_t*>(a.dst + ((((long long)n * a.T + pend_t) * H + h) * W + pcol) * 8) = pend;
        pend_t = -1;
    };

    issue_stage(0, 0, 0);
    issue_stage(0, 1, 1);
    int buf = 0;                                                               // buffer of the stage being consumed
    for (int f = 0; f < nf; ++f) {
        const int tin = tin_first + f;
        const bool live = row_ok && tin >= 0 && tin < a.T && !(a.wblocks & 1);                     // wave-uniform
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if constexpr (DPW == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            __builtin_amdgcn_s_barrier();                                      // stage (f, q) landed for everyone; everyone is done with the stage before it
            asm volatile("" ::: "memory");
            if (q == 0) flush();                                               // the store goes out BEFORE the next DMA batch (counted wait above)
            {
                const int b2 = buf >= 1 ? buf - 1 : 2;                         // (buf + 2) % 3: the buffer the previous stage left
                issue_stage(f + (q + 2) / 4, (q + 2) & 3, b2);
            }
            if (live) {
                const unsigned vb = lds0 + (unsigned)buf * STAGE;
                const unsigned a0 = vb + rd0, a1 = vb + rd1;
                bf16x8_t x[6][NT];
#pragma unroll
                for (int dh = 0; dh < 3; ++dh) {
                    x[2 * dh][0] = (dh == 0) ? nout2_read128<0>(a0) : (dh == 1) ? nout2_read128<ROWB>(a0) : nout2_read128<2 * ROWB>(a0);
                    if constexpr (NT == 2) x[2 * dh][1] = (dh == 0) ? nout2_read128<2048>(a0) : (dh == 1) ? nout2_read128<ROWB + 2048>(a0) : nout2_read128<2 * ROWB + 2048>(a0);
                    x[2 * dh + 1][0] = (dh == 0) ? nout2_read128<0>(a1) : (dh == 1) ? nout2_read128<ROWB>(a1) : nout2_read128<2 * ROWB>(a1);
                    if constexpr (NT == 2) x[2 * dh + 1][1] = (dh == 0) ? nout2_read128<2048>(a1) : (dh == 1) ? nout2_read128<ROWB + 2048>(a1) : nout2_read128<2 * ROWB + 2048>(a1);
                }
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    // reads complete in order: k-step k needs the first (k + 1) NT of the 6 NT
                    if constexpr (NT == 2) {
                        if (k == 0) asm volatile("s_waitcnt lgkmcnt(10)" ::: "memory");
                        if (k == 1) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
                        if (k == 2) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
                        if (k == 3) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                        if (k == 4) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
                        if (k == 5) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    } else {
                        if (k == 0) asm volatile("s_waitcnt lgkmcnt(5)" ::: "memory");
                        if (k == 1) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                        if (k == 2) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
                        if (k == 3) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
                        if (k == 4) asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
                        if (k == 5) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    }
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        asm volatile("" : "+v"(x[k][t]));
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[q * 6 + k], x[k][t], acc[t], 0, 0, 0);
                    }
                }
            }
            buf = buf == 2 ? 0 : buf + 1;
        }
        // the quad dt = 2 (registers 8-11, and register 14 for slot 8) now holds out[ts0 + f - 2]: slots 0-3 in lanes 0-31, 4-7 in lanes 32-63
        if (f >= 2 && row_ok) {
            float y[9];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float flo = acc[0][8 + r], fhi = acc[NT - 1][8 + r];       // (__builtin_bit_cast of a vector ELEMENT reads element 0: copy first)
                const unsigned lo = __float_as_uint(flo);
                const unsigned hi = NT == 2 ? __float_as_uint(fhi) : 0u;
                const auto sw = __builtin_amdgcn_permlane32_swap(lo, hi, false, false);    // [0]: lower halves of both tiles, [1]: upper halves
                const unsigned s0 = sw[0], s1 = sw[1];
                y[r] = __uint_as_float(s0);
                y[4 + r] = __uint_as_float(s1);
            }
            {
                const float flo = acc[0][14], fhi = acc[NT - 1][14];
                const unsigned lo = __float_as_uint(flo);
                const unsigned hi = NT == 2 ? __float_as_uint(fhi) : 0u;
                const auto sw = __builtin_amdgcn_permlane32_swap(lo, hi, false, false);
                const unsigned s0 = sw[0];
                y[8] = __uint_as_float(s0);
            }
            float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 3; ++c)
                if (c < a.cout) o[c] = (nout2_shift<0x138>(y[c]) + y[3 + c]) + (nout2_shift<0x130>(y[6 + c]) + bias_r[c]);
            pend = pack8(o);
            pend_t = ts0 + f - 2;
        }
        // partial sums move to the next frame tap: quad dt -> quad dt + 1, slot 8: register 12 -> 13 -> 14
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { acc[t][8 + r] = acc[t][4 + r]; acc[t][4 + r] = acc[t][r]; acc[t][r] = 0.f; }
            acc[t][14] = acc[t][13]; acc[t][13] = acc[t][12]; acc[t][12] = 0.f;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    flush();
}


// ------------------------------------------------------------------------------------------------------------------------------
// Weight gradients of the two narrow convolutions: stem CausalConv3d(3 -> 128) and head CausalConv3d(128 -> 3) (tokenizer.py:25,172).
// Both are      G[ch][(tap, c)] += sum_pixels BIG[pixel][ch] * SMALL[pixel + tap][c]              (ch < 128, 27 taps, c < 4)
// with BIG the 128-channel tensor (stem: the output gradient; head: the input) and SMALL the <= 4-channel one (stem: the input; head:
// the output gradient with the taps flipped): 16.8 MB + 0.4 MB read per clip, a 57-KB result.  The generic kernel gives every tap its
// own pass over BIG (27 x the traffic, 25-30 TFLOP/s, 1.2 % of a training step).  Here a workgroup walks 64-pixel chunks: the BIG
// tile [64][128] is register-staged one chunk ahead, an im2col tile [64 pixels][27 taps x 4 channels (+ a column of ones that makes
// G[:, 108] the sum of BIG = the stem's bias gradient)] is built in LDS from a (3 frames x rows x (CW + 2) pixels) image of SMALL with
// explicit zero borders, and both pixel-major tiles feed 32x32x16 MFMAs through ds_read_b64_tr_b16 -- one pass over BIG, G
// accumulated in registers over the whole pixel range and added to the fp32 result with atomics at the end.
// ------------------------------------------------------------------------------------------------------------------------------
struct NarrowWgradArgs {
    const bf16_t* big;      // CL [N][T][H][W][128]
    const bf16_t* small_;   // CL [N][T][H][W][sp]
    float* G;               // [128][128] fp32, column = tap * 4 + c (108: ones column), accumulated
    int N, T, H, W, sp;
    int t_lo;               // SMALL frame of tap plane dt = t + t_lo + dt
    int ones;               // 1: column 108 = 1.0
    int nchunks, chunks_per_block;
    // direct mode (genie_conv_narrow_wgrad_acc): G null, the tile goes straight into the parameter gradients
    float* dW;              // stem: [128][cs][27], head: [cs][128][27] (fp32, accumulated)
    float* dbias;           // stem: [128], head: [cs]; or null
    int stem, cs;           // cs = the narrow side's real channel count (<= 4)
    int wcl;                // 0: dW in (co, ci, tap) memory order; 1: (co, tap, ci) = torch.channels_last_3d, the layout the modules keep
    // wide side of more than 128 channels (LatentAction's proj_in / proj_out: 256), one launch per 128-channel slab (second-cut kernel only):
    int big_pitch;          // channels per pixel of BIG (128 for the tokenizer's stem / head)
    int wide0, wide_total;  // this launch's first channel of the wide side and their total number (dW / dbias addressing)
};

template <int CW, int RPC>      // chunk = RPC image rows of CW pixels (CW * RPC == 64); W == CW, or W == 128 with CW = 64 (half rows)
__global__ void __launch_bounds__(256) conv_narrow_wgrad_kernel(const NarrowWgradArgs a) {
    constexpr int IR = RPC + 2, IC = CW + 2, IMG = 3 * IR * IC;            // image pixels (16 B each)
    constexpr int IMG_LOADS = (IMG + 255) / 256;
    __shared__ __attribute__((aligned(16))) char AB_[2 * 64 * 256];
    char* const A_ = AB_;                                                  // BIG tile  [64 px][128 ch], chunk c of row r at c ^ ((r & 3) << 2)
    char* const B_ = AB_ + 64 * 256;                                       // im2col    [64 px][128 cols], same swizzle
    __shared__ __attribute__((aligned(16))) u32x4_t img[IMG];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int W = a.W, H = a.H, T = a.T;
    const int halves = W / CW;                                            // chunks per image-row group (2 at W = 128)
    int c0 = blockIdx.x * a.chunks_per_block, c1 = c0 + a.chunks_per_block;
    if (c1 > a.nchunks) c1 = a.nchunks;
    if (c0 >= c1) return;

    for (int i = tid; i < 64 * 16; i += 256) reinterpret_cast<u32x4_t*>(B_)[i] = u32x4_t{0u, 0u, 0u, 0u};   // columns >= 112 stay zero

    // chunk -> (n, t, h0, w0): chunks run along w (halves), then h (RPC rows each), then t, n
    auto decode = [&](int c, int& n, int& t, int& h0, int& w0) {
        w0 = (c % halves) * CW; c /= halves;
        const int hb = H / RPC;
        h0 = (c % hb) * RPC; c /= hb;
        t = c % T; n = c / T;
    };
    u32x4_t breg[4], ireg[IMG_LOADS];
    auto load_chunk = [&](int c) {
        int n, t, h0, w0;
        decode(c, n, t, h0, w0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {                                     // BIG: 64 px x 16 chunks of 16 B; thread -> (px = i * 16 + tid / 16, chunk tid % 16)
            const int px = i * 16 + (tid >> 4), ch = tid & 15;
            const int hh = h0 + px / CW, ww = w0 + px % CW;
            breg[i] = *reinterpret_cast<const u32x4_t*>(a.big + ((((long long)n * T + t) * H + hh) * W + ww) * 128 + ch * 8);
        }
#pragma unroll
        for (int i = 0; i < IMG_LOADS; ++i) {
            const int li = i * 256 + tid;
            u32x4_t v = u32x4_t{0u, 0u, 0u, 0u};
            if (li < IMG) {
                const int f = li / (IR * IC), r = (li / IC) % IR, cc = li % IC;
                const int tt = t + a.t_lo + f, hh = h0 - 1 + r, ww = w0 - 1 + cc;
                if ((unsigned)tt < (unsigned)T && (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W) {
                    const bf16_t* q = a.small_ + ((((long long)n * T + tt) * H + hh) * W + ww) * a.sp;
                    if (a.sp >= 8) v = *reinterpret_cast<const u32x4_t*>(q);
                    else { const u32x2_t h2 = *reinterpret_cast<const u32x2_t*>(q); v[0] = h2[0]; v[1] = h2[1]; }
                }
            }
            ireg[i] = v;
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int px = i * 16 + (tid >> 4), ch = tid & 15;
            *reinterpret_cast<u32x4_t*>(A_ + px * 256 + ((ch ^ ((px & 3) << 2)) << 4)) = breg[i];
        }
#pragma unroll
        for (int i = 0; i < IMG_LOADS; ++i) {
            const int li = i * 256 + tid;
            if (li < IMG) img[li] = ireg[i];
        }
    };
    // im2col: item (px, slot): slot < 27 = tap (dt, dh, dw) -> 4 channels of image pixel (dt, px / CW + dh, px % CW + dw); slot 27 = ones column
    // head conv in direct mode: the bias gradient is the plain sum of SMALL (= dy) -- every pixel is the (dt = -t_lo, dh = 1, dw = 1) tap of
    // exactly one chunk pixel
    const int self_slot = (a.G == nullptr && !a.stem && a.dbias) ? -a.t_lo * 9 + 4 : -1;
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    auto build = [&]() {
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const int it = i * 256 + tid;                                 // 64 * 28 = 1792 items
            const int px = it / 28, slot = it - px * 28;
            u32x2_t v;
            if (slot < 27) {
                const int dt = slot / 9, dh = (slot / 3) % 3, dw = slot % 3;
                const u32x4_t p = img[(dt * IR + px / CW + dh) * IC + px % CW + dw];
                v[0] = p[0]; v[1] = p[1];
                if (slot == self_slot) {
                    bsum[0] += __uint_as_float(v[0] << 16); bsum[1] += __uint_as_float(v[0] & 0xffff0000u);
                    bsum[2] += __uint_as_float(v[1] << 16); bsum[3] += __uint_as_float(v[1] & 0xffff0000u);
                }
            } else {
                v[0] = a.ones ? 0x00003F80u : 0u; v[1] = 0u;             // {1.0, 0, 0, 0}
            }
            const int chunk = slot >> 1;
            *reinterpret_cast<u32x2_t*>(B_ + px * 256 + ((chunk ^ ((px & 3) << 2)) << 4) + (slot & 1) * 8) = v;
        }
    };

    // transposing-read addresses (as in conv_wgrad.hip): lane = 16 g + 4 r + q reads k-row 8 (g >> 1) + r (+ 16 kstep, + 4 second read),
    // channels 16 (g & 1) + 4 q .. + 3 of a 32-wide tile
    const int g16 = lane >> 4, rr = (lane >> 2) & 3, qq = lane & 3;
    const int krow = 8 * (g16 >> 1) + rr;
    int a_off[2], b_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ca = wm * 64 + i * 32 + 16 * (g16 & 1) + 4 * qq, cb = wn * 64 + i * 32 + 16 * (g16 & 1) + 4 * qq;
        a_off[i] = krow * 256 + (((ca >> 3) ^ (rr << 2)) << 4) + (ca & 7) * 2;
        b_off[i] = krow * 256 + (((cb >> 3) ^ (rr << 2)) << 4) + (cb & 7) * 2;
    }
    auto tr16 = [&](const char* p) -> bf16x4_t {
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4_t*)(p));
    };
    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    load_chunk(c0);
    for (int c = c0; c < c1; ++c) {
        __syncthreads();                                                  // the previous chunk's MFMAs are done with A_ / B_ / img
        store_chunk();
        __syncthreads();
        if (c + 1 < c1) load_chunk(c + 1);                                // in flight under the build and the MFMAs
        build();
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8_t af[2], bfr[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const bf16x4_t lo = tr16(A_ + ks * 16 * 256 + a_off[i]), hi = tr16(A_ + (ks * 16 + 4) * 256 + a_off[i]);
                af[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                const bf16x4_t l2 = tr16(B_ + ks * 16 * 256 + b_off[i]), h2 = tr16(B_ + (ks * 16 + 4) * 256 + b_off[i]);
                bfr[i] = __builtin_shufflevector(l2, h2, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    }
    // D row = BIG channel (registers), col = im2col column (lane & 31)
    const int khalf = lane >> 5;
    if (a.G) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r16 = 0; r16 < 16; ++r16) {
                const int ch = wm * 64 + i * 32 + (r16 & 3) + 8 * (r16 >> 2) + 4 * khalf;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int col = wn * 64 + j * 32 + (lane & 31);
                    if (col < 112) atomicAdd(a.G + ch * 128 + col, acc[i][j][r16]);
                }
            }
    } else {
        // direct mode: the tile goes through LDS (the A / B tiles are done) in the ORDER OF THE PARAMETER, 64 BIG channels at a time, and leaves as
        // contiguous atomics -- scattered 4-byte atomics in the parameter's order straight from the MFMA layout cost 3x the whole kernel
        float* const stage = reinterpret_cast<float*>(AB_);               // stem: [64 ch][cs][27] (+ [64] bias sums); head: [cs][64 ch][27]
        const int cs = a.cs, per = 64 * cs * 27;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            __syncthreads();                                              // the MFMA reads of A_ / B_ (half 0) or the previous half's atomics are done
            if (wm == half) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r16 = 0; r16 < 16; ++r16) {
                        const int chl = i * 32 + (r16 & 3) + 8 * (r16 >> 2) + 4 * khalf;        // channel inside the half
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const int col = wn * 64 + j * 32 + (lane & 31);
                            const int tap = col >> 2, c = col & 3;
                            if (col < 108) {
                                if (c < cs) {
                                    const int e = a.stem ? (a.wcl ? (chl * 27 + tap) * cs + c : (chl * cs + c) * 27 + tap)
                                                         : (a.wcl ? (c * 27 + 26 - tap) * 64 + chl : (c * 64 + chl) * 27 + 26 - tap);
                                    stage[e] = acc[i][j][r16];
                                }
                            } else if (col == 108) {
                                stage[per + chl] = acc[i][j][r16];
                            }
                        }
                    }
            }
            __syncthreads();
            if (a.stem) {
                float* const dst = a.dW + (long long)half * per;
                for (int e = tid; e < per; e += 256) atomicAdd(dst + e, stage[e]);
                if (a.dbias && tid < 64) atomicAdd(a.dbias + half * 64 + tid, stage[per + tid]);
            } else {
                for (int e = tid; e < per; e += 256) {
                    int d;
                    if (a.wcl) d = (e >> 6) * 128 + half * 64 + (e & 63);                       // (co, tap) rows of 128 input channels
                    else { const int c = e / (64 * 27); d = (c * 128 + half * 64) * 27 + (e - c * (64 * 27)); }
                    atomicAdd(a.dW + d, stage[e]);
                }
            }
        }
    }
    if (self_slot >= 0) {
        // ONE atomic instruction per workgroup (lanes 0 .. cs - 1, one cache line): per-wave atomics to these three addresses serialise in L2 --
        // 6144 of them cost 65 us at 8 clips, more than the rest of the kernel
        float* const red = reinterpret_cast<float*>(img);
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float t = wave_sum(bsum[c]);
            if (lane == 0) red[wave * 4 + c] = t;
        }
        __syncthreads();
        if (tid < a.cs) atomicAdd(a.dbias + tid, (red[tid] + red[4 + tid]) + (red[8 + tid] + red[12 + tid]));
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Weight gradient, second cut (round 6).  What the counters said about the kernel above (profiles/r05_pmc_narrow_wgrad.txt): matrix pipe 19 %,
// 8.6 scalar + 9.4 vector instructions per MFMA, 1.9 of 3 waves per SIMD resident, 0.68 bank-conflict cycles per LDS instruction -- a chain of
// three barriers per 16-KB chunk (LDS store | im2col build | transposing reads + MFMAs) whose links are index arithmetic.  Re-cut:
//   * NO im2col tile.  ds_read_b64_tr_b16 takes a per-lane address: the lane that owns (k-row = pixel, columns = the 4 channels of tap
//     (dt, dh, dw)) reads those 8 bytes straight from the zero-bordered IMAGE tile of SMALL at pixel + (dt, dh, dw) -- its offset is a
//     loop-invariant lane constant plus a compile-time k-step term.  The ones column (bias gradient) and the pad columns read a constant slot.
//     Gone: 28 KB of LDS traffic, 1792 items of div / mod arithmetic and one barrier per chunk, 16 KB of LDS per workgroup.
//   * two stages {BIG tile, image} in LDS: chunk c + 1 is written while chunk c is multiplied -- ONE barrier per chunk -- and chunk c + 2
//     is in flight from HBM in registers meanwhile.
//   * chunk -> (n, t, h, w) by carries, the image items' (frame, row, column) decoded once per thread.
// Same arguments, tile layout and epilogue as the kernel above (which stays behind GENIE_NARROW_WGRAD_CUT=1).
// ------------------------------------------------------------------------------------------------------------------------------
template <int CW, int RPC, int TEAMS>      // TEAMS teams of four waves share a workgroup: each walks every TEAMS-th chunk of the block's range through its own
__global__ void __launch_bounds__(256 * TEAMS) conv_narrow_wgrad2_kernel(const NarrowWgradArgs a) {      // two LDS stages; ONE tile of atomics per workgroup at the end
    constexpr int IR = RPC + 2, IC = CW + 2, IMG = 3 * IR * IC;            // image pixels (8 B each: the first 4 channels)
    constexpr int IMG_LOADS = (IMG + 255) / 256;
    constexpr int IMGB = ((IMG * 8 + 15) & ~15) + 16;                      // + the constant slot {1, 0, 0, 0 | 0, 0, 0, 0}
    constexpr int STG = 64 * 256 + IMGB;
    extern __shared__ __attribute__((aligned(16))) char smem_all[];        // TEAMS x 2 stages (>= 64 KB for the epilogue's exchange when TEAMS > 1)
    const int team = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);
    char* const smem = smem_all + team * (2 * STG);
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int W = a.W, H = a.H, T = a.T;
    const int halves = W / CW, hb = H / RPC;
    int c0 = blockIdx.x * a.chunks_per_block, c1 = c0 + a.chunks_per_block;
    if (c1 > a.nchunks) c1 = a.nchunks;
    if (c0 >= c1) return;
    const int niter = (c1 - c0 + TEAMS - 1) / TEAMS;                       // every team runs the same number of barrier rounds
    c0 += team;                                                           // this team's chunks: c0, c0 + TEAMS, ...

    if (tid < 4) {
#pragma unroll
        for (int s = 0; s < 2; ++s) reinterpret_cast<uint32_t*>(smem + s * STG + 64 * 256 + IMGB - 16)[tid] = (tid == 0 && a.ones) ? 0x00003F80u : 0u;
    }
    // chunk position, advanced by carries: chunks run along w (halves), then h (RPC rows each), then t, n
    int pn, pt, ph, pw;
    {
        int c = c0;
        pw = c % halves; c /= halves;
        ph = c % hb; c /= hb;
        pt = c % T; pn = c / T;
    }
    // this thread's image items: (frame f, row r, column cc) of item i * 256 + tid, fixed for the launch
    int it_f[IMG_LOADS], it_r[IMG_LOADS], it_c[IMG_LOADS];
#pragma unroll
    for (int i = 0; i < IMG_LOADS; ++i) {
        const int li = i * 256 + tid;
        it_f[i] = li / (IR * IC); it_r[i] = (li / IC) % IR; it_c[i] = li % IC;
    }
    const int self_f = (a.G == nullptr && !a.stem && a.dbias) ? -a.t_lo : -1;      // head conv, direct mode: the bias gradient is the plain sum of SMALL
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    u32x4_t breg[4];
    u32x2_t ireg[IMG_LOADS];
    long long big_off = (long long)c0 * 64 * a.big_pitch + a.wide0;       // a chunk is 64 consecutive pixels of BIG; this launch reads 128 of their big_pitch channels
    auto load_chunk = [&]() {                                             // the chunk at (pn, pt, ph, pw) / big_off
#pragma unroll
        for (int i = 0; i < 4; ++i)                                       // thread -> (px = i * 16 + tid / 16, 16-B piece tid % 16): 256 B per 16 lanes
            breg[i] = *reinterpret_cast<const u32x4_t*>(a.big + big_off + (long long)(i * 16 + (tid >> 4)) * a.big_pitch + (tid & 15) * 8);
        const int h0 = ph * RPC, w0 = pw * CW;
#pragma unroll
        for (int i = 0; i < IMG_LOADS; ++i) {
            u32x2_t v = u32x2_t{0u, 0u};
            const int tt = pt + a.t_lo + it_f[i], hh = h0 - 1 + it_r[i], ww = w0 - 1 + it_c[i];
            if (i * 256 + tid < IMG && (unsigned)tt < (unsigned)T && (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W)
                v = *reinterpret_cast<const u32x2_t*>(a.small_ + ((((long long)pn * T + tt) * H + hh) * W + ww) * a.sp);
            ireg[i] = v;
        }
    };
    auto advance = [&]() {
        big_off += (long long)64 * a.big_pitch * TEAMS;
#pragma unroll
        for (int i = 0; i < TEAMS; ++i)
            if (++pw == halves) { pw = 0; if (++ph == hb) { ph = 0; if (++pt == T) { pt = 0; ++pn; } } }
    };
    auto store_chunk = [&](int s) {
        char* const A_ = smem + s * STG;
        char* const I_ = A_ + 64 * 256;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int px = i * 16 + (tid >> 4), ch = tid & 15;
            *reinterpret_cast<u32x4_t*>(A_ + px * 256 + ((ch ^ ((px & 3) << 2)) << 4)) = breg[i];
        }
#pragma unroll
        for (int i = 0; i < IMG_LOADS; ++i) {
            if (i * 256 + tid < IMG) {
                *reinterpret_cast<u32x2_t*>(I_ + (i * 256 + tid) * 8) = ireg[i];
                if (it_f[i] == self_f && it_r[i] >= 1 && it_r[i] <= RPC && it_c[i] >= 1 && it_c[i] <= CW) {
                    bsum[0] += __uint_as_float(ireg[i][0] << 16); bsum[1] += __uint_as_float(ireg[i][0] & 0xffff0000u);
                    bsum[2] += __uint_as_float(ireg[i][1] << 16); bsum[3] += __uint_as_float(ireg[i][1] & 0xffff0000u);
                }
            }
        }
    };
    // transposing-read addresses: lane = 16 g + 4 r + q reads k-row 8 (g >> 1) + r (+ 16 k-step, + 4 second read), 4 columns 16 (g & 1) + 4 q .. + 3 of a
    // 32-wide tile.  A: BIG channels from the swizzled tile.  B: column group = tap 8 (2 wn + j) + 4 (g & 1) + q -- from the image at
    // pixel + tap offset (8 B per pixel), or from the constant slot (tap 27: ones; taps 28..31: zeros; no k-step term there)
    const int g16 = lane >> 4, rr = (lane >> 2) & 3, qq = lane & 3;
    const int krow = 8 * (g16 >> 1) + rr;
    int a_off[2], b_off[2], b_mul[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ca = wm * 64 + i * 32 + 16 * (g16 & 1) + 4 * qq;
        a_off[i] = krow * 256 + (((ca >> 3) ^ (rr << 2)) << 4) + (ca & 7) * 2;
        const int tap = 8 * (2 * wn + i) + 4 * (g16 & 1) + qq;
        if (tap < 27) {
            const int dt = tap / 9, dh = (tap / 3) % 3, dw = tap % 3;
            b_off[i] = 64 * 256 + ((dt * IR + dh) * IC + dw + krow) * 8;       // (krow < 16 stays inside one image row: 16 | CW)
            b_mul[i] = 8;
        } else {
            b_off[i] = 64 * 256 + IMGB - 16 + (tap == 27 ? 0 : 8);
            b_mul[i] = 0;
        }
    }
    auto tr16 = [&](const char* p) -> bf16x4_t {
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4_t*)(p));
    };
    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (c0 < c1) {
        load_chunk();
        store_chunk(0);
        advance();
        if (c0 + TEAMS < c1) load_chunk();
    }
    __syncthreads();
    for (int k = 0; k < niter; ++k) {
        const int c = c0 + k * TEAMS, s = k & 1;
        if (c + TEAMS < c1) {
            store_chunk(s ^ 1);                                           // the team's next chunk (registers) -> the other stage; its last reader passed the barrier below
            advance();
            if (c + 2 * TEAMS < c1) load_chunk();                         // the one after: in flight under this chunk's MFMAs and the next store
        }
        const char* const S_ = smem + s * STG;
        if (c < c1)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            // pixel of k-row krow in k-step ks: 16 ks + krow (+ 4) -> image (row (16 ks) / CW, column (16 ks) % CW + krow)
            constexpr int dummy = 0; (void)dummy;
            const int kpix = ((16 * ks) / CW) * IC + (16 * ks) % CW;
            bf16x8_t af[2], bfr[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const bf16x4_t lo = tr16(S_ + ks * 16 * 256 + a_off[i]), hi = tr16(S_ + (ks * 16 + 4) * 256 + a_off[i]);
                af[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                const bf16x4_t l2 = tr16(S_ + b_off[i] + kpix * b_mul[i]), h2 = tr16(S_ + b_off[i] + (kpix + 4) * b_mul[i]);
                bfr[i] = __builtin_shufflevector(l2, h2, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();                                                  // stage s is free again; stage s ^ 1 is complete
    }
    if (self_f >= 0) {
        // ONE atomic instruction per team (lanes 0 .. cs - 1, one cache line): per-wave atomics to these three addresses serialise in L2
        float* const red = reinterpret_cast<float*>(smem);                // (the team's stages are done: the loop ended on a barrier)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float t = wave_sum(bsum[c]);
            if (lane == 0) red[wave * 4 + c] = t;
        }
        __syncthreads();
        if (tid < a.cs) atomicAdd(a.dbias + tid, (red[tid] + red[4 + tid]) + (red[8 + tid] + red[12 + tid]));
        __syncthreads();
    }
    // the teams' tiles meet in team 0 (through LDS, one team at a time: 64 KB each), which alone runs the epilogue
    if constexpr (TEAMS > 1) {
        float* const xch = reinterpret_cast<float*>(smem_all);
#pragma unroll
        for (int tm = 1; tm < TEAMS; ++tm) {
            if (team == tm) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) xch[(((i * 2 + j) * 16 + r) * 4 + wave) * 64 + lane] = acc[i][j][r];
            }
            __syncthreads();
            if (team == 0) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][j][r] += xch[(((i * 2 + j) * 16 + r) * 4 + wave) * 64 + lane];
            }
            __syncthreads();
        }
        if (team != 0) return;                                            // (s_barrier counts the surviving waves only)
    }
    // D row = BIG channel (registers), col = im2col column (lane & 31)
    const int khalf = lane >> 5;
    if (a.G) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r16 = 0; r16 < 16; ++r16) {
                const int ch = wm * 64 + i * 32 + (r16 & 3) + 8 * (r16 >> 2) + 4 * khalf;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int col = wn * 64 + j * 32 + (lane & 31);
                    if (col < 112) atomicAdd(a.G + ch * 128 + col, acc[i][j][r16]);
                }
            }
    } else {
        // direct mode: the tile goes through LDS (the stages are done) in the ORDER OF THE PARAMETER, 64 BIG channels at a time, and leaves as
        // contiguous atomics -- scattered 4-byte atomics in the parameter's order straight from the MFMA layout cost 3x the whole kernel
        float* const stage = reinterpret_cast<float*>(smem);              // stem: [64 ch][cs][27] (+ [64] bias sums); head: [cs][64 ch][27]
        const int cs = a.cs, per = 64 * cs * 27;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (half) __syncthreads();                                    // the previous half's atomics have read the stage
            if (wm == half) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r16 = 0; r16 < 16; ++r16) {
                        const int chl = i * 32 + (r16 & 3) + 8 * (r16 >> 2) + 4 * khalf;        // channel inside the half
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const int col = wn * 64 + j * 32 + (lane & 31);
                            const int tap = col >> 2, cc = col & 3;
                            if (col < 108) {
                                if (cc < cs) {
                                    const int e = a.stem ? (a.wcl ? (chl * 27 + tap) * cs + cc : (chl * cs + cc) * 27 + tap)
                                                         : (a.wcl ? (cc * 27 + 26 - tap) * 64 + chl : (cc * 64 + chl) * 27 + 26 - tap);
                                    stage[e] = acc[i][j][r16];
                                }
                            } else if (col == 108) {
                                stage[per + chl] = acc[i][j][r16];
                            }
                        }
                    }
            }
            __syncthreads();
            if (a.stem) {
                float* const dst = a.dW + ((long long)a.wide0 * cs * 27) + (long long)half * per;
                for (int e = tid; e < per; e += 256) atomicAdd(dst + e, stage[e]);
                if (a.dbias && tid < 64) atomicAdd(a.dbias + a.wide0 + half * 64 + tid, stage[per + tid]);
            } else {
                for (int e = tid; e < per; e += 256) {
                    int d;
                    if (a.wcl) d = (e >> 6) * a.wide_total + a.wide0 + half * 64 + (e & 63);    // (co, tap) rows of wide_total input channels
                    else { const int cc = e / (64 * 27); d = (cc * a.wide_total + a.wide0 + half * 64) * 27 + (e - cc * (64 * 27)); }
                    atomicAdd(a.dW + d, stage[e]);
                }
            }
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

extern "C" int genie_conv_narrow_out(const void* src_cl, const void* wpack, const float* bias, void* dst_cl, int N, int T, int H, int W, int cout,
                                     int t_lo, void* stream) {
    GENIE_CHECK_ARG(src_cl && wpack && dst_cl, "genie_conv_narrow_out: null pointer");
    GENIE_CHECK_ARG(W >= 32 && W % 32 == 0, "genie_conv_narrow_out: image width %d is not a multiple of 32", W);
    GENIE_CHECK_ARG(cout >= 1 && cout <= 3 && t_lo >= -2 && t_lo <= 0, "genie_conv_narrow_out: cout %d (1..3) / t_lo %d (-2..0)", cout, t_lo);
    GENIE_CHECK_ARG(N >= 1 && T >= 1 && H >= 1, "genie_conv_narrow_out: bad geometry");
    GENIE_CHECK_ARG((long long)H * W * 256 < (1ll << 32), "genie_conv_narrow_out: frame of %d x %d pixels exceeds the 32-bit in-frame offset", H, W);
    NarrowOutArgs a;
    a.src = (const bf16_t*)src_cl; a.wpack = (const bf16_t*)wpack; a.bias = bias; a.dst = (bf16_t*)dst_cl;
    a.N = N; a.T = T; a.H = H; a.W = W; a.cout = cout; a.t_lo = t_lo;
    a.hblocks = (H + 3) / 4;
    hipStream_t s = (hipStream_t)stream;
    static const int cut = [] { const char* e = getenv("GENIE_NARROW_OUT_CUT"); return e ? atoi(e) : 2; }();
    if (W <= 64 && cut == 2) {
        // second cut: a wave owns a whole image row; 72 KB (W = 64) / 36 KB of LDS: two workgroups per CU
        a.wblocks = 0;                                 // (timing-probe bits: 1 = no MFMAs, 8 = every DMA from the zero page)
        int tsegs = 1;
        // every frame segment re-reads two input frames: split time only until every CU has ONE workgroup (8 clips: 256 workgroups of 10 input
        // frames reach 0.58 of the HBM peak, 384 / 512 of 7 / 6 frames 0.46 / 0.47)
        while ((long long)N * a.hblocks * tsegs < 256 && (T + tsegs) / (tsegs + 1) >= 4) ++tsegs;
        a.tseg_len = (T + tsegs - 1) / tsegs;
        a.tsegs = (T + a.tseg_len - 1) / a.tseg_len;
        const long long blocks = (long long)N * a.hblocks * a.tsegs;
        GENIE_CHECK_ARG(blocks < (1ll << 31), "genie_conv_narrow_out: too many workgroups");
        static bool configured2 = false;
        if (!configured2) {
            GENIE_CHECK_ARG(hipFuncSetAttribute((const void*)conv_narrow_out2_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) == hipSuccess,
                            "hipFuncSetAttribute failed");
            configured2 = true;
        }
        if (W == 64) conv_narrow_out2_kernel<2><<<(unsigned)blocks, 256, 3 * 6 * 64 * 64, s>>>(a);
        else conv_narrow_out2_kernel<1><<<(unsigned)blocks, 256, 3 * 6 * 32 * 64, s>>>(a);
        GENIE_CHECK_LAUNCH();
        return GENIE_OK;
    }
    a.wblocks = W / 32;
    // every frame segment re-reads two input frames: split time only while the chip is not yet filled twice over
    int tsegs = 1;
    while ((long long)N * a.hblocks * a.wblocks * tsegs < 512 && (T + tsegs) / (tsegs + 1) >= 4) ++tsegs;
    a.tseg_len = (T + tsegs - 1) / tsegs;
    a.tsegs = (T + a.tseg_len - 1) / a.tseg_len;
    const long long blocks = (long long)N * a.hblocks * a.wblocks * a.tsegs;
    GENIE_CHECK_ARG(blocks < (1ll << 31), "genie_conv_narrow_out: too many workgroups");
    const int lds = NOUT_NBUF * NOUT_BUF_BYTES;
    static bool configured = false;
    if (!configured) {
        GENIE_CHECK_ARG(hipFuncSetAttribute((const void*)conv_narrow_out_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess,
                        "hipFuncSetAttribute failed");
        configured = true;
    }
    conv_narrow_out_kernel<<<(unsigned)blocks, 256, lds, s>>>(a);
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}

static int narrow_wgrad_launch(NarrowWgradArgs& a, const char* who, void* stream) {
    const int N = a.N, T = a.T, H = a.H, W = a.W;
    GENIE_CHECK_ARG(a.big_pitch >= 128 && a.big_pitch % 8 == 0 && a.wide0 >= 0 && a.wide0 + 128 <= a.big_pitch && a.wide_total >= a.wide0 + 128, "%s: wide side %d + 128 of %d (pitch %d)", who, a.wide0, a.wide_total, a.big_pitch);
    GENIE_CHECK_ARG(W == 32 || W == 64 || W == 128, "%s: image width %d not in {32, 64, 128}", who, W);
    GENIE_CHECK_ARG(a.sp >= 4 && a.sp % 4 == 0, "%s: pitch %d of the narrow tensor", who, a.sp);
    GENIE_CHECK_ARG(N >= 1 && T >= 1 && H >= 1 && a.t_lo >= -2 && a.t_lo <= 0, "%s: bad geometry / t_lo %d", who, a.t_lo);
    GENIE_CHECK_ARG(W != 32 || H % 2 == 0, "%s: W = 32 needs an even image height (64-pixel chunks of two rows), got %d", who, H);
    const long long nch = (long long)N * T * H * W / 64;
    GENIE_CHECK_ARG(nch >= 1 && nch < (1ll << 31), "%s: chunk count", who);
    a.nchunks = (int)nch;
    // (tried: TWO chunks in flight per workgroup in registers -- 200 VGPRs, two workgroups per CU: 0.50 instead of 0.53 at 64 clips, unchanged at 8: the
    // chunk loop is its own chain of three barriers, the LDS tile store, the im2col build and the transposing reads, not the HBM round trip)
    // three workgroups fit a CU (42 KB LDS, 162 VGPRs): one full round of 768 when there are >= 16 chunks for each, else two per CU; >= 8 chunks per
    // workgroup to amortise the 57-KB atomics tail (8 clips: 512 workgroups 0.058 ms, 768 0.063, 1024 0.072; 64 clips: 0.281 / 0.270 / 0.265 stem,
    // 0.256 / 0.246 / 0.247 head)
    long long blocks = 768;
    if (blocks * 16 > nch) blocks = 512;
    if (blocks * 8 > nch) blocks = (nch + 7) / 8;
    a.chunks_per_block = (int)((nch + blocks - 1) / blocks);
    blocks = (nch + a.chunks_per_block - 1) / a.chunks_per_block;
    hipStream_t s = (hipStream_t)stream;
    static const int cut_env = [] { const char* e = getenv("GENIE_NARROW_WGRAD_CUT"); return e ? atoi(e) : 3; }();
    const int cut = (a.big_pitch != 128 || a.wide_total != 128) && cut_env == 1 ? 3 : cut_env;      // (the round-5 kernel knows 128-channel rows only)
    if (cut == 2 || cut == 3) {
        // cut 3 (default): 256 workgroups of THREE four-wave teams -- the same twelve waves per CU as three workgroups of one team, a third of the
        // 55-KB atomic tiles at the end (768 tiles = 8 M fp32 atomics were ~20 us of a 240-us launch, and of a 60-us one at 8 clips); cut 2: one team
        const int teams = cut == 3 ? 3 : 1;
        if (teams == 3) {
            long long b3 = 256;
            if (b3 * 3 * 4 > nch) b3 = (nch + 11) / 12;                   // at least four chunks per team
            a.chunks_per_block = (int)((nch + b3 - 1) / b3);
            blocks = (nch + a.chunks_per_block - 1) / a.chunks_per_block;
        }
        const int lds32 = teams * 2 * (64 * 256 + ((3 * 4 * 34 * 8 + 15) & ~15) + 16), lds64 = teams * 2 * (64 * 256 + ((3 * 3 * 66 * 8 + 15) & ~15) + 16);
        static bool configured = false;
        if (!configured) {
            GENIE_CHECK_ARG(hipFuncSetAttribute((const void*)conv_narrow_wgrad2_kernel<32, 2, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess &&
                            hipFuncSetAttribute((const void*)conv_narrow_wgrad2_kernel<64, 1, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess,
                            "hipFuncSetAttribute failed");
            configured = true;
        }
        const int lds = W == 32 ? lds32 : lds64;
        if (W == 32 && teams == 3) conv_narrow_wgrad2_kernel<32, 2, 3><<<(unsigned)blocks, 768, lds, s>>>(a);
        else if (W == 32) conv_narrow_wgrad2_kernel<32, 2, 1><<<(unsigned)blocks, 256, lds > 64 * 112 * 4 + 64 ? lds : 64 * 112 * 4 + 64, s>>>(a);
        else if (teams == 3) conv_narrow_wgrad2_kernel<64, 1, 3><<<(unsigned)blocks, 768, lds, s>>>(a);
        else conv_narrow_wgrad2_kernel<64, 1, 1><<<(unsigned)blocks, 256, lds > 64 * 112 * 4 + 64 ? lds : 64 * 112 * 4 + 64, s>>>(a);
    } else {
        if (W == 32) conv_narrow_wgrad_kernel<32, 2><<<(unsigned)blocks, 256, 0, s>>>(a);
        else conv_narrow_wgrad_kernel<64, 1><<<(unsigned)blocks, 256, 0, s>>>(a);
    }
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}

extern "C" int genie_conv_narrow_wgrad(const void* big_cl, const void* small_cl, int small_pitch, float* G, int N, int T, int H, int W, int t_lo,
                                       int ones, void* stream) {
    GENIE_CHECK_ARG(big_cl && small_cl && G, "genie_conv_narrow_wgrad: null pointer");
    NarrowWgradArgs a;
    a.big = (const bf16_t*)big_cl; a.small_ = (const bf16_t*)small_cl; a.G = G;
    a.N = N; a.T = T; a.H = H; a.W = W; a.sp = small_pitch; a.t_lo = t_lo; a.ones = ones;
    a.dW = nullptr; a.dbias = nullptr; a.stem = 0; a.cs = 0; a.wcl = 0;
    a.big_pitch = 128; a.wide0 = 0; a.wide_total = 128;
    return narrow_wgrad_launch(a, "genie_conv_narrow_wgrad", stream);
}

extern "C" int genie_conv_narrow_wgrad_acc(const void* big_cl, const void* small_cl, int small_pitch, float* dW, float* dbias, int N, int T, int H,
                                           int W, int t_lo, int stem, int cs, int w_channels_last, void* stream) {
    GENIE_CHECK_ARG(big_cl && small_cl && dW, "genie_conv_narrow_wgrad_acc: null pointer");
    GENIE_CHECK_ARG(cs >= 1 && cs <= 4 && cs <= small_pitch, "genie_conv_narrow_wgrad_acc: %d channels on the narrow side (1..4, pitch %d)", cs, small_pitch);
    NarrowWgradArgs a;
    a.big = (const bf16_t*)big_cl; a.small_ = (const bf16_t*)small_cl; a.G = nullptr;
    a.N = N; a.T = T; a.H = H; a.W = W; a.sp = small_pitch; a.t_lo = t_lo; a.ones = (stem && dbias) ? 1 : 0;
    a.dW = dW; a.dbias = dbias; a.stem = stem ? 1 : 0; a.cs = cs; a.wcl = w_channels_last ? 1 : 0;
    a.big_pitch = 128; a.wide0 = 0; a.wide_total = 128;
    return narrow_wgrad_launch(a, "genie_conv_narrow_wgrad_acc", stream);
}

// The same for a wide side of a multiple of 128 channels (LatentAction.proj_in 3 -> 256 / proj_out 256 -> 3, action.py:60-70): one launch per 128-channel
// slab [wide0, wide0 + 128) of the big_pitch-channel tensor; dW / dbias are the WHOLE parameter gradients (wide_total channels on the wide side).
extern "C" int genie_conv_narrow_wgrad_wide(const void* big_cl, int big_pitch, int wide0, int wide_total, const void* small_cl, int small_pitch, float* dW,
                                            float* dbias, int N, int T, int H, int W, int t_lo, int stem, int cs, int w_channels_last, void* stream) {
    GENIE_CHECK_ARG(big_cl && small_cl && dW, "genie_conv_narrow_wgrad_wide: null pointer");
    GENIE_CHECK_ARG(cs >= 1 && cs <= 4 && cs <= small_pitch, "genie_conv_narrow_wgrad_wide: %d channels on the narrow side (1..4, pitch %d)", cs, small_pitch);
    NarrowWgradArgs a;
    a.big = (const bf16_t*)big_cl; a.small_ = (const bf16_t*)small_cl; a.G = nullptr;
    a.N = N; a.T = T; a.H = H; a.W = W; a.sp = small_pitch; a.t_lo = t_lo; a.ones = (stem && dbias) ? 1 : 0;
    a.dW = dW; a.dbias = (stem || wide0 == 0) ? dbias : nullptr;        // head: the bias gradient is the sum of the NARROW tensor -- once, with the first slab
    a.stem = stem ? 1 : 0; a.cs = cs; a.wcl = w_channels_last ? 1 : 0;
    a.big_pitch = big_pitch; a.wide0 = wide0; a.wide_total = wide_total;
    return narrow_wgrad_launch(a, "genie_conv_narrow_wgrad_wide", stream);
}
