/* genie_hip.h -- C ABI of libgenie_hip.so, the MI355X (gfx950) implementation of open-genie's hot path.
 *
 * The reference (myscience/open-genie) has no FFI/plugin layer: its hot path is a set of Python
 * nn.Modules that call PyTorch ATen.  The drop-in seam is therefore the Python module registry
 * (reference genie/module/__init__.py:23-93); this header is the boundary UNDER that seam -- what a
 * maintainer's ctypes/cffi stub binds (INTEGRATION.md shows the stub).  Every entry point names the
 * reference call site(s) it replaces.
 *
 * Conventions
 *   - plain C: pointers are DEVICE pointers unless stated, sizes are explicit, no torch types.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  All functions only ENQUEUE
 *     work; none synchronises, allocates or frees device memory -> safe inside hipGraph capture.
 *   - return value: 0 = ok, <0 = error (GENIE_ERR_*); genie_last_error() returns the message for the
 *     calling thread.
 *   - activations are "CL": bf16, channels-last (N, T, H, W, Cp); the channel pitch Cp is a multiple
 *     of 8 and pad channels [C, Cp) hold zeros.  A CL tensor is what torch calls a
 *     (N, C, T, H, W) tensor in channels_last_3d strides.
 *   - fp32 for parameters, statistics, accumulation and losses.
 */
#ifndef GENIE_HIP_H
#define GENIE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GENIE_ABI_VERSION 13

#define GENIE_F32 0
#define GENIE_BF16 1

#define GENIE_ERR_ARG (-1)
#define GENIE_ERR_HIP (-2)

int genie_abi_version(void);
const char* genie_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Layout conversion at the model boundary.
 * replaces: the implicit NCTHW fp32 tensors the reference feeds to nn.Conv3d (video.py:185-192) and
 *           returns from VideoTokenizer.decode (tokenizer.py:319-330).
 * dims = {N, C, T, H, W}; strides in ELEMENTS of the strided tensor, same order.
 * ------------------------------------------------------------------------------------------- */
int genie_to_channels_last(const void* src, int src_dtype, const int64_t* dims, const int64_t* strides,
                           void* dst_cl, int cpitch, void* stream);
/* Inverse depth-to-space-time on a CL tensor: src CL [N][T P][H Q][W R][src_pitch] with cf % 8 == 0 channels -> dst CL
 * [N][T][H][W][P Q R cf], channel ((p Q + q) R + r) cf + c (sub-pixel-major).  replaces: the inverse of video.py:403-408's rearrange
 * in the backward-data pass of DepthToSpaceTimeUpsample. */
int genie_unshuffle_cl(const void* src_cl, int src_pitch, void* dst_cl, int N, int T, int H, int W, int cf, int P, int Q, int R, void* stream);
int genie_from_channels_last(const void* src_cl, int cpitch, const int64_t* dims, void* dst, int dst_dtype,
                             const int64_t* strides, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Conv3d family as one gather-GEMM (conv_igemm.hip).
 * replaces: F.pad + nn.Conv3d in CausalConv3d.forward (video.py:178-192), nn.Conv3d in
 *           VideoResidualBlock (video.py:580-620) and the ST-block FFN (attention.py:429-438),
 *           SpaceTimeDownsample (video.py:457-483), DepthToSpaceTimeUpsample conv + Rearrange
 *           (video.py:396-408), and their autograd backward-data passes.
 *
 *   D[m][n] = sum_taps sum_{c<nch} SRC[pix(m)*step + (dt,dh,dw)][c0 + c] * WGT[row(n)][wofs + c]
 *   dest element = DST[(pix(m) * dm + do + subpixel(n / shuf_c))][n % shuf_c]  (+ bias, + resid)
 * ------------------------------------------------------------------------------------------- */
typedef struct GenieTap {
    int32_t dt, dh, dw; /* source coordinate offset of this tap                              */
    int32_t wofs;       /* element offset of this tap's K segment inside a weight row        */
    int32_t c0;         /* first source channel                                              */
    int32_t nch;        /* channels (multiple of 8)                                          */
    int32_t pad0, pad1;
} GenieTap;

/* One (dt, dh, channel block of 64) step of a kw-triple convolution (conv_igemm3.hip): the three taps dw = -1, 0, +1 share
 * one staged activation tile.  a_delta = ((dt * Hs + dh) * Ws) * Cs + c0 + 64 * block (elements); wofs{0,1,2} = element offset
 * of the K segment of tap dw = -1 / 0 / +1 (same channel block) inside a weight row. */
typedef struct GenieTriStep {
    int32_t a_delta, dt, dh;
    int32_t wofs0, wofs1, wofs2;
    int32_t rows_per_dt; /* > 0: the table is sorted by dt and every dt value owns this many CONSECUTIVE rows -- a row tile that lies   */
    int32_t dt_min;      /* inside one frame t then runs only the rows with 0 <= t + dt < T (the others multiply the zero padding);   */
                         /* dt of row i = dt_min + i / rows_per_dt.  0: no such structure, every row is executed (ABI <= 8 tables)    */
} GenieTriStep;

typedef struct GenieConvDesc {
    const void* src;      /* CL bf16 (N, Ts, Hs, Ws, Cs)                                       */
    const void* wgt;      /* bf16 rows, K-contiguous segments (see genie_pack_weight)          */
    void* dst;            /* CL bf16 (N, Td, Hd, Wd, Cd)                                       */
    const void* resid;    /* optional CL bf16, same geometry as dst: added in the epilogue     */
    const float* bias;    /* optional fp32 [Ncols], indexed by the NATURAL weight row          */
    const GenieTap* taps; /* DEVICE pointer, ntaps entries                                     */
    int32_t ntaps;
    int32_t nk;           /* sum over taps of ceil(nch / 64) (ignored when small_c)            */
    int32_t small_c;      /* 1: Cs in {8,16,32}, taps packed back to back inside 64-wide K chunks;
                             needs nch == Cs, c0 == 0 and wofs == tap * Cs for every tap        */
    int32_t N, Ts, Hs, Ws, Cs;
    int32_t To, Ho, Wo;   /* row grid                                                          */
    int32_t st, sh, sw;   /* source step per row-grid step                                     */
    int32_t Ncols;        /* GEMM N = number of weight rows used                               */
    int32_t w_row_stride; /* elements between weight rows                                      */
    int32_t perm_c, perm_f; /* weight row of column n = (n % perm_c) * perm_f + n / perm_c; perm_f<=1: identity */
    int32_t Td, Hd, Wd, Cd;
    int32_t dmt, dmh, dmw; /* dest coordinate = row-grid coordinate * dm + do (+ sub-pixel)    */
    int32_t dot, doh, dow;
    int32_t shuf_c, shuf_q, shuf_r; /* column n -> sub-pixel (p,q,r) = unravel(n / shuf_c, (P, shuf_q, shuf_r)), channel n % shuf_c;
                                       no shuffle: shuf_c >= Ncols, shuf_q = shuf_r = 1          */
    int32_t act;          /* epilogue activation: 0 none, 1 SiLU                               */
    void* splitk_ws;      /* optional fp32 scratch: enables split-K when the output has too few tiles to fill the chip */
    int64_t splitk_ws_bytes;
    /* optional kw-triple schedule (stride-1, same-size convs whose taps come as dw = -1, 0, +1 triples over whole 64-channel
     * blocks): DEVICE pointer to n_tri_steps entries covering the same K range as `taps`.  tri_bm: 0 = choose the row tile,
     * 128 / 256 = force it, -1 = ignore the schedule; tri_flags (debug / A-B timing) bit 0: drain every barrier instead of
     * counted waits, bit 1: one-tile-ahead schedule for the 256-row tile instead of the deep-prefetch one; bits 2-5: timing ablations
     * (wrong results); bit 6: deep-prefetch schedule without the pre-read; bit 7: persistent form of the 256-row kernel (one
     * block per CU walks its tiles); bit 8: 4 waves of 128 x 64 instead of 8 of 64 x 64 (both measured, neither faster). */
    const GenieTriStep* tri_steps;
    int32_t n_tri_steps;
    int32_t tri_bm;
    int32_t tri_flags;
    int32_t pointwise;    /* 1: plain GEMM rows -- one tap with dt = dh = dw = 0, c0 = 0, wofs = 0 and nch = 64 * nk (a 1x1x1 stride-1
                             convolution or a Linear layer): eligible for the persistent GEMM kernel (conv_gemm.hip) */
    /* Optional GroupNorm work in the epilogue (reference video.py:578-613: every conv of a residual block is followed / preceded by a
     * one-group GroupNorm + SiLU).  Only for a plain destination (no shuffle, dm = 1, do = 0) whose samples are whole multiples of 256
     * rows; the library applies it when the kernel it picks has the epilogue for it and says so through
     * genie_last_conv_gn_fused() -- the caller runs the stand-alone pass (genie_groupnorm_fwd / _bwd) for whatever was not fused.
     *   gn_sums  : fp64 [N][2], ACCUMULATED (zero it first): sum and sum of squares of the bf16 outputs of sample n -- the statistics of
     *              a one-group GroupNorm over this tensor (genie_groupnorm_fwd_from_sums);
     *   gnb_x    : this call computes the gradient of a GroupNorm(+activation) OUTPUT (it is a backward-data pass) and gnb_x is that
     *              GroupNorm's INPUT (bf16, destination layout); gnb_gamma / gnb_beta fp32 [C] or NULL, gnb_mean / gnb_rstd fp32 [N],
     *              gnb_act 0 none / 1 SiLU / 2 LeakyReLU(0.01); gnb_part fp32 [N][gnb_nblk][Cd][2] receives, per 256-row tile and channel,
     *              (sum dz, sum dz * xhat), dz = dst * act'(xhat * gamma + beta) -- the input of genie_groupnorm_bwd_from_part;
     *              gnb_nblk = rows per sample / 256. */
    void* gn_sums;
    const void* gnb_x;
    const float* gnb_gamma;
    const float* gnb_beta;
    const float* gnb_mean;
    const float* gnb_rstd;
    float* gnb_part;
    int32_t gnb_act;
    int32_t gnb_nblk;
} GenieConvDesc;

int genie_conv_igemm(const GenieConvDesc* desc, void* stream);

/* Which kernel the calling thread's last genie_conv_igemm / genie_conv_wgrad launched (profiling aid). */
#define GENIE_VARIANT_IGEMM_128 0
#define GENIE_VARIANT_IGEMM_128_SMALLC 1
#define GENIE_VARIANT_IGEMM_32 2
#define GENIE_VARIANT_IGEMM_32_SMALLC 3
#define GENIE_VARIANT_IGEMM3_128 4
#define GENIE_VARIANT_IGEMM3_256 5
#define GENIE_VARIANT_GEMM_PW 6
#define GENIE_VARIANT_IGEMM3_256_SPLITK 7
#define GENIE_VARIANT_WGRAD_128 8
#define GENIE_VARIANT_WGRAD_128x32 9
#define GENIE_VARIANT_WGRAD_32x128 10
#define GENIE_VARIANT_WGRAD3 11
#define GENIE_VARIANT_IGEMM3_WIDE 12     /* igemm3w_kernel: 256 x 256 tile, layers with >= 256 output channels */
#define GENIE_VARIANT_WGRAD_PW 13        /* wgrad_pw_kernel: pointwise weight gradient, 256 x 256 tile */
#define GENIE_VARIANT_WGRAD3_LEAN 14     /* wgrad3l_kernel: kw-triple weight gradient, buffer-addressed LDS-DMA + scalar bookkeeping */
#define GENIE_VARIANT_IGEMM3_H 15        /* igemm3h_kernel: 256 x 128 tile, 32-channel K-tiles, two blocks per CU (<= 128 output columns) */
int genie_last_conv_variant(void);
/* GroupNorm work the calling thread's last genie_conv_igemm did in its epilogue: bit 0 = gn_sums, bit 1 = gnb_part. */
int genie_last_conv_gn_fused(void);

/* Weight gradient: dW[row(n)][tap][c] += sum_m DY[dpix(m)][n'] * SRC[pix(m)*step + off_tap][c]
 * replaces: the weight/bias gradient of nn.Conv3d computed by autograd for every conv above.
 * Accumulates with fp32 atomics into a strided fp32 gradient (any layout: pass element strides). */
typedef struct GenieWgradDesc {
    const void* src;      /* CL bf16 (N, Ts, Hs, Ws, Cs): the conv INPUT                        */
    const void* dy;       /* CL bf16 (N, Td, Hd, Wd, Cd): gradient of the conv OUTPUT           */
    float* dw;            /* fp32, element (cout, tap, cin) at cout*s_cout + tap*s_tap + cin*s_cin */
    float* dbias;         /* optional fp32 [Cout]                                               */
    const GenieTap* taps; /* DEVICE pointer: dt/dh/dw per tap (wofs/c0/nch unused)              */
    int32_t ntaps;
    int32_t N, Ts, Hs, Ws, Cs, Cin;
    int32_t To, Ho, Wo, st, sh, sw;
    int32_t Td, Hd, Wd, Cd, Cout;
    int32_t dmt, dmh, dmw, dot, doh, dow;
    int32_t shuf_c, shuf_q, shuf_r;
    int64_t s_cout, s_tap, s_cin;
    int32_t split_k;      /* 0 = choose */
    int32_t tri_mode;     /* 1: the taps are ordered as kw-triples (dw = -1, 0, +1 consecutive, same dt / dh) of a stride-1,
                             same-size convolution -> conv_wgrad3.hip may take it (2: must, whatever the problem size); 0: generic kernel */
    int32_t pointwise;    /* 1: ONE tap with dt = dh = dw = 0 (a 1x1x1 stride-1 convolution or a Linear layer): with >= 256 channels on
                             both sides and >= 192 output tiles of 256 x 256 (the vocabulary head) the transposing-read GEMM of
                             conv_wgrad_pw.hip takes it; 2: that kernel whatever the tile count */
    int32_t dy_unshuffled; /* 1 (ABI 8): the conv is a depth-to-space-time upsample conv (shuf_c = final channel count < Cout) but `dy` has ALREADY
                             been un-shuffled into the conv's own row grid (Td, Hd, Wd) = (To, Ho, Wo), dm* = 1, with sub-pixel-major channels
                             co' = sub * shuf_c + ch (genie_unshuffle_cl, the tensor the backward-data pass consumes as well).  The gradient
                             row of co' is the natural weight row (co' % shuf_c) * (Cout / shuf_c) + co' / shuf_c.  Served by the lean
                             kw-triple kernel only (GENIE_ERR_ARG when its preconditions do not hold: tri_mode != 0, W in {8, 16, 32, 64},
                             H * W a multiple of 64, Cin, Cout >= 64) */
    int32_t row_px;        /* (ABI 13) 0, or: `src` and `dy` are a W-WINDOW of wider tensors -- the memory holds row_px pixels per image row and the
                             problem covers columns [px0, px0 + Ws) of them, zero-padded at the window's edges like a whole image.  Lean kw-triple kernel
                             only, Ws == Wo == Wd == 64 (GENIE_ERR_ARG otherwise).  Used to run 128-pixel-wide layers (BASELINE configs[4]) as two
                             64-column windows; the caller adds the two seam terms (dy column 63 x column 64 at kw = +1, dy 64 x 63 at kw = -1) */
    int32_t px0;
} GenieWgradDesc;

int genie_conv_wgrad(const GenieWgradDesc* desc, void* stream);

/* Strided fp32 (R, J, K) -> dense bf16 [R][J][roundup8(K)], zero padded; optional K permutation
 * natural k = (k' % perm_c) * perm_f + k' / perm_c.  Builds the forward pack (R=cout, J=tap, K=cin) and the
 * transposed pack for dgrad (R=cin, J=tap, K=cout) from an nn.Conv3d weight in any strides. */
int genie_pack_weight(const float* src, void* dst, int R, int J, int K, int64_t sR, int64_t sJ, int64_t sK,
                      int perm_c, int perm_f, void* stream);
int genie_cast_f32_to_bf16(const float* src, void* dst, int64_t numel, void* stream);

/* All transposed (backward-data) weight packs of a model in ONE launch, from the bf16 mirror of the parameter arena
 * (genie_adamw_step_mirror): job i turns bf16 [R = cout][J = tap][K = cin] at src + src_off (the channels_last_3d layout of an
 * nn.Conv3d weight; K % 8 == 0) into bf16 [cin][tap][roundup8(cout)] at dst + dst_off, optionally in depth-to-space column
 * order (perm_c, perm_f as in genie_pack_weight).  tiles_r = ceil(R / 64), tiles_k = ceil(K / 64); first_block = sum over the
 * earlier jobs of tiles_r * J * tiles_k (the table lives in device memory, ascending); total_blocks = the grand total. */
typedef struct GeniePackJob {
    int64_t src_off, dst_off;
    int32_t R, J, K;
    int32_t perm_c, perm_f;
    int32_t tiles_r, tiles_k;
    int32_t first_block;
} GeniePackJob;
int genie_pack_transpose_batched(const GeniePackJob* jobs_dev, int njobs, int total_blocks, const void* src_bf16, void* dst_bf16,
                                 void* stream);

/* ---------------------------------------------------------------------------------------------
 * GroupNorm (+ optional adaptive scale/shift, + optional SiLU), forward and backward (norm.hip).
 * replaces: nn.GroupNorm + nn.SiLU (video.py:578-579, 612-613; tokenizer.py descs 'group_norm','silu'),
 *           AdaptiveGroupNorm.forward (norm.py:55-69), ForwardBlock's GroupNorm (misc.py:92).
 *   y = act( ((x - mean[n,g]) * rstd[n,g] * gamma[c] + beta[c]) * ada_scale[n,c] + ada_shift[n,c] )
 * Workspaces are caller-owned device fp32 buffers; sizes from genie_groupnorm_ws_floats().
 * ------------------------------------------------------------------------------------------- */
int64_t genie_groupnorm_ws_floats(int N, int C, int G);
int genie_groupnorm_fwd(const void* x, void* y, int N, int64_t npix, int C, int cpitch, int G, const float* gamma,
                        const float* beta, const float* ada_scale, const float* ada_shift, float eps, int act,
                        float* mean, float* rstd, float* ws, void* stream);
/* dgamma/dbeta (fp32 [C]) are ACCUMULATED; dada_scale/dada_shift (fp32 [N][C]) are written. */
int genie_groupnorm_bwd(const void* x, const void* dy, void* dx, int N, int64_t npix, int C, int cpitch, int G,
                        const float* gamma, const float* beta, const float* ada_scale, const float* ada_shift, int act,
                        const float* mean, const float* rstd, float* dgamma, float* dbeta, float* dada_scale,
                        float* dada_shift, float* ws, void* stream);
/* The same two passes when a convolution already did the statistics / the backward reduction in its epilogue (GenieConvDesc.gn_sums,
 * gnb_part): one group, no adaptive scale / shift.  fwd: mean / rstd (fp32 [N], written) from the fp64 sums, then the apply pass;
 * bwd: parameter gradients (accumulated) + coefficients from the per-tile partials, then the apply pass.  ws: fp32 scratch of
 * genie_groupnorm_bwd_from_part_ws_floats(N). */
/* tests: 1 when a clip barrier of the one-pass forward (norm.hip) ever gave up waiting; synchronises the device */
int genie_gn_fused_error(void);
int genie_groupnorm_fwd_from_sums(const void* x, void* y, int N, int64_t npix, int C, int cpitch, const float* gamma, const float* beta,
                                  float eps, int act, float* mean, float* rstd, const double* sums, void* stream);
int64_t genie_groupnorm_bwd_from_part_ws_floats(int N);
int genie_groupnorm_bwd_from_part(const void* x, const void* dy, void* dx, int N, int64_t npix, int C, int cpitch, const float* gamma,
                                  const float* beta, int act, const float* mean, const float* rstd, float* dgamma, float* dbeta,
                                  const float* part, int nblk, float* ws, void* stream);

/* ---------------------------------------------------------------------------------------------
 * BlurPooling3d with num_groups = 1 (elementwise.hip).   replaces: F.conv3d with the expanded Pascal kernel, video.py:516-534
 * (every output channel = strided blur of the SUM over the input channels -- the reference's dense-conv behaviour).
 * dims = {N, C, T, H, W} of the input; kernel / stride / pad = {t, h, w}; taps: device fp32 [kt*kh*kw];
 * ws: device fp32, max(N*T*H*W, N*To*Ho*Wo) floats.  Output geometry (To, Ho, Wo) = floor((size + 2 pad - k) / stride) + 1.
 * ------------------------------------------------------------------------------------------- */
int genie_blur_pool3d_fwd(const void* x_cl, int cpitch, const int64_t* dims, const float* taps, const int* kernel, const int* stride,
                          const int* pad, void* out_cl, int out_channels, int out_pitch, float* ws, void* stream);
int genie_blur_pool3d_bwd(const void* dy_cl, int out_channels, int out_pitch, const int64_t* dims, const float* taps,
                          const int* kernel, const int* stride, const int* pad, void* dx_cl, int cpitch, float* ws, void* stream);

int genie_silu_fwd(const void* x, void* y, int64_t numel, void* stream);
int genie_silu_bwd(const void* x, const void* dy, void* dx, int64_t numel, void* stream);
/* LeakyReLU over a CL buffer.  replaces: nn.LeakyReLU in ImageResidualBlock (image.py:118-131) and FrameDiscriminator.to_logits
 * (discriminator.py:91), forward and backward.  (GroupNorm fuses it as act = 2, slope 0.01: genie_groupnorm_fwd / _bwd.) */
int genie_leaky_relu_fwd(const void* x, void* y, int64_t numel, float slope, void* stream);
int genie_leaky_relu_bwd(const void* x, const void* dy, void* dx, int64_t numel, float slope, void* stream);
/* GELU (exact erf form) over a CL buffer.  replaces: nn.GELU, the activation between the layers of ForwardBlock (misc.py:78,94-98) when
 * SpaceTimeAttention is built with hid_dim (attention.py:429-438), forward and backward.  (ABI 10) */
int genie_gelu_fwd(const void* x, void* y, int64_t numel, void* stream);
int genie_gelu_bwd(const void* x, const void* dy, void* dx, int64_t numel, void* stream);
int genie_add(const void* a, const void* b, void* y, int64_t numel, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Lookup-free quantisation (lfq.hip).   replaces: LookupFreeQuantization.forward, quantization.py:77-133
 * z: (ntok, pitch) rows holding num_codebook*codebook_dim values; idx: int64 (ntok, num_codebook),
 * MSB-first bit order; quant = sign(z) (may be NULL).
 * ------------------------------------------------------------------------------------------- */
int genie_lfq_quantize(const void* z, int dtype, int64_t ntok, int num_codebook, int codebook_dim, int64_t pitch,
                       void* quant, int64_t* idx, void* stream);

/* Training loss of LFQ, forward AND gradient in one sweep (quantization.py:116-131).
 * loss4 = {total, per-token entropy, entropy of the mean distribution, commit MSE}; dz_loss = d total / d z as
 * fp32 dense [ntok][num_codebook * codebook_dim].  ws: genie_lfq_loss_ws_floats() floats. */
int64_t genie_lfq_loss_ws_floats(int64_t ntok, int num_codebook, int codebook_dim);
int genie_lfq_loss(const void* z, int dtype, int64_t ntok, int num_codebook, int codebook_dim, int64_t pitch, float beta,
                   float commit_weight, float entropy_weight, float diversity_weight, float* ws, float* loss4, float* dz_loss,
                   void* stream);
/* out = dy (straight-through, may be NULL) + (*grad_loss) * dz_loss (may be NULL), in z's dtype/pitch; pad columns zeroed */
int genie_lfq_bwd(const void* dy, const float* dz_loss, const float* grad_loss, void* out, int dtype, int64_t ntok, int width,
                  int64_t pitch, void* stream);

/* Masked token cross-entropy over bf16 logits rows (lfq.hip).   replaces: logits[mask] + F.cross_entropy in
 * DynamicsModel.compute_loss, dynamics.py:92-97.  logits: bf16 [nrow][pitch >= V]; target: int64 [nrow]; mask: uint8 [nrow] or
 * NULL (= all rows).  fwd: row_lse[r] = logsumexp(row r) for masked rows, *loss_sum += sum over masked rows of
 * (lse - logit[target]) (caller zeroes it and divides by the masked-row count).  bwd: dlogits = (softmax - onehot) * *scale on
 * masked rows (scale = upstream gradient / count, device scalar), zeros elsewhere. */
int genie_masked_ce_fwd(const void* logits_bf16, int64_t pitch, int64_t nrow, int V, const int64_t* target, const unsigned char* mask,
                        float* row_lse, float* loss_sum, void* stream);
int genie_masked_ce_bwd(const void* logits_bf16, int64_t pitch, int64_t nrow, int V, const int64_t* target, const unsigned char* mask,
                        const float* row_lse, const float* scale, void* dlogits_bf16, int64_t dpitch, void* stream);

/* Fused vocabulary head + masked token cross-entropy (linear_ce.hip; ABI 11).   replaces: `self.head` Linear(D -> V) of
 * DynamicsModel.forward (dynamics.py:44, :62) TOGETHER WITH logits[mask] + F.cross_entropy of compute_loss (dynamics.py:89-97) -- the
 * `linear_cross_entropy(h, W, b, target, mask)` of SURVEY.md section 8(b).  The M x V logits are never written: both directions are
 * flash-attention-shaped sweeps with D-wide accumulators.
 *   h: bf16 [M][h_pitch >= D] (the gathered trunk rows); W: bf16 [V][w_pitch >= D] (the head's forward weight pack); bias: fp32 [V] or NULL;
 *   target: int64 [M]; valid: uint8 [M] or NULL (= every row counts).  D in {64, 128, 256, 512}; genie_linear_ce_supported() says whether
 *   a shape is taken (otherwise use genie_conv_igemm + genie_masked_ce_*).
 * fwd: row_lse[m] = logsumexp_v(h[m] . W[v] + b[v]) (fp32 logits);  *loss_sum += sum over valid rows of (lse - logit[target]) (caller zeroes
 *      it and divides by the valid-row count; a target outside [0, V) poisons it with NaN);  row_e: [M rounded up to 64] scratch the backward
 *      consumes (-lse, -inf for rows that are off);  dh_f32 (may be NULL = no gradient wanted): [M][D] fp32, softmax(logits) W - W[target],
 *      i.e. d loss_sum / d h, zeros for rows that are off.  ws: genie_linear_ce_ws_floats(M, D, V, dh_f32 != NULL) floats.
 * bwd: dh_bf16[m] = dh_f32[m] * *scale (skipped when both are NULL);  dW [V][D] fp32 += *scale * (softmax - onehot)^T h (skipped when NULL),
 *      dbias [V] += *scale * column sums (may be NULL).  scale = upstream gradient / valid-row count, a device scalar. */
int genie_linear_ce_supported(int64_t M, int D, int64_t V, int64_t h_pitch, int64_t w_pitch);
int64_t genie_linear_ce_ws_floats(int64_t M, int D, int64_t V, int with_grad);
int genie_linear_ce_fwd(const void* h_bf16, int64_t h_pitch, int64_t M, int D, const void* w_bf16, int64_t w_pitch, int64_t V,
                        const float* bias, const int64_t* target, const unsigned char* valid, float* ws, int64_t ws_floats,
                        float* row_lse, float* row_e, float* loss_sum, float* dh_f32, void* stream);
int genie_linear_ce_bwd(const void* h_bf16, int64_t h_pitch, int64_t M, int D, const void* w_bf16, int64_t w_pitch, int64_t V,
                        const float* bias, const int64_t* target, const float* row_e, const float* scale, const float* dh_f32,
                        void* dh_bf16, int64_t dh_pitch, float* dW, float* dbias, void* stream);

/* uint8 video frames [npix = N T H W][C] (what a decoder / a .npy frame array holds) -> CL bf16 [npix][cpitch], value / 255, pad channels zero
 * (elementwise.hip; ABI 11).   replaces: `video / 255.` + rearrange 't h w c -> c t h w' of Platformer2D.load_video_slice (genie/module/data.py:218-231)
 * done on the host, followed by the model-boundary layout conversion.  Same numbers as that path (fp32 quotient, one rounding to bf16). */
int genie_u8_frames_to_cl(const void* src_u8, int64_t npix, int C, void* dst_cl, int cpitch, void* stream);

/* Embedding lookup and its sparse backward (elementwise.hip; ABI 12).   replaces: nn.Embedding tok_emb / act_emb of DynamicsModel
 * (genie/dynamics.py:31-38, 52-55) and autograd's index_add_ scatter behind it.  fwd: out[n][:] = weight[idx[n]][:] (fp32 [V][D], D % 4 == 0; an index
 * outside [0, V) poisons its row with NaN -- nn.Embedding raises).  bwd: grad[idx[n]][:] += dy[n][:] (dy fp32 or bf16, dense [N][D]; grad fp32 [V][D],
 * accumulated with fp32 atomics, one per distinct index per 64-row chunk and column; out-of-range rows are skipped). */
int genie_embedding_fwd(const int64_t* idx, const float* weight, float* out, int64_t N, int D, int64_t V, void* stream);
int genie_embedding_bwd(const int64_t* idx, const void* dy, int dy_dtype, float* grad, int64_t N, int D, int64_t V, void* stream);

/* Skinny linears in fp32 arithmetic (linear_small.hip; ABI 12): y = x W^T + bias with min(in K, out N) <= 32.   replaces: F.linear of
 * AdaptiveGroupNorm.std / .avg (genie/module/norm.py:55-69), LookupFreeQuantization.proj_inp / proj_out (quantization.py:52-58), Adapter.to_k / to_v on a
 * conditioning vector (attention.py:128-129), LatentAction.to_act (action.py:83-90).  x: [M][x_pitch >= K] fp32 / bf16; W: fp32, W[n][k] at n * w_sn +
 * k * w_sk (the backward-data pass is the same call with dy as x and the strides exchanged); bias: fp32 [N] or NULL; y: [M][y_pitch >= N] fp32 / bf16.
 * K > 32 walks the reduction axis in register-resident weight slices; more than one slice (K > 1024 / 512 / 256 for N <= 10 / 16 / 32) needs
 * genie_linear_small_ws_floats(M, K, N) floats of scratch (partials summed in a fixed order).  wgrad: dW[n * w_sn + k * w_sk] += sum_m dy[m][n] x[m][k],
 * dbias[n] += sum_m dy[m][n] (NULL: skipped), rows split over workgroups, partials in ws (genie_linear_small_wgrad_ws_floats) summed in a fixed order:
 * no atomics, bit-reproducible. */
int64_t genie_linear_small_ws_floats(int64_t M, int K, int N);
int genie_linear_small_fwd(const void* x, int x_dtype, int64_t x_pitch, int64_t M, int K, const float* W, int64_t w_sn, int64_t w_sk,
                           const float* bias, void* y, int y_dtype, int64_t y_pitch, int N, float* ws, int64_t ws_floats, void* stream);
int64_t genie_linear_small_wgrad_ws_floats(int64_t M, int N, int K);
int genie_linear_small_wgrad(const void* dy, int dy_dtype, int64_t dy_pitch, const void* x, int x_dtype, int64_t x_pitch, int64_t M, int N, int K,
                             float* dW, int64_t w_sn, int64_t w_sk, float* dbias, float* ws, int64_t ws_floats, void* stream);

/* Guard-page device allocations for the memory-safety harness (guard.hip; tests/guard.py).  *ptr: `bytes` bytes of device memory whose last byte
 * (up to 15 bytes of alignment slack) is the last byte of a mapping with an UNMAPPED page on either side -- an out-of-bounds access of a
 * kernel in either direction is a GPU page fault on every run, not only when the caching allocator happens to leave a hole there.
 * Test infrastructure (ABI 11); nothing on the product path calls these. */
int genie_guard_alloc(int64_t bytes, void** ptr, void** handle);
int genie_guard_free(void* handle);

/* MaskGIT sampling step (maskgit.hip).   replaces: softmax(logits / temp) + torch.multinomial + gather (confidence) and
 * topk + gather + scatter_ of DynamicsModel.generate, dynamics.py:138-158.  The multinomial draw is an inverse-CDF draw from an
 * INJECTED uniform per row (device RNG streams are not reproducible across devices; oracle/genie_oracle.py::sample_from_uniform):
 *   p = softmax(row / temp) in fp32;  pred = #{ j : cumsum_f64(p)_j <= u * sum(p) } clamped to V - 1;  conf = p[pred].
 * Row r of the (rows, V) logits lives at element offset (r / rows_per_sample) * sample_stride + (r % rows_per_sample) * pitch
 * (so the last-frame slice logits[:, -1] of a (B, T, h, w, V) tensor is addressed in place).  dtype GENIE_BF16 or GENIE_F32. */
int genie_maskgit_sample(const void* logits, int dtype, int64_t rows, int64_t rows_per_sample, int64_t sample_stride, int64_t pitch,
                         int64_t V, const float* u, float temp, int64_t* pred, float* conf, void* stream);
/* per sample b < batch: the k positions with the largest conf among those with mask != 0 (ties: lower index) get
 * code[b][i] = pred[b][i], mask[b][i] = 0.  conf / pred / code / mask: [batch][n], n <= 32768. */
int genie_maskgit_paint(const float* conf, const int64_t* pred, int64_t batch, int64_t n, int64_t k, int64_t* code,
                        unsigned char* mask, void* stream);

/* ---------------------------------------------------------------------------------------------
 * HBM-bound CausalConv3d with <= 4 input channels and 128 output channels, 3x3x3, stride 1 (conv_narrow.hip).
 * replaces: the tokenizer's stem CausalConv3d(3 -> 128) (video.py:154-192 as used by tokenizer.py:25) and the backward-data pass of
 *           its head conv CausalConv3d(128 -> 3) (tokenizer.py:172) -- 16.8 MB of output per 0.39 MB of input and clip.
 * src: CL [N][T][H][W][src_pitch] (the first 4 channels of a pixel are read); dst: CL [N][T][H][W][dst_pitch >= 128].
 * wpack: bf16 [128][112], k = tap * 4 + c with tap = ((dt - t_lo) * 3 + (dh + 1)) * 3 + (dw + 1); k = 108 / 109 hold a bias as a
 * bf16 hi / lo pair (0 for none).  out[p][co] = sum_tap sum_c src[p + (dt, dh, dw)][c] * w[co][tap][c] (+ bias), zero outside.
 * t_lo = - (causal front padding): -2 for the stem's forward, 0 for the head's backward-data pass, -1 for a symmetric conv.
 * ------------------------------------------------------------------------------------------- */
int genie_conv_narrow_in(const void* src_cl, int src_pitch, const void* wpack, void* dst_cl, int dst_pitch, int N, int T, int H, int W,
                         int t_lo, void* stream);
/* The other direction: 128 input channels -> cout <= 3 output channels (the head conv CausalConv3d(128 -> 3) forward, tokenizer.py:172).
 * src: CL [N][T][H][W][128]; dst: CL [N][T][H][W][8] (whole 16-byte pixels are written, channels >= cout zero); bias fp32 [cout] or NULL.
 * wpack: bf16 [16][1152], row = 4 * (dt - t_lo) + co (every other row zero), k = ((dh + 1) * 3 + (dw + 1)) * 128 + ci.  t_lo in [-2, 0];
 * W a multiple of 32, H * W * 256 < 2^32.  W <= 64: the kernel with the column tap in the MFMA rows (a wave owns an image row); wider images:
 * the 32-column-block kernel (GENIE_NARROW_OUT_CUT=1 forces it). */
int genie_conv_narrow_out(const void* src_cl, const void* wpack, const float* bias, void* dst_cl, int N, int T, int H, int W, int cout,
                          int t_lo, void* stream);
/* Weight gradients of the two narrow convolutions above (stem 3 -> 128: big = output gradient, small = input, t_lo = -2, ones = 1;
 * head 128 -> 3: big = input, small = output gradient, taps flipped by the caller, t_lo = 0): one pass over the 128-channel tensor.
 * G: fp32 [128][128], ACCUMULATED (zero it first): G[ch][tap * 4 + c] = sum_pixels big[p][ch] * small[p + (t_lo + dt, dh - 1, dw - 1)][c]
 * for tap = (dt * 3 + dh) * 3 + dw, c < 4 (channels >= small's real count read as stored: keep its pad channels zero); with ones = 1
 * column 108 holds sum_pixels big[p][ch] (the bias gradient of the stem).  W in {32, 64, 128} (W = 32: even H). */
int genie_conv_narrow_wgrad(const void* big_cl, const void* small_cl, int small_pitch, float* G, int N, int T, int H, int W, int t_lo,
                            int ones, void* stream);
/* The same pass accumulated straight into the parameter gradients (no G, no scatter afterwards): dW fp32 with the shape of the reference's
 * nn.Conv3d weight -- stem (stem = 1): (128, cs, 3, 3, 3), dbias [128] (or NULL); head (stem = 0): (cs, 128, 3, 3, 3) with the taps un-flipped
 * here, dbias [cs] = the plain sum of the output gradient over all pixels (or NULL).  cs = real channels of the narrow tensor (<= 4).
 * w_channels_last: 0 = dW dense in (co, ci, kt, kh, kw) order, 1 = dense in (co, kt, kh, kw, ci) order (torch.channels_last_3d). */
int genie_conv_narrow_wgrad_acc(const void* big_cl, const void* small_cl, int small_pitch, float* dW, float* dbias, int N, int T, int H, int W,
                                int t_lo, int stem, int cs, int w_channels_last, void* stream);
/* The same for a wide side of a multiple of 128 channels (ABI 12; LatentAction.proj_in 3 -> 256 / proj_out 256 -> 3, genie/action.py:60-70): one call per
 * 128-channel slab [wide0, wide0 + 128) of the big_pitch-channel tensor; dW / dbias are the whole parameter gradients (wide_total channels on the wide side). */
int genie_conv_narrow_wgrad_wide(const void* big_cl, int big_pitch, int wide0, int wide_total, const void* small_cl, int small_pitch, float* dW,
                                 float* dbias, int N, int T, int H, int W, int t_lo, int stem, int cs, int w_channels_last, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Losses and optimiser (elementwise.hip).
 * replaces: F.mse_loss (tokenizer.py:364, action.py:166) and torch.optim.AdamW (tokenizer.py:437-442).
 * ------------------------------------------------------------------------------------------- */
int genie_mse_fwd(const void* rec_cl, int cpitch, const void* target, int target_dtype, const int64_t* dims,
                  const int64_t* strides, float* partial_ws /* >= 1024 floats */, float* loss, void* stream);
int genie_mse_bwd(const void* rec_cl, int cpitch, const void* target, int target_dtype, const int64_t* dims,
                  const int64_t* strides, const float* grad_loss /* device scalar or NULL (=1) */, void* drec_cl,
                  void* stream);
int genie_adamw_step(float* p, float* g, float* m, float* v, int64_t numel, float lr, float beta1, float beta2, float eps,
                     float weight_decay, int step, float grad_scale, int zero_grad, void* stream);
/* the same, additionally writing a bf16 image of the updated parameters (p_bf16[numel]): for channels_last_3d Conv3d weights
 * with in_channels % 8 == 0 that image IS the forward weight pack, so no per-step repacking is needed */
int genie_adamw_step_mirror(float* p, float* g, float* m, float* v, void* p_bf16, int64_t numel, float lr, float beta1, float beta2,
                            float eps, float weight_decay, int step, float grad_scale, int zero_grad, void* stream);
/* capture-safe form (hipGraph replay): nothing that changes from step to step is a launch argument.  `state` = 6 device floats:
 * [0] the step count so far (int32 bits; incremented by the call), [1] learning rate, [2] weight decay (both written by the host, e.g. a
 * scheduler, between replays), [3..5] scratch for the step's coefficients.  p_bf16 may be NULL (no mirror). */
int genie_adamw_step_graph(float* p, float* g, float* m, float* v, void* p_bf16, int64_t numel, float* state, float beta1, float beta2,
                           float eps, float grad_scale, int zero_grad, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Space-time transformer attention (attention.hip).
 * replaces: RotaryEmbedding + nn.LayerNorm + Adapter head split + F.scaled_dot_product_attention + head merge in
 *           Attention.forward (attention.py:199-239) and the rearrange/pack/unpack of SpatialAttention.forward
 *           (:279-307) / TemporalAttention.forward (:347-371), forward and backward.
 *
 * Token rows are [C] bf16; a token of sequence `seq` at position `pos` lives at element offset
 *     (seq / map[0]) * map[1] + (seq % map[0]) * map[2] + pos * map[3]        map = {n_inner, stride_outer, stride_inner, pos_stride}
 * cos_sin: fp32 [npos][C], pairs (cos, sin) of angle pos * freq_i in slots (2i, 2i+1); position of token row r is
 * (r / pos_div) % pos_mod.  stats: fp32 [ntok][2] (mean, rstd of the rotated row) saved for backward.
 * ------------------------------------------------------------------------------------------- */
int genie_rotary_layernorm_fwd(const void* x, void* u, int64_t ntok, int C, int64_t pitch, const float* cos_sin, int64_t pos_div,
                               int pos_mod, const float* gamma, const float* beta, float eps, float* stats, void* stream);
/* dx = d/dx [LN(rot(x))](du) (+ dres); dgamma / dbeta (fp32 [C]) are ACCUMULATED */
int genie_rotary_layernorm_bwd(const void* x, const void* du, const void* dres, void* dx, int64_t ntok, int C, int64_t pitch,
                               const float* cos_sin, int64_t pos_div, int pos_mod, const float* gamma, const float* stats,
                               float* dgamma, float* dbeta, void* stream);
/* out = softmax(scale * q k^T [causal]) v (+ resid); o_attn (optional) receives the same without the residual (what backward
 * needs); lse: fp32 [out tokens][nhead] (token = element offset / out_channels) */
int genie_attention_fwd(const void* q, const void* k, const void* v, const void* resid, void* out, void* o_attn, float* lse, int nseq, int nhead,
                        int d_head, int Sq, int Sk, const int64_t* q_map, const int64_t* kv_map, const int64_t* out_map, float scale,
                        int causal, int out_channels, void* stream);
/* `out` must be the attention output WITHOUT the residual (o_attn of the forward) when resid is NULL; with resid given,
 * out - resid is used (less accurate).  Self-attention (q == k == v): dq receives dQ + dK + dV.  Otherwise dq gets dQ and
 * dk / dv (addressed by dkv_map, which must not alias across sequences) get dK / dV.  D_ws: fp32 [3][out_tokens][nhead] scratch
 * (ABI 10: D = rowsum(dO * O), then lse * log2 e and -D -- the forms the exp2-domain backward kernels consume).  ALL THREE planes are
 * written for EVERY d_head >= 32 and every kernel family (the preprocess kernel does not know which family follows): a caller that still
 * allocates the ABI <= 9 size [out_tokens][nhead] gets out-of-bounds writes.  d_head 8 / 16 (attention_narrow.hip) use the first plane only. */
int genie_attention_bwd(const void* q, const void* k, const void* v, const void* out, const void* resid, const void* dO,
                        const float* lse, float* D_ws, void* dq, void* dk, void* dv, int nseq, int nhead, int d_head, int Sq, int Sk,
                        const int64_t* q_map, const int64_t* kv_map, const int64_t* out_map, const int64_t* dkv_map, float scale,
                        int causal, int out_channels, int64_t out_tokens, void* stream);

/* Backward of CONDITIONED short sequences (ABI 12; attention.hip: attn_smallx_bwd_kernel) -- temporal attention of the LatentAction decoder, whose keys /
 * values are Linear(8 -> C) of the per-clip action codes (reference attention.py:128-129, 222-223; action.py:136-160): Sq = Sk = S <= 32, d_head 32 / 64,
 * kv_map with inner stride 0 (the n_inner sequences of a clip share the rows).  The forward of this form is taken by genie_attention_fwd on its own.
 * dq: q_map addressing, bf16.  dk_f32 / dv_f32: fp32 [nseq / n_inner][S][kv_channels], ACCUMULATED with atomics (zero them first): the gradients of the
 * condition rows summed over every sequence of the clip -- genie_attention_bwd would return them per sequence.  Returns < 0 when the problem is not of this
 * form (use genie_attention_bwd then).  D_ws as in genie_attention_bwd. */
int genie_attention_bwd_cond(const void* q, const void* k, const void* v, const void* out, const void* resid, const void* dO, const float* lse,
                             float* D_ws, void* dq, float* dk_f32, float* dv_f32, int nseq, int nhead, int d_head, int S, const int64_t* q_map,
                             const int64_t* kv_map, const int64_t* out_map, float scale, int causal, int out_channels, int kv_channels,
                             int64_t out_tokens, void* stream);

/* Attention dropout (ABI 13).  Reference attention.py:225-230 hands `dropout_p=self.dropout` to F.scaled_dot_product_attention (in training AND in
 * eval: the functional form has no training switch): out = (softmax(S) o M / (1 - p)) V, M Bernoulli(1 - p) per (sequence, head, query, key).
 * Here M is a pure function of (seed, sequence, head, query, key) (csrc/attn_args.h: two rounds of a multiply-xorshift mixer over q * Sk + k, keyed per
 * (sequence, head) from the seed), evaluated in registers by the forward and by both backward kernels: no mask is stored, the backward call passes the
 * forward's (dropout_p, seed).  Arguments and contracts otherwise as genie_attention_fwd / genie_attention_bwd; lse is the softmax's, o_attn the DROPPED
 * output (so that D = rowsum(dO o o_attn)).  d_head 32 / 64 / 128 on the general MFMA kernels (the packed, conditioned and lean families take no mask:
 * dropout costs the fast paths, as it does in every flash implementation's bookkeeping); d_head 8 / 16 on the fp32 kernels of attention_narrow.hip.
 * dropout_p = 0 is the plain call.
 * torch's Philox stream cannot be reproduced (and differs between its own backends): parity with the reference is through the mask --
 * genie_attention_dropout_mask writes the decisions out, keep[((seq * nhead + head) * Sq + q) * Sk + k], for the oracle to apply. */
int genie_attention_fwd_dropout(const void* q, const void* k, const void* v, const void* resid, void* out, void* o_attn, float* lse, int nseq, int nhead,
                                int d_head, int Sq, int Sk, const int64_t* q_map, const int64_t* kv_map, const int64_t* out_map, float scale,
                                int causal, int out_channels, float dropout_p, uint64_t seed, void* stream);
int genie_attention_bwd_dropout(const void* q, const void* k, const void* v, const void* out, const void* resid, const void* dO,
                                const float* lse, float* D_ws, void* dq, void* dk, void* dv, int nseq, int nhead, int d_head, int Sq, int Sk,
                                const int64_t* q_map, const int64_t* kv_map, const int64_t* out_map, const int64_t* dkv_map, float scale,
                                int causal, int out_channels, int64_t out_tokens, float dropout_p, uint64_t seed, void* stream);
int genie_attention_dropout_mask(uint8_t* keep, int nseq, int nhead, int Sq, int Sk, float dropout_p, uint64_t seed, void* stream);

/* d_head 64 runs on register-lean kernels (attention_lean.hip: four / three waves per SIMD) where their preconditions hold.  mask: bit 0
 * forward, bit 1 backward dQ, bit 2 backward dK / dV; bit 3: reserved; bit 4: the forward's running maximum is
 * deferred (O, l rescaled only when a tile's maximum exceeds it by more than 2^8 in the exp2 domain; P <= 2^8 instead of <= 1, the row's
 * largest weight is then rounded to bf16 like every other one instead of being exactly 1); bit 5: plain grid instead of the XCD-aware one;
 * bit 6: forward blocks of four waves at every length (default: eight waves from 2048 queries on; bit-identical results); bit 7 (with bit 4):
 * sum-triggered form of the deferred maximum -- a tile is exponentiated against the running maximum as it is and redone with its exact
 * maximum only when a lane's row sum exceeds 2^8 (same bound on P; no per-tile maximum in the steady state); bit 8: self-attention through one
 * tensor (q == k == v), non-causal, 64 < S <= 1024 runs with the whole K / V sequence resident in LDS, one workgroup per (sequence, head) -- parity-tested,
 * measured 5-7 % slower than the ring kernel, off by default.
 * A negative mask only queries.  Returns the previous mask (default 151 = bits 0, 1, 2, 4, 7, or the GENIE_ATTN_LEAN environment variable).
 * Process-wide; meant for A/B timing and for tests that cover both kernel families and both maximum rules. (ABI 10) */
int genie_attention_lean_mode(int mask);

/* Resident blocks per CU of a lean kernel at its launch configuration (which: 0 forward, 1 backward dQ, 2 backward dK / dV), as the HIP
 * runtime computes it; -1 on error.  Budget: 4 / 3 / 3. */
int genie_attention_lean_occupancy(int which);

/* Debug / bring-up probes (used by tests only). */
int genie_probe_ds_read_tr16(const void* lds_image_u16_2048, const int32_t* lane_byte_addr_64, void* out_u16_64x4,
                             void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GENIE_HIP_H */
