// Attention core for narrow heads, d_head = 8 or 16 (reference genie/module/attention.py:199-239 with the blueprints the reference
// itself ships and tests: n_head 4 x d_head 16, genie/__init__.py:15-50, test/test_dynamics.py:17-25).
//
// A 16-wide head is half of one 32x32x16 MFMA k-step and a quarter of its M tile, and the models that use such heads are the
// reference's small configurations (C = 64), where a whole (sequence, head) K/V set is 8-32 KB and lives in L1/L2.  These kernels
// therefore do the arithmetic in fp32 on the VALU with one lane per (sequence, position, head):
//   forward : online softmax over the keys (one pass), out = P V (+ resid), lse = log-sum-exp of the scaled scores
//   backward: D = rowsum(dO * O);  dQ_i = scale * sum_j dS_ij K_j  (lane = query);   dK_j = scale * sum_i dS_ij Q_i,
//             dV_j = sum_i P_ij dO_i  (lane = key);   P_ij = exp(scale * s_ij - lse_i),  dS_ij = P_ij (dO_i . V_j - D_i)
// Same contracts as the MFMA kernels of attention.hip (address maps, causal = "key <= query", lse / D indexed by token row,
// self-attention fused to du = dQ + dK + dV); nothing is rounded to bf16 before the final store, and there are no atomics.
// Dropout (genie_attention_fwd_dropout / _bwd_dropout): the counter-based keep mask of attn_args.h, one evaluation per score.
#include "attn_args.h"
#include "genie_hip.h"

namespace {

template <int DH>
__device__ __forceinline__ void load_row(const bf16_t* p, float* f) {
#pragma unroll
    for (int c = 0; c < DH / 8; ++c) unpack8(*reinterpret_cast<const u32x4_t*>(p + 8 * c), f + 8 * c);
}
template <int DH>
__device__ __forceinline__ void store_row(bf16_t* p, const float* f) {
#pragma unroll
    for (int c = 0; c < DH / 8; ++c) *reinterpret_cast<u32x4_t*>(p + 8 * c) = pack8(f + 8 * c);
}
template <int DH>
__device__ __forceinline__ float dot(const float* a, const float* b) {
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < DH; ++d) s = __builtin_fmaf(a[d], b[d], s);
    return s;
}

template <int DH>
__global__ void __launch_bounds__(256) attn_narrow_fwd_kernel(const AttnArgs a) {
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    if (g >= (long long)a.nseq * a.Sq) return;
    const int seq = (int)(g / a.Sq), qi = (int)(g % a.Sq), head = blockIdx.y;
    float q[DH], o[DH];
    load_row<DH>(a.q + seq_base(a.qm, seq) + (long long)qi * a.qm.pos_stride + head * DH, q);
#pragma unroll
    for (int d = 0; d < DH; ++d) { q[d] *= a.scale; o[d] = 0.f; }
    const bf16_t* kb = a.k + seq_base(a.km, seq) + head * DH;
    const bf16_t* vb = a.v + seq_base(a.km, seq) + head * DH;
    const int kend = (a.causal && qi + 1 < a.Sk) ? qi + 1 : a.Sk;
    float m = -INFINITY, l = 0.f;
    const unsigned drop_key = attn_drop_seqkey(a.drop_key, seq, a.nhead, head);
#pragma unroll 2
    for (int kj = 0; kj < kend; ++kj) {
        float kr[DH], vr[DH];
        load_row<DH>(kb + (long long)kj * a.km.pos_stride, kr);
        load_row<DH>(vb + (long long)kj * a.km.pos_stride, vr);
        const float s = dot<DH>(q, kr);
        const float mn = fmaxf(m, s);
        const float corr = __expf(m - mn), p = __expf(s - mn);        // first key: exp(-inf) = 0
        l = __builtin_fmaf(l, corr, p);
        // dropout (genie_attention_fwd_dropout): the row sum is of the undropped weights, the output takes the kept ones (x 1 / (1 - p) below)
        const float pk = (a.drop_thr && !attn_drop_keep(drop_key, a.drop_thr, qi, kj, a.Sk)) ? 0.f : p;
#pragma unroll
        for (int d = 0; d < DH; ++d) o[d] = __builtin_fmaf(o[d], corr, pk * vr[d]);
        m = mn;
    }
    const float inv = l > 0.f ? a.drop_scale / l : 0.f;
#pragma unroll
    for (int d = 0; d < DH; ++d) o[d] *= inv;
    const long long orow = seq_base(a.om, seq) + (long long)qi * a.om.pos_stride;
    const long long ooff = orow + head * DH;
    if (a.lse) a.lse[(orow / a.C) * a.nhead + head] = m + __logf(l);
    if (a.oattn) store_row<DH>(a.oattn + ooff, o);
    if (a.resid) {
        float r[DH];
        load_row<DH>(a.resid + ooff, r);
#pragma unroll
        for (int d = 0; d < DH; ++d) o[d] += r[d];
    }
    store_row<DH>(a.out + ooff, o);
}

// lane = query: D and dQ
template <int DH>
__global__ void __launch_bounds__(256) attn_narrow_dq_kernel(const AttnBwdArgs a, float* __restrict__ D_out) {
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    if (g >= (long long)a.nseq * a.Sq) return;
    const int seq = (int)(g / a.Sq), qi = (int)(g % a.Sq), head = blockIdx.y;
    float q[DH], dO[DH], dq[DH];
    load_row<DH>(a.q + seq_base(a.qm, seq) + (long long)qi * a.qm.pos_stride + head * DH, q);
    const long long orow = seq_base(a.om, seq) + (long long)qi * a.om.pos_stride;
    const long long ooff = orow + head * DH;
    load_row<DH>(a.dO + ooff, dO);
    float D;
    {
        float ov[DH];
        load_row<DH>(a.out + ooff, ov);
        if (a.resid) {
            float r[DH];
            load_row<DH>(a.resid + ooff, r);
#pragma unroll
            for (int d = 0; d < DH; ++d) ov[d] -= r[d];
        }
        D = dot<DH>(dO, ov);
    }
    const long long tok = orow / a.C;
    D_out[tok * a.nhead + head] = D;
    const float lse = a.lse[tok * a.nhead + head];
#pragma unroll
    for (int d = 0; d < DH; ++d) { q[d] *= a.scale; dq[d] = 0.f; }
    const bf16_t* kb = a.k + seq_base(a.km, seq) + head * DH;
    const bf16_t* vb = a.v + seq_base(a.km, seq) + head * DH;
    const int kend = (a.causal && qi + 1 < a.Sk) ? qi + 1 : a.Sk;
    const unsigned drop_key = attn_drop_seqkey(a.drop_key, seq, a.nhead, head);
#pragma unroll 2
    for (int kj = 0; kj < kend; ++kj) {
        float kr[DH], vr[DH];
        load_row<DH>(kb + (long long)kj * a.km.pos_stride, kr);
        load_row<DH>(vb + (long long)kj * a.km.pos_stride, vr);
        const float p = __expf(dot<DH>(q, kr) - lse);
        float dp = dot<DH>(dO, vr);
        if (a.drop_thr) dp = attn_drop_keep(drop_key, a.drop_thr, qi, kj, a.Sk) ? dp * a.drop_scale : 0.f;      // dS = P o (M dP / (1 - p) - D)
        const float ds = p * (dp - D);
#pragma unroll
        for (int d = 0; d < DH; ++d) dq[d] = __builtin_fmaf(ds, kr[d], dq[d]);
    }
#pragma unroll
    for (int d = 0; d < DH; ++d) dq[d] *= a.scale;
    store_row<DH>(a.dq + seq_base(a.qm, seq) + (long long)qi * a.qm.pos_stride + head * DH, dq);
}

// lane = key: dK, dV (reads the D written by the dq kernel)
template <int DH>
__global__ void __launch_bounds__(256) attn_narrow_dkv_kernel(const AttnBwdArgs a) {
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    if (g >= (long long)a.nseq * a.Sk) return;
    const int seq = (int)(g / a.Sk), kj = (int)(g % a.Sk), head = blockIdx.y;
    float kr[DH], vr[DH], dk[DH], dv[DH];
    const long long koff = seq_base(a.km, seq) + (long long)kj * a.km.pos_stride + head * DH;
    load_row<DH>(a.k + koff, kr);
    load_row<DH>(a.v + koff, vr);
#pragma unroll
    for (int d = 0; d < DH; ++d) { kr[d] *= a.scale; dk[d] = 0.f; dv[d] = 0.f; }
    const bf16_t* qb = a.q + seq_base(a.qm, seq) + head * DH;
    const long long ob = seq_base(a.om, seq);
    const int qbeg = a.causal ? kj : 0;
    const unsigned drop_key = attn_drop_seqkey(a.drop_key, seq, a.nhead, head);
#pragma unroll 2
    for (int qi = qbeg; qi < a.Sq; ++qi) {
        float qr[DH], dO[DH];
        load_row<DH>(qb + (long long)qi * a.qm.pos_stride, qr);
        const long long orow = ob + (long long)qi * a.om.pos_stride;
        load_row<DH>(a.dO + orow + head * DH, dO);
        const long long tok = orow / a.C;
        const float lse = a.lse[tok * a.nhead + head], D = a.D[tok * a.nhead + head];
        const float p = __expf(dot<DH>(qr, kr) - lse);
        float dp = dot<DH>(dO, vr), pd = p;
        if (a.drop_thr) {
            const bool keep = attn_drop_keep(drop_key, a.drop_thr, qi, kj, a.Sk);
            dp = keep ? dp * a.drop_scale : 0.f;
            pd = keep ? p * a.drop_scale : 0.f;
        }
        const float ds = p * (dp - D);
#pragma unroll
        for (int d = 0; d < DH; ++d) {
            dv[d] = __builtin_fmaf(pd, dO[d], dv[d]);
            dk[d] = __builtin_fmaf(ds, qr[d], dk[d]);
        }
    }
    const long long o = seq_base(a.dkm, seq) + (long long)kj * a.dkm.pos_stride + head * DH;
    if (a.fuse_self) {                 // du = dQ (already in the buffer) + dK + dV
        float r[DH];
        load_row<DH>(a.dq_in + o, r);
#pragma unroll
        for (int d = 0; d < DH; ++d) r[d] += __builtin_fmaf(dk[d], a.scale, dv[d]);
        store_row<DH>(a.dk + o, r);
    } else {
#pragma unroll
        for (int d = 0; d < DH; ++d) dk[d] *= a.scale;
        store_row<DH>(a.dk + o, dk);
        store_row<DH>(a.dv + o, dv);
    }
}

}  // namespace

int genie_attn_narrow_fwd(const AttnArgs& a, int d_head, hipStream_t s) {
    const long long lanes = (long long)a.nseq * a.Sq;
    GENIE_CHECK_ARG((lanes + 255) / 256 < (1ll << 31) && a.nhead <= 65535, "genie_attention_fwd: grid too large");
    dim3 grid((unsigned)((lanes + 255) / 256), a.nhead);
    if (d_head == 8) attn_narrow_fwd_kernel<8><<<grid, 256, 0, s>>>(a);
    else attn_narrow_fwd_kernel<16><<<grid, 256, 0, s>>>(a);
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}

int genie_attn_narrow_bwd(const AttnBwdArgs& a, int d_head, hipStream_t s) {
    const long long ql = (long long)a.nseq * a.Sq, kl = (long long)a.nseq * a.Sk;
    GENIE_CHECK_ARG((ql + 255) / 256 < (1ll << 31) && (kl + 255) / 256 < (1ll << 31) && a.nhead <= 65535, "genie_attention_bwd: grid too large");
    dim3 gq((unsigned)((ql + 255) / 256), a.nhead), gk((unsigned)((kl + 255) / 256), a.nhead);
    float* D = const_cast<float*>(a.D);
    if (d_head == 8) {
        attn_narrow_dq_kernel<8><<<gq, 256, 0, s>>>(a, D);
        attn_narrow_dkv_kernel<8><<<gk, 256, 0, s>>>(a);
    } else {
        attn_narrow_dq_kernel<16><<<gq, 256, 0, s>>>(a, D);
        attn_narrow_dkv_kernel<16><<<gk, 256, 0, s>>>(a);
    }
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}
