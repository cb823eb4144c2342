// Argument blocks shared by the attention kernel families (attention.hip: MFMA flash kernels for d_head 32 / 64 / 128;
// attention_narrow.hip: fp32 VALU kernels for d_head 8 / 16).
#pragma once
#include "common.h"

struct SeqMap {
    int n_inner;
    long long stride_outer, stride_inner, pos_stride;
};

__device__ __forceinline__ long long seq_base(const SeqMap& m, int seq) {
    return (long long)(seq / m.n_inner) * m.stride_outer + (long long)(seq % m.n_inner) * m.stride_inner;
}

struct AttnArgs {
    const bf16_t* q; const bf16_t* k; const bf16_t* v; const bf16_t* resid; bf16_t* out; bf16_t* oattn;
    float* lse;                 // [token][nhead], token = (element offset of the token row) / C  (may be null)
    int C;                      // channels per token row (= nhead * DH for the q/out tensor)
    SeqMap qm, km, om;          // q/out/resid share qm for addressing of q; om for out & resid
    int nseq, nhead, Sq, Sk;
    float scale;
    int causal;
    int kv_same;                // k == v tile (one LDS image)
    int xcd_swizzle;            // lean kernels: 1-D grid, workgroup -> (sequence, head, tile) through attn_block_of (0: blockIdx.x / .y as is)
    // attention dropout (genie_attention_fwd_dropout; general kernels only): an element is DROPPED when attn_drop_hash(...) < drop_thr
    unsigned drop_thr = 0, drop_key = 0;
    float drop_scale = 1.f;     // 1 / (1 - p)
};

struct AttnBwdArgs {
    const bf16_t* q; const bf16_t* k; const bf16_t* v; const bf16_t* dO;
    const float* lse; const float* D;
    const float* lse2;                        // lse * log2(e) and
    const float* negD;                        // -D, written behind D by the preprocess kernel (D_ws = [D | lse2 | negD], ABI 10)
    const bf16_t* out; const bf16_t* resid;   // narrow kernels: D is computed in the dQ kernel (out = o_attn when resid is null)
    bf16_t* dq; bf16_t* dk; bf16_t* dv;      // dq: q-map addressing; dk/dv: kv-map addressing (dkv kernel)
    const bf16_t* dq_in;                      // dkv kernel, fused self-attention: dk row += dq_in row (then dk holds dQ + dK + dV)
    SeqMap qm, km, om, dkm;
    int nseq, nhead, Sq, Sk, C, Ckv;
    float scale;
    int causal, kv_same, fuse_self;
    int xcd_swizzle;            // as AttnArgs::xcd_swizzle
    unsigned drop_thr = 0, drop_key = 0;      // as AttnArgs (genie_attention_bwd_dropout)
    float drop_scale = 1.f;
};

// Attention dropout (reference attention.py:229 hands `dropout_p` to scaled_dot_product_attention).  Counter-based: the decision for
// (sequence, head, query, key) is a pure function of the call's seed -- two rounds of a multiply-xorshift mixer over the element's index,
// keyed per (sequence, head) -- so the forward kernel, both backward kernels and genie_attention_dropout_mask (which writes the decisions out
// for the parity tests) agree without any stored mask.
__device__ __forceinline__ unsigned attn_drop_mix(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ unsigned attn_drop_seqkey(unsigned call_key, int seq, int nhead, int head) {
    return attn_drop_mix(call_key + (unsigned)(seq * nhead + head) * 0x9E3779B9U);
}
__device__ __forceinline__ bool attn_drop_keep(unsigned seqkey, unsigned thr, int q, int k, int Sk) {
    return attn_drop_mix(((unsigned)q * (unsigned)Sk + (unsigned)k) ^ seqkey) >= thr;
}



// d_head 8 / 16 (attention_narrow.hip); same contracts as the MFMA kernels behind genie_attention_fwd / genie_attention_bwd
int genie_attn_narrow_fwd(const AttnArgs& a, int d_head, hipStream_t s);
int genie_attn_narrow_bwd(const AttnBwdArgs& a, int d_head, hipStream_t s);

// register-lean d_head = 64 kernels (attention_lean.hip).  genie_attn_lean_fwd_ok: the forward call may take them;
// genie_attn_lean_bwd_mask: bit 0 = the dQ kernel, bit 1 = the dK / dV kernel may (D / lse2 already computed by the preprocess kernel)
bool genie_attn_lean_fwd_ok(const AttnArgs& a, int d_head);
int genie_attn_lean_fwd(const AttnArgs& a, hipStream_t s);
int genie_attn_lean_bwd_mask(const AttnBwdArgs& a, int d_head);
int genie_attn_lean_bwd_dq(const AttnBwdArgs& a, hipStream_t s);
int genie_attn_lean_bwd_dkv(const AttnBwdArgs& a, hipStream_t s);
