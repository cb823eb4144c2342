// Shared device/host helpers for the genie HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

typedef uint16_t bf16_t;   // raw bfloat16 bits

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;   // MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;  // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) float f32x4_t;    // 16x16 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;

#define GENIE_OK 0
#define GENIE_ERR_ARG (-1)
#define GENIE_ERR_HIP (-2)

void genie_set_error(const char* fmt, ...);
void genie_note_variant(int v);   // GENIE_VARIANT_* of the kernel just launched
void genie_note_gn_fused(int mask);   // GroupNorm work the conv call just launched does in its epilogue (genie_last_conv_gn_fused)

#define GENIE_CHECK_ARG(cond, ...)                      \
    do {                                                \
        if (!(cond)) {                                  \
            genie_set_error(__VA_ARGS__);               \
            return GENIE_ERR_ARG;                       \
        }                                               \
    } while (0)

#define GENIE_CHECK_LAUNCH()                                                        \
    do {                                                                            \
        hipError_t e__ = hipGetLastError();                                         \
        if (e__ != hipSuccess) {                                                    \
            genie_set_error("%s:%d: HIP launch failed: %s", __FILE__, __LINE__,     \
                            hipGetErrorString(e__));                                \
            return GENIE_ERR_HIP;                                                   \
        }                                                                           \
    } while (0)

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even, NaN preserved (same rule as torch's float -> bfloat16)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

// two fp32 -> packed bf16x2, round-to-nearest-even: ONE v_cvt_pk_bf16_f32 on gfx950 (the bit-twiddling form above costs ~10 VALU)
typedef __attribute__((ext_vector_type(2))) float genie_f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 genie_bf16x2_t;
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    const genie_f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, genie_bf16x2_t));
}

__device__ __forceinline__ void unpack8(const u32x4_t v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = __uint_as_float(v[i] << 16);
        f[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
    }
}

__device__ __forceinline__ u32x4_t pack8(const float* f) {
    u32x4_t v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = pack_bf16x2(f[2 * i], f[2 * i + 1]);
    return v;
}

// Wave-wide sum, result in every lane.  Four DPP adds reduce each 16-lane row in place (quad_perm, row_half_mirror, row_mirror),
// the four row sums are combined through readlane.  The ds_bpermute butterfly this replaces cost six dependent LDS round trips
// per sum (about 800 cycles in the one-wave-per-token LayerNorm kernels).  Call with all 64 lanes active.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f32<0xB1>(v);        // quad_perm [1, 0, 3, 2]
    v += dpp_f32<0x4E>(v);        // quad_perm [2, 3, 0, 1]
    v += dpp_f32<0x141>(v);       // row_half_mirror
    v += dpp_f32<0x140>(v);       // row_mirror
    const int b = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
    return (r0 + r1) + (r2 + r3);
}

// sigmoid through v_exp_f32 + v_rcp_f32 (1 ulp each): `1.f / x` and `a / x` compile to the IEEE division sequence (v_div_scale,
// v_rcp, four FMAs, v_div_fmas, v_div_fixup -- ~10 VALU) without -ffast-math, and these run once per element of every GroupNorm pass
__device__ __forceinline__ float sigmoid_fast_f(float z) { return __builtin_amdgcn_rcpf(1.f + __expf(-z)); }
__device__ __forceinline__ float silu_f(float z) { return z * sigmoid_fast_f(z); }
// d silu(z) / dz = s (1 + z (1 - s)),  s = sigmoid(z)
__device__ __forceinline__ float silu_grad_f(float z) {
    const float s = sigmoid_fast_f(z);
    return s * (1.f + z * (1.f - s));
}

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
