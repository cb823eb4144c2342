// Device helpers shared by the MFMA attention kernel families (attention.hip, attention_lean.hip): LDS address-space casts, the
// transposing LDS read, the 16-B-chunk swizzle of a K / V / Q tile, and the fp32 row staging of the epilogues.
#pragma once
#include <type_traits>
#include "common.h"

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// ds_read_b64_tr_b16 as inline asm: hipcc treats the builtin as possibly aliasing an in-flight LDS-DMA and puts
// s_waitcnt vmcnt(0) in front of it, which would drain the K/V prefetch every tile.  The caller waits (lgkmcnt) before use.
__device__ __forceinline__ bf16x4_t attn_tr16(uint32_t lds_addr) {
    bf16x4_t v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(lds_addr));
    return v;
}
// 16-B chunk `chunk` of LDS row `row` (CPR chunks per row) lives at slot attn_swz<CPR>(row, chunk).  The key is a bijection of the
// low row bits, so the 32-row ds_read_b128 fragment reads stay conflict-free, and its bit pattern also separates the four
// consecutive rows of a transposing ds_read_b64_tr_b16 group (rows r and r + 2 of a 128-B-pitch tile share banks otherwise:
// the first version's (row >> 1) & 7 key cost 31 % of the LDS cycles in conflicts, SQ_LDS_BANK_CONFLICT).
template <int CPR>
__device__ __forceinline__ int attn_swz(int row, int chunk) {
    if (CPR == 16) return chunk ^ (((row & 3) << 2) | ((row >> 2) & 3));
    if (CPR == 8) return chunk ^ ((((row >> 1) & 1) << 2) | ((row >> 2) & 3));
    if (CPR == 4) return chunk ^ ((row >> 2) & 3);
    return chunk;
}

template <int IMM>
__device__ __forceinline__ bf16x4_t attn_tr16i(uint32_t lds_addr) {      // same read with a compile-time byte offset
    bf16x4_t v;
    if constexpr (IMM <= 65535) {                                        // fits the 16-bit offset field
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "i"(IMM));
    } else {
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(lds_addr + (uint32_t)IMM));
    }
    return v;
}
__device__ __forceinline__ uint32_t attn_lds_offset(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}

// Epilogue staging.  In the MFMA C layout a lane holds 4 consecutive channels of ONE token row, so a direct store instruction
// touches 32 different rows with 16 B each; measured on the T = 16 / S = 64 shapes, where the kernel is pure traffic, that
// pattern ran at ~1.8 TB/s against 5.8 TB/s for whole 16-B chunks, 8 lanes per row.  Every kernel therefore hands its 32 x DH
// wave tile back through LDS (fp32 where a residual / partner gradient is still to be added, so nothing is rounded twice).
template <int DH>
__device__ __forceinline__ void rows_put_f32(float* fl, int lr, int h, int dg, const f32x4_t f) {
    *reinterpret_cast<f32x4_t*>(fl + lr * DH + ((dg ^ (lr & (DH / 8 - 1))) << 3) + 4 * h) = f;
}
template <int DH>
__device__ __forceinline__ void rows_get_f32(const float* fl, int row, int c, float (&f)[8]) {
    const float* src = fl + row * DH + ((c ^ (row & (DH / 8 - 1))) << 3);
    const f32x4_t f0 = *reinterpret_cast<const f32x4_t*>(src), f1 = *reinterpret_cast<const f32x4_t*>(src + 4);
    f[0] = f0[0]; f[1] = f0[1]; f[2] = f0[2]; f[3] = f0[3]; f[4] = f1[0]; f[5] = f1[1]; f[6] = f1[2]; f[7] = f1[3];
}

