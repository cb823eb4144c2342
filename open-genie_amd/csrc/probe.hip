// Bring-up probes for gfx950 primitives whose lane semantics the kernels rely on (tests only).
#include "common.h"
#include "genie_hip.h"

// Every lane supplies its own LDS byte address; returns the 4 x u16 each lane receives from
// ds_read_b64_tr_b16.  LDS image: 2048 u16 copied from global.
__global__ void __launch_bounds__(64) probe_tr16_kernel(const uint16_t* __restrict__ img, const int* __restrict__ addr,
                                                         uint16_t* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = img[i];
    __syncthreads();
    const int a = addr[threadIdx.x];
    bf16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) bf16x4_t*)(reinterpret_cast<char*>(lds) + a));
#pragma unroll
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (uint16_t)v[j];
}

extern "C" int genie_probe_ds_read_tr16(const void* img, const int32_t* addr, void* out, void* stream) {
    GENIE_CHECK_ARG(img && addr && out, "genie_probe_ds_read_tr16: null pointer");
    probe_tr16_kernel<<<1, 64, 0, (hipStream_t)stream>>>((const uint16_t*)img, addr, (uint16_t*)out);
    GENIE_CHECK_LAUNCH();
    return GENIE_OK;
}
