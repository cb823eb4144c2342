// HBM streaming yardsticks on one MI355X: what a copy / read / write / "2 reads + 1 write" pass reaches as a function of the launch shape
// (blocks, 16-byte accesses in flight per thread, block-contiguous vs grid-strided chunks, non-temporal hints).  Stand-alone probe:
//   hipcc -O3 --offload-arch=gfx950 -o scripts/probes/hbm_stream scripts/probes/hbm_stream.hip && scripts/probes/hbm_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <bool NT> __device__ __forceinline__ u32x4 ld(const u32x4* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st(u32x4* p, u32x4 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// MODE 0 copy, 1 read (sum), 2 write, 3 two reads + one write.  CONTIG: a block owns one contiguous range (thread-interleaved inside);
// else grid-strided.  K accesses of 16 B in flight per thread.
template <int MODE, int K, bool CONTIG, bool NT>
__global__ void __launch_bounds__(256) stream_kernel(const u32x4* __restrict__ a, const u32x4* __restrict__ b, u32x4* __restrict__ o, long long n, u32x4* sink) {
    const long long per_blk = (n + gridDim.x - 1) / gridDim.x;
    long long i0, i1, step;
    if (CONTIG) { i0 = blockIdx.x * per_blk; i1 = i0 + per_blk < n ? i0 + per_blk : n; i0 += threadIdx.x; step = 256; }
    else { i0 = (long long)blockIdx.x * 256 + threadIdx.x; i1 = n; step = (long long)gridDim.x * 256; }
    u32x4 acc = {0u, 0u, 0u, 0u};
    for (long long i = i0; i < i1; i += step * K) {
        u32x4 va[K], vb[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const long long j = i + k * step;
            if (MODE != 2) va[k] = j < i1 ? ld<NT>(a + j) : u32x4{0u, 0u, 0u, 0u};
            if (MODE == 3) vb[k] = j < i1 ? ld<NT>(b + j) : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const long long j = i + k * step;
            if (MODE == 1) acc ^= va[k];
            else if (j < i1) st<NT>(o + j, MODE == 0 ? va[k] : (MODE == 2 ? u32x4{1u, 2u, 3u, (unsigned)j} : va[k] ^ vb[k]));
        }
    }
    if (MODE == 1 && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) *sink = acc;
}

template <int MODE, int K, bool CONTIG, bool NT>
static void run(const char* name, int blocks, const u32x4* a, const u32x4* b, u32x4* o, long long n, u32x4* sink) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ms;
    for (int it = 0; it < 12; ++it) {
        CK(hipEventRecord(e0));
        stream_kernel<MODE, K, CONTIG, NT><<<blocks, 256>>>(a, b, o, n, sink);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1));
        if (it >= 2) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    const double t = ms[ms.size() / 2] * 1e-3;
    const double bytes = (double)n * 16 * (MODE == 0 ? 2 : MODE == 3 ? 3 : 1);
    printf("{\"probe\": \"%s\", \"blocks\": %d, \"K\": %d, \"contig\": %d, \"nt\": %d, \"ms\": %.4f, \"TBps\": %.3f}\n", name, blocks, K, (int)CONTIG, (int)NT, t * 1e3, bytes / t / 1e12);
    fflush(stdout);
}

template <int MODE> static void sweep(const char* name, const u32x4* a, const u32x4* b, u32x4* o, long long n, u32x4* sink) {
    for (int blocks : {1024, 2048, 4096, 8192, 16384, 65536}) {
        run<MODE, 4, false, false>(name, blocks, a, b, o, n, sink);
        run<MODE, 8, false, false>(name, blocks, a, b, o, n, sink);
        run<MODE, 4, true, false>(name, blocks, a, b, o, n, sink);
        run<MODE, 8, true, false>(name, blocks, a, b, o, n, sink);
        run<MODE, 8, true, true>(name, blocks, a, b, o, n, sink);
        run<MODE, 8, false, true>(name, blocks, a, b, o, n, sink);
    }
    run<MODE, 1, false, false>(name, (int)((n + 255) / 256), a, b, o, n, sink);      // one access per thread, no loop
    run<MODE, 2, false, false>(name, (int)((n / 2 + 255) / 256), a, b, o, n, sink);
}

int main(int argc, char** argv) {
    const long long bytes = (argc > 1 ? atoll(argv[1]) : 1024) * (1ll << 20);
    const long long n = bytes / 16;
    u32x4 *a, *b, *o, *sink;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&o, bytes)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes)); CK(hipMemset(o, 0, bytes));
    {   // the runtime's own copy as the yardstick the other numbers have been compared with so far
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int it = 0; it < 3; ++it) CK(hipMemcpyAsync(o, a, bytes, hipMemcpyDeviceToDevice, 0));
        CK(hipEventRecord(e0));
        for (int it = 0; it < 5; ++it) CK(hipMemcpyAsync(o, a, bytes, hipMemcpyDeviceToDevice, 0));
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1));
        printf("{\"probe\": \"hipMemcpyAsync d2d\", \"ms\": %.4f, \"TBps\": %.3f}\n", t / 5, 2.0 * bytes / (t / 5 * 1e-3) / 1e12);
    }
    sweep<0>("copy", a, b, o, n, sink);
    sweep<1>("read", a, b, o, n, sink);
    sweep<2>("write", a, b, o, n, sink);
    sweep<3>("2r1w", a, b, o, n, sink);
    return 0;
}
