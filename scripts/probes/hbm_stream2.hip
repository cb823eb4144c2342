// Second HBM streaming probe: WHY persistent blocks with one contiguous range each stop at ~5.3 TB/s when tiny in-order blocks reach 6.2-7.
//   A  tiny blocks, no loop: a block owns K x 4 KB contiguous (K accesses per thread)
//   B  persistent, block-contiguous range, start ROTATED inside the range by (block * rot) chunks of 4 KB (wraps) -- de-synchronises
//      the low address bits of the blocks (channel camping test)
//   C  persistent, chunk-interleaved: block b takes chunks b, b + G, ... of CH KB, K accesses in flight inside the chunk
// each for read-only (nt / plain) and for 2 reads + 1 write (nt loads, plain stores).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
template <bool NT> __device__ __forceinline__ u32x4 ld(const u32x4* p) { return NT ? __builtin_nontemporal_load(p) : *p; }

template <int MODE, int K, bool NT>     // MODE 1 read, 3 2r1w;  unit = 256 threads x 16 B = 4 KB; processes units [u0, u0 + K) of one contiguous run
__device__ __forceinline__ void do_units(const u32x4* a, const u32x4* b, u32x4* o, long long u0, long long nunits, u32x4& acc) {
    u32x4 va[K], vb[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const long long j = (u0 + k) * 256 + threadIdx.x;
        va[k] = u0 + k < nunits ? ld<NT>(a + j) : u32x4{0u, 0u, 0u, 0u};
        if (MODE == 3) vb[k] = u0 + k < nunits ? ld<NT>(b + j) : u32x4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const long long j = (u0 + k) * 256 + threadIdx.x;
        if (MODE == 1) acc ^= va[k];
        else if (u0 + k < nunits) o[j] = va[k] ^ vb[k];
    }
}

template <int MODE, int K, bool NT>
__global__ void __launch_bounds__(256) tiny_kernel(const u32x4* a, const u32x4* b, u32x4* o, long long nunits, u32x4* sink) {
    u32x4 acc = {0u, 0u, 0u, 0u};
    do_units<MODE, K, NT>(a, b, o, (long long)blockIdx.x * K, nunits, acc);
    if (MODE == 1 && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) *sink = acc;
}
template <int MODE, int K, bool NT>
__global__ void __launch_bounds__(256) rot_kernel(const u32x4* a, const u32x4* b, u32x4* o, long long nunits, int rot, u32x4* sink) {
    const long long per = (nunits + gridDim.x - 1) / gridDim.x;      // units per block
    const long long base = blockIdx.x * per;
    const long long mine = base + per <= nunits ? per : (nunits > base ? nunits - base : 0);
    u32x4 acc = {0u, 0u, 0u, 0u};
    const long long groups = (mine + K - 1) / K;
    long long g = groups ? ((long long)blockIdx.x * rot) % groups : 0;
    for (long long it = 0; it < groups; ++it) {
        do_units<MODE, K, NT>(a, b, o, base + g * K, base + mine, acc);
        if (++g == groups) g = 0;
    }
    if (MODE == 1 && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) *sink = acc;
}
template <int MODE, int K, bool NT>
__global__ void __launch_bounds__(256) chunk_kernel(const u32x4* a, const u32x4* b, u32x4* o, long long nunits, int chunk_units, u32x4* sink) {
    const long long nchunks = (nunits + chunk_units - 1) / chunk_units;
    u32x4 acc = {0u, 0u, 0u, 0u};
    for (long long c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const long long u0 = c * chunk_units, u1 = u0 + chunk_units < nunits ? u0 + chunk_units : nunits;
        for (long long u = u0; u < u1; u += K) do_units<MODE, K, NT>(a, b, o, u, u1, acc);
    }
    if (MODE == 1 && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) *sink = acc;
}

template <typename F> static void timeit(const char* what, double bytes, F launch) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ms;
    for (int it = 0; it < 12; ++it) {
        CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1)); if (it >= 2) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    printf("%-64s %.4f ms  %.3f TB/s\n", what, ms[ms.size() / 2], bytes / (ms[ms.size() / 2] * 1e-3) / 1e12); fflush(stdout);
}

template <int MODE, bool NT> static void family(const u32x4* a, const u32x4* b, u32x4* o, long long nunits, u32x4* sink) {
    const double bytes = (double)nunits * 4096 * (MODE == 3 ? 3 : 1);
    char nm[128];
    const char* m = MODE == 1 ? "read" : "2r1w";
#define TINY(K) { snprintf(nm, sizeof nm, "%s nt=%d A tiny blocks of %d KB", m, (int)NT, 4 * K); \
        timeit(nm, bytes, [&] { tiny_kernel<MODE, K, NT><<<(unsigned)((nunits + K - 1) / K), 256>>>(a, b, o, nunits, sink); }); }
    TINY(1) TINY(2) TINY(4) TINY(8)
    for (int blocks : {2048, 8192})
        for (int rot : {0, 1, 3, 7}) {
            snprintf(nm, sizeof nm, "%s nt=%d B %d blocks contiguous, K=8, start rotated by %d", m, (int)NT, blocks, rot);
            timeit(nm, bytes, [&] { rot_kernel<MODE, 8, NT><<<blocks, 256>>>(a, b, o, nunits, rot, sink); });
        }
    for (int blocks : {2048, 4096})
        for (int ch : {8, 16, 64}) {       // chunk = ch units of 4 KB
            snprintf(nm, sizeof nm, "%s nt=%d C %d blocks, chunks of %d KB interleaved, K=8", m, (int)NT, blocks, 4 * ch);
            timeit(nm, bytes, [&] { chunk_kernel<MODE, 8, NT><<<blocks, 256>>>(a, b, o, nunits, ch, sink); });
            snprintf(nm, sizeof nm, "%s nt=%d C %d blocks, chunks of %d KB interleaved, K=4", m, (int)NT, blocks, 4 * ch);
            timeit(nm, bytes, [&] { chunk_kernel<MODE, 4, NT><<<blocks, 256>>>(a, b, o, nunits, ch, sink); });
        }
}

int main(int argc, char** argv) {
    const long long bytes = (argc > 1 ? atoll(argv[1]) : 1024) * (1ll << 20);
    const long long nunits = bytes / 4096;
    u32x4 *a, *b, *o, *sink;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&o, bytes)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes)); CK(hipMemset(o, 0, bytes));
    family<1, false>(a, b, o, nunits, sink);
    family<1, true>(a, b, o, nunits, sink);
    family<3, false>(a, b, o, nunits, sink);
    family<3, true>(a, b, o, nunits, sink);
    return 0;
}
