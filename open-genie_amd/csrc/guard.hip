// Guard-page device allocations for the memory-safety harness (tests/guard.py, scripts/guard_run.py).
//
// Every kernel of this library takes raw pointers + pitches; a kernel that walks a pitch, a slice view or a ragged tail one element too far
// reads (or writes) memory that belongs to somebody else -- harmless or fatal depending on where the caching allocator happened to put the
// tensor (round 4's chan_sum_kernel read up to pitch - C channels past the end of a channel-slice view and faulted only in ONE order of the test
// files).  genie_guard_alloc maps exactly the pages an operand needs into a LARGER reserved virtual range, with the operand's last byte on
// the last mapped byte and an unmapped page on either side: the first out-of-bounds access in either direction is a GPU page fault, on every
// run, wherever the allocator would have put the tensor.  HIP virtual-memory API (hipMemAddressReserve / hipMemCreate / hipMemMap).
// Test infrastructure: nothing on the product path calls these.
#include "common.h"
#include "genie_hip.h"

namespace {
struct GuardBlock {
    void* va;
    size_t va_size, map_size, gran;
    hipMemGenericAllocationHandle_t handle;
};
}  // namespace

#define GUARD_CHECK(call)                                                                              \
    do {                                                                                               \
        hipError_t e__ = (call);                                                                       \
        if (e__ != hipSuccess) {                                                                       \
            genie_set_error("genie_guard_alloc: %s failed: %s", #call, hipGetErrorString(e__));       \
            return GENIE_ERR_HIP;                                                                      \
        }                                                                                              \
    } while (0)

extern "C" int genie_guard_alloc(int64_t bytes, void** ptr, void** handle) {
    GENIE_CHECK_ARG(bytes >= 0 && ptr && handle, "genie_guard_alloc: bad arguments");
    int dev = 0;
    GUARD_CHECK(hipGetDevice(&dev));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    GUARD_CHECK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    const size_t want = (size_t)((bytes + 15) & ~15ll);                  // 16-byte aligned start; at most 15 slack bytes behind the operand
    const size_t map = ((want > 0 ? want : 16) + gran - 1) / gran * gran;
    GuardBlock* g = new GuardBlock{nullptr, map + 2 * gran, map, gran, {}};
    GUARD_CHECK(hipMemAddressReserve(&g->va, g->va_size, gran, nullptr, 0));
    GUARD_CHECK(hipMemCreate(&g->handle, map, &prop, 0));
    GUARD_CHECK(hipMemMap((char*)g->va + gran, map, 0, g->handle, 0));
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    GUARD_CHECK(hipMemSetAccess((char*)g->va + gran, map, &acc, 1));
    *ptr = (char*)g->va + gran + map - want;
    *handle = g;
    return GENIE_OK;
}

extern "C" int genie_guard_free(void* handle) {
    GuardBlock* g = (GuardBlock*)handle;
    if (!g) return GENIE_OK;
    (void)hipMemUnmap((char*)g->va + g->gran, g->map_size);
    (void)hipMemRelease(g->handle);
    (void)hipMemAddressFree(g->va, g->va_size);
    delete g;
    return GENIE_OK;
}
