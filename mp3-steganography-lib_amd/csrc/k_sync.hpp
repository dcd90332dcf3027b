// Launch-wide status without a fill launch in front of every kernel: ONE 64-bit word owned by the context, zero between
// launches -- low half = workgroups that have finished, high half = OR of their error bits.  Each group speaks once, through
// one atomic read-modify-write of that word; all modifications of one atomic object are totally ordered, so the group whose
// own RMW returns count n-1 has, in that same returned value, the bits of every group before it.  (Rounds 1-3 kept count and
// bits in two words and leaned on the order of two relaxed atomics of one lane: not something the memory model promises.)
#pragma once
#include <hip/hip_runtime.h>

namespace mp3s {

// -> true for the group that arrives last, *all = every group's bits (the word is zero again when this returns)
__device__ __forceinline__ bool arrive_with_bits(int32_t *pair /* 8-byte aligned */, unsigned bits, unsigned n_groups, unsigned *all)
{
    unsigned long long *w = reinterpret_cast<unsigned long long *>(pair);
    unsigned long long old;
    if (bits == 0) old = atomicAdd(w, 1ull);                       // the common case: nothing to report
    else {
        // a group with something to report (damaged input: rare) adds its arrival and ORs its bits in one compare-and-swap
        old = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (;;) {
            const unsigned long long want = (old + 1ull) | ((unsigned long long)bits << 32);
            const unsigned long long seen = atomicCAS(w, old, want);
            if (seen == old) break;
            old = seen;
        }
    }
    if ((unsigned)old != n_groups - 1u) return false;
    *all = (unsigned)(old >> 32) | bits;
    atomicExch(w, 0ull);
    return true;
}

}  // namespace mp3s
