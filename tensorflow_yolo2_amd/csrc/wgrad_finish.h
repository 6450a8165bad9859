// In-kernel finish of a split-K weight gradient (round 5).
//
// Every block of a split-K launch leaves its partial tile in slab[part][taps*Cin*Cout] (plain stores) and then calls
// splitk_finish.  The `parts` partials of a tile form a fixed tree of groups of <= kWgFinG consecutive indices; a counter
// per group counts arrivals.  The block that completes a group sums the group's partials IN INDEX ORDER into the
// group's first slot and goes on as that group's representative on the next level; the block that completes the top
// group writes dW = scale * sum.  Which block that is depends on timing -- what it computes does not: the sum is
// bit-identical from run to run (tests: test_backward_is_bit_reproducible, test_resnet_backward_is_bit_reproducible).
// No block ever waits for another (an arrival that is not the last simply returns), so nothing here needs the blocks
// of a launch to be co-resident: no deadlock beside the dgrad stream or a second process on the GPU.
// Visibility across the 8 XCDs (one L2 each) WITHOUT cache maintenance: an agent-scope fence writes back and invalidates
// the block's whole L2 (buffer_wbl2 / buffer_inv) -- with a thousand blocks per launch that destroyed the L2 locality of
// the dgrad kernels running beside the weight gradients (measured: 8.7 -> 10.9 ms per step).  Instead every slab access
// of a launch that sums in-kernel is an agent-scope RELAXED atomic store / load (global_store / global_load with the
// sc1 bit: written through to, and read from, the level all XCDs share); the block drains its stores
// (s_waitcnt vmcnt(0)) before its counter increment, and the summing block's loads are issued after it has observed the
// count -- release / acquire by construction, no fence instruction.
// Counters clean themselves (the completing block stores 0), so one zero-fill at bind time serves every launch.
#pragma once
#include "common.h"
#include "kernels.h"

namespace y2 {

// sum(first_slot, slot_stride, count, final): v = sum_i slab[first_slot + i * slot_stride][o] for this thread's elements o,
// in order; final ? dW[o] = v * scale : slab[first_slot][o] = v
// s_last: one int of LDS shared by the WHOLE block (the two tap-group arms of wgrad9 are different instantiations of the
// body: a static __shared__ here would be two variables)
Y2_DEV void slab_store(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
Y2_DEV float slab_load(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <typename F>
Y2_DEV void splitk_finish(int* s_last_p, int* cnt, int part, int parts, F&& sum) {
    volatile int* s_last = s_last_p;
    int n = parts, idx = part, stride = 1;
    while (true) {
        const bool top = n <= kWgFinG;
        const int ng = top ? 1 : (n + kWgFinG - 1) / kWgFinG;
        const int g = top ? 0 : idx / kWgFinG;
        const int first = g * kWgFinG;
        const int gsz = top ? n : (n - first < kWgFinG ? n - first : kWgFinG);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this thread's slab stores (sc1: written through) have landed
        __syncthreads();
        if (threadIdx.x == 0) {
            const int old = __hip_atomic_fetch_add(cnt + g, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = old == gsz - 1;
            // nobody else touches this group's counter again in this launch
            if (last) __hip_atomic_store(cnt + g, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *s_last = last;
        }
        __syncthreads();
        const bool last = *s_last != 0;   // block-uniform
        __syncthreads();                 // s_last is written again on the next level
        if (!last) return;
        sum(first * stride, stride, gsz, top);
        if (top) return;
        cnt += ng;
        idx = g;
        n = ng;
        stride *= kWgFinG;
    }
}

// v = ((p[0] + p[1]) + p[2]) + ... over `count` slots, loads issued eight at a time
Y2_DEV float splitk_sum_slots(const float* p, size_t slot_stride, int count) {
    float v = 0.f;
    int i = 0;
    for (; i + 8 <= count; i += 8) {
        float t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = slab_load(p + (size_t)(i + k) * slot_stride);
#pragma unroll
        for (int k = 0; k < 8; ++k) v += t[k];
    }
    for (; i < count; ++i) v += slab_load(p + (size_t)i * slot_stride);
    return v;
}

}  // namespace y2
