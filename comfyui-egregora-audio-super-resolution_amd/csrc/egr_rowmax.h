// Per-batch-row operand maxima for the two-term fp16 scheme of the split contractions (csrc/egr_conv.h h2_row_scale): kernels that
// hold a tensor's values in registers anyway leave max |x| per batch row behind, so that the contraction reading the tensor needs
// no pass of its own over it (csrc/egr_flashsr.cpp row_amax_of is that pass, for every other producer).
#pragma once
#include <hip/hip_runtime.h>

namespace egr {

__device__ __forceinline__ float amax4(const float4& v, float m) {
    return fmaxf(fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))), m);
}

// Raises *slot (bits of a non-negative float: they order like unsigned integers) to the wave's maximum of m.  EVERY lane of the
// wave must arrive here together (lanes without work pass 0).  Every wave of a launch hits the same few words, and what a shared
// cache line sustains is about one access per nanosecond: a producer that calls this once per wave of a short-lived grid (200 k
// waves) ran 2.5x slower than without it.  The producers therefore launch (row chunks) x (batch rows) grids of a few thousand
// workgroups that LOOP over their row: a thread keeps a running maximum and the wave commits ONCE, at its end.
__device__ __forceinline__ void row_amax_commit(unsigned* slot, float m) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    const unsigned bits = __float_as_uint(m);
    if ((threadIdx.x & 63) == 0 && bits > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, bits);
}

// The same with ONE commit per workgroup (all threads of the workgroup must arrive; 256 threads): lets a producer whose waves are
// short keep a fine grid (no tail of idle compute units) at a few thousand commits per launch.
__device__ __forceinline__ void row_amax_commit_wg(unsigned* slot, float m) {
    __shared__ float wg_max[4];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) wg_max[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned bits = __float_as_uint(fmaxf(fmaxf(wg_max[0], wg_max[1]), fmaxf(wg_max[2], wg_max[3])));
        if (bits > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, bits);
    }
}

// grid.x of such a launch: enough workgroups of 256 threads for `per_row` items (at least `min_items` per thread: every
// workgroup ends with a commit, and same-line atomics serialise at ~20 ns each), at most ~max_wg workgroups over the B rows
static inline unsigned row_grid_x(long long per_row, int B, int max_wg = 4096, int min_items = 1) {
    long long nx = (per_row + 256LL * min_items - 1) / (256LL * min_items), cap = max_wg / (B < 1 ? 1 : B);
    if (cap < 1) cap = 1;
    if (nx > cap) nx = cap;
    return (unsigned)(nx < 1 ? 1 : nx);
}

}  // namespace egr
