// dev microbenchmark: cost of a device-wide barrier inside a persistent kernel on MI355X, with the agent-scope release / acquire
// fences a cross-XCD hand-off of global data needs (per-XCD L2s are not coherent with each other: MI355X_MICROARCH.md).
// Each "phase" every workgroup writes a 64 KB slice of a buffer and, after the barrier, reads the slice another workgroup wrote
// (checks the value), so the number includes the L2 write-back / invalidate work.  Bounded spins: a stuck barrier ends the kernel.
//   hipcc --offload-arch=gfx950 -O3 -o grid_barrier grid_barrier.hip && ./grid_barrier [wgs_per_cu] [threads] [lds_kb]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target, unsigned* fail) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __threadfence();                                           // release: this workgroup's global writes
        atomicAdd(counter, 1u);
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 22)) { ok = false; atomicExch(fail, 1u); break; }
        }
        __threadfence();                                           // acquire
    }
    __syncthreads();
    return ok;
}

__global__ __launch_bounds__(512) void k_persist(float* buf, int phases, int slice, unsigned* counter, unsigned* fail, unsigned* bad,
                                                  int do_io) {
    extern __shared__ char smem[];
    const unsigned nwg = gridDim.x;
    for (int ph = 0; ph < phases; ++ph) {
        if (do_io) {
            float* mine = buf + (size_t)blockIdx.x * slice;
            for (int i = threadIdx.x; i < slice; i += blockDim.x) mine[i] = (float)(ph * 1000 + (int)blockIdx.x);
        }
        if (!grid_barrier(counter, (unsigned)(ph + 1) * nwg, fail)) return;
        if (*(volatile unsigned*)fail) return;
        if (do_io) {
            const unsigned other = (blockIdx.x * 37u + 11u) % nwg;  // a slice written by a workgroup on (mostly) another XCD
            const float* theirs = buf + (size_t)other * slice;
            float acc = 0.f;
            for (int i = threadIdx.x; i < slice; i += blockDim.x) acc += (theirs[i] != (float)(ph * 1000 + (int)other)) ? 1.f : 0.f;
            if (acc != 0.f) atomicAdd(bad, 1u);
            if (!grid_barrier(counter + 1, (unsigned)(ph + 1) * nwg, fail)) return;   // readers done before the next overwrite
        }
    }
}

int main(int argc, char** argv) {
    const int per_cu = argc > 1 ? atoi(argv[1]) : 2, threads = argc > 2 ? atoi(argv[2]) : 512, lds_kb = argc > 3 ? atoi(argv[3]) : 72;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    hipFuncSetAttribute((const void*)k_persist, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024);
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_persist, threads, lds_kb * 1024);
    printf("CUs %d, occupancy API %d blocks/CU at %d threads + %d KB LDS\n", cus, occ, threads, lds_kb);
    if (occ < per_cu) { printf("not co-resident at %d per CU\n", per_cu); return 0; }
    const int nwg = cus * per_cu, slice = 16384;
    float* buf; unsigned* ctr;
    hipMalloc(&buf, (size_t)nwg * slice * 4);
    hipMalloc(&ctr, 64);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int do_io = 0; do_io < 2; ++do_io)
        for (int phases : {100, 1000}) {
            hipMemset(ctr, 0, 64);
            hipEventRecord(a);
            hipLaunchKernelGGL(k_persist, dim3(nwg), dim3(threads), lds_kb * 1024, 0, buf, phases, slice, ctr, ctr + 4, ctr + 5, do_io);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms = 0;
            hipEventElapsedTime(&ms, a, b);
            unsigned h[16];
            hipMemcpy(h, ctr, 64, hipMemcpyDeviceToHost);
            printf("wgs %d io %d phases %d: %.3f ms -> %.2f us per phase (%d barrier%s + %s), fail %u bad %u\n", nwg, do_io, phases, ms,
                   1e3 * ms / phases, do_io ? 2 : 1, do_io ? "s" : "", do_io ? "64 KB write + 64 KB cross-workgroup read per WG" : "nothing else", h[4], h[5]);
        }
    return 0;
}
