// Dev micro-benchmark: sustained v_mfma_f32_32x32x2_f32 rate on this chip, (a) registers only, (b) with the conv
// kernel's LDS operand traffic, (c) b + a streaming global read.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_f32_peak.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, const float* g, int iters, long long* cyc) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 256 * 20];
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int lane = threadIdx.x & 63, li = lane & 31, lk = lane >> 5, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 2 * 256 * 20; i += 256) lds[i] = (float)(i % 7) * 0.125f;
    __syncthreads();
    float4 a0 = make_float4(1.f, 2.f, 3.f, 4.f), a1 = a0, b0 = a0, b1 = a0;
    float gsum = 0.f;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE >= 1) {
            const float* A = lds + ((wave & 1) * 64 + li) * 20 + lk * 8 + (it & 1) * 4;
            const float* B = lds + 256 * 20 + ((wave >> 1) * 64 + li) * 20 + lk * 8 + (it & 1) * 4;
            a0 = *(const float4*)A; a1 = *(const float4*)(A + 32 * 20);
            b0 = *(const float4*)B; b1 = *(const float4*)(B + 32 * 20);
        }
        if (MODE >= 2) gsum += g[((size_t)blockIdx.x * 256 + threadIdx.x) * 4 + ((size_t)it * 262144 * 4) % (64u << 20)];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float av0 = ((float*)&a0)[s], av1 = ((float*)&a1)[s], bv0 = ((float*)&b0)[s], bv1 = ((float*)&b1)[s];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0, bv0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0, bv1, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1, bv0, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1, bv1, acc[3], 0, 0, 0);
        }
    }
    const long long t1 = clock64();
    float s = gsum;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE> void run(const char* name, int blocks_per_cu) {
    const int blocks = 256 * blocks_per_cu, iters = 20000;
    float *out, *g; long long* cyc;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&g, (size_t)(64u << 20) * 4 + (1 << 24)); hipMalloc(&cyc, 8);
    hipMemset(g, 0, (size_t)(64u << 20) * 4 + (1 << 24));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, g, 1000, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, g, iters, cyc);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double flops = (double)blocks * 4 /*waves*/ * iters * 16.0 * (2.0 * 32 * 32 * 2);
    printf("%-28s blocks/CU=%d  %.3f ms  %.1f TF/s  block0 cycles=%lld -> %.3f GHz (s_memtime ticks/time)\n", name, blocks_per_cu, ms,
           flops / ms / 1e9, c, c / (ms * 1e6));
    hipFree(out); hipFree(g); hipFree(cyc);
}
int main() {
    run<0>("mfma only", 1); run<0>("mfma only", 3);
    run<1>("mfma + LDS b128 operands", 1); run<1>("mfma + LDS b128 operands", 3);
    run<2>("mfma + LDS + global stream", 3);
    return 0;
}
