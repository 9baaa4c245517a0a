// Dev micro-benchmark: sustained v_mfma_f32_32x32x16_f16 rate in the shape of the two-term fp16 scheme (three products per block, two
// operand planes), (0) registers only, (1) with 8 ds_read_b128 per 12 MFMAs per wave, (2) = 1 + a barrier per slab; constant operands and
// random ones (11 random significand bits per term).  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_f16_peak mfma_f16_peak.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f16x8 as_hf(const uint4& v) { return __builtin_bit_cast(f16x8, v); }
__device__ __forceinline__ unsigned rbf(unsigned h) {   // two random finite fp16 values: sign, exponent 8..15 of 31 (2^-7 .. 1), 10 random significand bits
    h = h * 2654435761u + 12345u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    const unsigned lo = (h & 0x83ffu) | ((8u + ((h >> 10) & 0x7u)) << 10), hi = ((h >> 16) & 0x83ffu) | ((8u + ((h >> 26) & 0x7u)) << 10);
    return lo | (hi << 16);
}
template <int MODE, int RND>
__global__ __launch_bounds__(256) void k(float* out, int iters, long long* cyc) {
    __shared__ uint4 lds[2][2][256 * 2];
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int lane = threadIdx.x & 63, li = lane & 31, lk = lane >> 5, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 2 * 2 * 512; i += 256) ((uint4*)lds)[i] = RND ? make_uint4(rbf(i * 4 + blockIdx.x), rbf(i * 4 + 1), rbf(i * 4 + 2), rbf(i * 4 + 3)) : make_uint4(0x3c003c00u, 0x38003800u, 0x34003400u, 0x30003000u);
    __syncthreads();
    const int o_slot = li * 2 + (lk ^ ((li >> 3) & 1));
    uint4 a[2][2], b[2][2];
    for (int i = 0; i < 2; ++i) for (int q = 0; q < 2; ++q) { a[i][q] = RND ? make_uint4(rbf(threadIdx.x * 64 + i * 8 + q), rbf(threadIdx.x * 64 + i * 8 + q + 3), rbf(threadIdx.x * 97 + i + q), rbf(threadIdx.x * 31 + i * 5 + q)) : make_uint4(0x3c003c00u, 0x38003800u, 0x34003400u, 0x30003000u); b[i][q] = a[i][q]; b[i][q].x ^= 0x00110011u * (q + 1); }
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE >= 1) {
            const int cur = it & 1;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    a[i][q] = lds[cur][q][((wave >> 1) * 64 + i * 32) * 2 + o_slot];
                    b[i][q] = lds[cur ^ 1][q][((wave & 1) * 64 + i * 32) * 2 + o_slot];
                }
        }
#define MMA(QA, QB) _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[i][j] = \
    __builtin_amdgcn_mfma_f32_32x32x16_f16(as_hf(a[i][QA]), as_hf(b[j][QB]), acc[i][j], 0, 0, 0);
        MMA(1, 0) MMA(0, 1) MMA(0, 0)
        if (MODE >= 2) __syncthreads();
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE, int RND> void run(const char* name, int blocks_per_cu) {
    const int blocks = 256 * blocks_per_cu, iters = 20000;
    float* out; long long* cyc;
    (void)hipMalloc(&out, blocks * 256 * 4); (void)hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, RND>), dim3(blocks), dim3(256), 0, 0, out, 1000, cyc);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, RND>), dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double flops = (double)blocks * 4 * iters * 12.0 * (2.0 * 32 * 32 * 16);
    printf("%-34s rnd=%d blocks/CU=%d  %.3f ms  %.0f TF/s f16 (%.0f fp32-equivalent)  block0 ticks=%lld -> %.1f cycles/MFMA-slot at 2.4 GHz: %.1f\n", name, RND,
           blocks_per_cu, ms, flops / ms / 1e9, flops / ms / 1e9 / 3, c, (double)c / (iters * 12.0 * blocks_per_cu), ms * 1e-3 * 2.4e9 / (iters * 12.0 * blocks_per_cu));
    (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
    run<0, 0>("mfma only", 2); run<0, 1>("mfma only", 2);
    run<1, 0>("mfma + 8 ds_read_b128 / 12", 2); run<1, 1>("mfma + 8 ds_read_b128 / 12", 2);
    run<2, 0>("mfma + reads + barrier", 2); run<2, 1>("mfma + reads + barrier", 2);
    run<2, 1>("mfma + reads + barrier", 2); run<2, 1>("mfma + reads + barrier", 2); run<2, 0>("mfma + reads + barrier", 2);
    return 0;
}
