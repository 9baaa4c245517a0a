// Dev micro-benchmark (round 6): energy per flop of the two f16 MFMA shapes gfx950 offers the two-term fp16 scheme --
// v_mfma_f32_32x32x16_f16 (what k_conv_s3 issues) against v_mfma_f32_16x16x32_f16 -- on a 64 x 64 wave tile, three products per block,
// registers only or with the scheme's LDS operand reads, random 11-bit significands.  The binary runs ONE configuration for a given time;
// tools/r06_mfma_energy.sh samples rocm-smi (socket power, shader clock) next to it: J / TFLOP = W / (TFLOP/s).
//   build: hipcc --offload-arch=gfx950 -O3 -o mfma_f16_energy mfma_f16_energy.hip        run: ./mfma_f16_energy <shape 0|1> <lds 0|1> <seconds>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f16x8 as_hf(const uint4& v) { return __builtin_bit_cast(f16x8, v); }
__device__ __forceinline__ unsigned rbf(unsigned h) {
    h = h * 2654435761u + 12345u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    const unsigned lo = (h & 0x83ffu) | ((8u + ((h >> 10) & 0x7u)) << 10), hi = ((h >> 16) & 0x83ffu) | ((8u + ((h >> 26) & 0x7u)) << 10);
    return lo | (hi << 16);
}
__device__ __forceinline__ uint4 rnd4(unsigned s) { return make_uint4(rbf(s), rbf(s + 77u), rbf(s * 3u + 1u), rbf(s * 5u + 2u)); }

// SHAPE 0: per iteration a 64 x 64 x 16 block = 2 x 2 tiles of 32x32x16, x 3 products = 12 MFMAs (393 216 flops per wave)
// SHAPE 1: per iteration a 64 x 64 x 32 block = 4 x 4 tiles of 16x16x32, x 3 products = 48 MFMAs (786 432 flops per wave)
template <int SHAPE, int LDSR>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ uint4 lds[2][2][512];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 2 * 2 * 512; i += 256) ((uint4*)lds)[i] = rnd4(i * 4 + blockIdx.x);
    __syncthreads();
    float s = 0.f;
    if (SHAPE == 0) {
        f32x16 acc[2][2];
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        uint4 a[2][2], b[2][2];
        for (int i = 0; i < 2; ++i) for (int q = 0; q < 2; ++q) { a[i][q] = rnd4(threadIdx.x * 64 + i * 8 + q); b[i][q] = rnd4(threadIdx.x * 131 + i * 8 + q + 3); }
        for (int it = 0; it < iters; ++it) {
            if (LDSR) {
                const int cur = it & 1, li = lane & 31, lk = lane >> 5, slot = li * 2 + (lk ^ ((li >> 3) & 1));
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int q = 0; q < 2; ++q) { a[i][q] = lds[cur][q][((wave >> 1) * 64 + i * 32) * 2 + slot]; b[i][q] = lds[cur ^ 1][q][((wave & 1) * 64 + i * 32) * 2 + slot]; }
            }
#define MMA(QA, QB) _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[i][j] = \
    __builtin_amdgcn_mfma_f32_32x32x16_f16(as_hf(a[i][QA]), as_hf(b[j][QB]), acc[i][j], 0, 0, 0);
            MMA(1, 0) MMA(0, 1) MMA(0, 0)
#undef MMA
        }
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    } else {
        f32x4 acc[4][4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
        uint4 a[4][2], b[4][2];
        for (int i = 0; i < 4; ++i) for (int q = 0; q < 2; ++q) { a[i][q] = rnd4(threadIdx.x * 64 + i * 8 + q); b[i][q] = rnd4(threadIdx.x * 131 + i * 8 + q + 3); }
        for (int it = 0; it < iters; ++it) {
            if (LDSR) {        // the same bytes per flop from LDS as SHAPE 0 would need for K = 32: 16 ds_read_b128 per 48 MFMAs
                const int cur = it & 1, li = lane & 15, lk = lane >> 4;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int q = 0; q < 2; ++q) { a[i][q] = lds[cur][q][((wave >> 1) * 64 + i * 16 + li) * 4 % 512 + lk]; b[i][q] = lds[cur ^ 1][q][((wave & 1) * 64 + i * 16 + li) * 4 % 512 + lk]; }
            }
#define MMA(QA, QB) _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[i][j] = \
    __builtin_amdgcn_mfma_f32_16x16x32_f16(as_hf(a[i][QA]), as_hf(b[j][QB]), acc[i][j], 0, 0, 0);
            MMA(1, 0) MMA(0, 1) MMA(0, 0)
#undef MMA
        }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main(int argc, char** argv) {
    const int shape = argc > 1 ? atoi(argv[1]) : 0, ldsr = argc > 2 ? atoi(argv[2]) : 0;
    const double secs = argc > 3 ? atof(argv[3]) : 3.0;
    const int blocks = 512, iters = shape ? 10000 : 20000;
    float* out; (void)hipMalloc(&out, blocks * 256 * 4);
    auto launch = [&]() {
        if (shape == 0) { if (ldsr) hipLaunchKernelGGL((k<0, 1>), dim3(blocks), dim3(256), 0, 0, out, iters); else hipLaunchKernelGGL((k<0, 0>), dim3(blocks), dim3(256), 0, 0, out, iters); }
        else { if (ldsr) hipLaunchKernelGGL((k<1, 1>), dim3(blocks), dim3(256), 0, 0, out, iters); else hipLaunchKernelGGL((k<1, 0>), dim3(blocks), dim3(256), 0, 0, out, iters); }
    };
    launch(); (void)hipDeviceSynchronize();
    const double flops_per_launch = (double)blocks * 4 * iters * (shape ? 48.0 * 2 * 16 * 16 * 32 : 12.0 * 2 * 32 * 32 * 16);
    const auto t0 = std::chrono::steady_clock::now();
    int n = 0;
    double el = 0.0;
    do {
        for (int i = 0; i < 8; ++i) launch();
        (void)hipDeviceSynchronize();
        n += 8;
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    } while (el < secs);
    printf("shape %s lds_reads %d: %d launches in %.2f s -> %.0f TFLOP/s of executed f16\n", shape ? "16x16x32" : "32x32x16", ldsr, n, el, flops_per_launch * n / el / 1e12);
    return 0;
}
