// Device helpers shared by the split-operand contraction kernels (egr_nn_gemm_s3.hip: implicit-GEMM / 1-D / batched kernels;
// egr_nn_conv3x3.hip: the input-stationary 3x3 kernels): the exact operand splits (three bf16 terms / two fp16 terms of the
// pre-scaled operand), the partial-product sequences, and the wave layout of a block tile.
#pragma once
#include "egr_conv.h"

namespace egr {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define S3_BM 128
#define S3_BK 16

__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {        // RNE; a -> bits 0..15, b -> bits 16..31
    f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

// (a, b) -> packed bf16 pairs of the three split terms
__device__ __forceinline__ void split3_pair(float a, float b, uint32_t& p0, uint32_t& p1, uint32_t& p2) {
    p0 = pk_bf16(a, b);
    const float ra = a - __uint_as_float(p0 << 16), rb = b - __uint_as_float(p0 & 0xffff0000u);
    p1 = pk_bf16(ra, rb);
    const float sa = ra - __uint_as_float(p1 << 16), sb = rb - __uint_as_float(p1 & 0xffff0000u);
    p2 = pk_bf16(sa, sb);
}

__device__ __forceinline__ void split3_x8(const float4& u, const float4& v, uint4& q0, uint4& q1, uint4& q2) {
    split3_pair(u.x, u.y, q0.x, q1.x, q2.x);
    split3_pair(u.z, u.w, q0.y, q1.y, q2.y);
    split3_pair(v.x, v.y, q0.z, q1.z, q2.z);
    split3_pair(v.z, v.w, q0.w, q1.w, q2.w);
}

// ---- scheme 1: two fp16 terms of the pre-scaled operand (x s = h0 + h1, h0 = f16(x s), h1 = f16(x s - h0); s a power of two that
// brings the tensor's largest magnitude near 2^12, so every element down to 2^-15 of it keeps 22 significand bits and smaller ones
// an absolute error of 2^-37 of the maximum).  f16 x f16 products are exact in fp32, so  a0 b0 + a0 b1 + a1 b0  carries the fp32
// product up to a1 b1 and the two term roundings (each < 2^-22 |a b|) with HALF the matrix instructions of the bf16 scheme and
// one third fewer LDS operand bytes; fewer accumulator roundings per 16 k (3 instead of 6) make the measured error against
// float64 no larger (tests/test_gpu_split_h2.py).  s is per BATCH ROW and comes from the device: row_amax[b] holds the bits of
// max |x| of row b (left there by the tensor's producer, or by k_absmax_rows), and h2_row_scale turns its exponent into the power
// of two that puts the row's maximum in [2^14, 2^15) -- no host round trip, no history, and a quiet row next to a loud one keeps
// its own 22 bits (csrc/egr_conv.h, csrc/egr_flashsr.cpp).
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split2h_pair(float a, float b, float s, uint32_t& p0, uint32_t& p1) {
    const float as = a * s, bs = b * s;
    const f32x2 v = {as, bs};
    const f16x2 hi = __builtin_convertvector(v, f16x2);                // RNE (v_cvt_pk_f16_f32)
    p0 = __builtin_bit_cast(uint32_t, hi);
    const f32x2 r = {as - (float)hi[0], bs - (float)hi[1]};            // exact
    p1 = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, f16x2));
}

__device__ __forceinline__ void split2h_x8(const float4& u, const float4& v, float s, uint4& q0, uint4& q1) {
    split2h_pair(u.x, u.y, s, q0.x, q1.x);
    split2h_pair(u.z, u.w, s, q0.y, q1.y);
    split2h_pair(v.x, v.y, s, q0.z, q1.z);
    split2h_pair(v.z, v.w, s, q0.w, q1.w);
}

// raises *slot (the bits of a non-negative float, which order like unsigned integers) to the wave's maximum
__device__ __forceinline__ void amax_commit(unsigned* slot, float m) {
    if (!slot) return;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    const unsigned bits = __float_as_uint(m);
    if ((threadIdx.x & 63) == 0 && bits > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, bits);
}

// the operand split of scheme SCH into its NP planes q[0..NP)
template <int SCH>
__device__ __forceinline__ void split_x8(const float4& u, const float4& v, float s, uint4 (&q)[3]) {
    if constexpr (SCH == 0) split3_x8(u, v, q[0], q[1], q[2]);
    else split2h_x8(u, v, s, q[0], q[1]);
}

// the partial products of one 32x32x16 block, smallest terms first
template <int SCH>
__device__ __forceinline__ void mma_split(const uint4 (&a)[3], const uint4 (&b)[3], f32x16& acc) {
    if constexpr (SCH == 0) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[2]), __builtin_bit_cast(bf16x8, b[0]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[0]), __builtin_bit_cast(bf16x8, b[2]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[1]), __builtin_bit_cast(bf16x8, b[1]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[1]), __builtin_bit_cast(bf16x8, b[0]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[0]), __builtin_bit_cast(bf16x8, b[1]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[0]), __builtin_bit_cast(bf16x8, b[0]), acc, 0, 0, 0);
    } else {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[1]), __builtin_bit_cast(f16x8, b[0]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[0]), __builtin_bit_cast(f16x8, b[1]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[0]), __builtin_bit_cast(f16x8, b[0]), acc, 0, 0, 0);
    }
}

// <BM, BN> block tile, 4 waves.  256x128: waves 2x2, each 128x64 (TM 4, TN 2; the large-M workhorse: half the split work and
// 0.7x the L2 traffic per MFMA of the 128x128 tile).  128xBN: small-M / thin-Cout layers and split-K.
template <int BM, int BN> struct S3Cfg;
template <> struct S3Cfg<256, 128> { static constexpr int WM = 2, WN = 2, TM = 4, TN = 2; };
template <> struct S3Cfg<128, 256> { static constexpr int WM = 2, WN = 2, TM = 2, TN = 4; };   // Cout >= 256: half the A work per MFMA
template <> struct S3Cfg<128, 128> { static constexpr int WM = 2, WN = 2, TM = 2, TN = 2; };
template <> struct S3Cfg<128, 64> { static constexpr int WM = 2, WN = 2, TM = 2, TN = 1; };
template <> struct S3Cfg<128, 32> { static constexpr int WM = 4, WN = 1, TM = 1, TN = 1; };

__device__ __forceinline__ bf16x8 as_bf(const uint4& v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ f16x8 as_hf(const uint4& v) { return __builtin_bit_cast(f16x8, v); }

}  // namespace egr
