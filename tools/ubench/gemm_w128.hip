// Dev experiment (round 4): the fp16-two-term GEMM of csrc/egr_nn_gemm_s3.hip with ONE wave per SIMD and a 128 x 128 wave tile
// (256 x 256 block tile, 256 accumulator registers per lane -> AGPRs), against the shipped k_conv_s3<128, 256, 1, false, 1>
// (64 x 128 wave tiles, two workgroups per CU) on the dominant GEMM shapes of the FlashSR stage.  Stand-alone (no torch):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I comfyui-egregora-audio-super-resolution_amd/csrc tools/ubench/gemm_w128.hip \
//         -L comfyui-egregora-audio-super-resolution_amd -legregora_amd -Wl,-rpath,$PWD/comfyui-egregora-audio-super-resolution_amd -o /tmp/gemm_w128
//   /tmp/gemm_w128 [M N K reps]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include <type_traits>
#include "egregora_amd.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split2h_pair(float a, float b, float s, uint32_t& p0, uint32_t& p1) {
    const float as = a * s, bs = b * s;
    const f32x2 v = {as, bs};
    const f16x2 hi = __builtin_convertvector(v, f16x2);
    p0 = __builtin_bit_cast(uint32_t, hi);
    const f32x2 r = {as - (float)hi[0], bs - (float)hi[1]};
    p1 = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, f16x2));
}
__device__ __forceinline__ void split2h_x8(const float4& u, const float4& v, float s, uint4& q0, uint4& q1) {
    split2h_pair(u.x, u.y, s, q0.x, q1.x);
    split2h_pair(u.z, u.w, s, q0.y, q1.y);
    split2h_pair(v.x, v.y, s, q0.z, q1.z);
    split2h_pair(v.z, v.w, s, q0.w, q1.w);
}
__device__ __forceinline__ float row_scale(unsigned bits) {
    const int e = min(max((int)((bits >> 23) & 0xffu), 15), 254);
    return __uint_as_float((unsigned)(268 - e) << 23);
}
__device__ __forceinline__ float row_inv(unsigned bits) {
    const int e = min(max((int)((bits >> 23) & 0xffu), 15), 254);
    return __uint_as_float((unsigned)(e - 14) << 23);
}
__device__ __forceinline__ f16x8 as_hf(const uint4& v) { return __builtin_bit_cast(f16x8, v); }

// C [M][N] = A [M][K] (fp32) x W (two fp16 planes of w * w_scale, [K/16][2][N][16]); M % 256 == 0, N % 256 == 0, K % 16 == 0
// Pipeline per 16-k slab s (one wave per SIMD, nobody else hides a stall): global loads run 2-3 slabs ahead into registers, the
// split + LDS stores 2 slabs ahead into a ring of THREE LDS buffers, the MFMA operands of slab s + 1 are read from LDS into a second
// register set while slab s multiplies (its buffer was published by the barrier that ended slab s - 1).
#ifndef SGB
#define SGB 1
#endif
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_gemm_w128(
    const float* __restrict__ A, const uint4* __restrict__ W2, float* __restrict__ C, int M, int N, int K, const unsigned* __restrict__ row_amax,
    int rows_div, float out_scale) {
    __shared__ uint4 As[3][2][256 * 2];
    __shared__ uint4 Bs[3][2][256 * 2];
    __shared__ float os_tab[256];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm0 = (wave >> 1) * 128, wn0 = (wave & 1) * 128;
    const int m0 = blockIdx.x * 256, n0 = blockIdx.y * 256;
    const int ar = tid >> 1, ah = tid & 1;
    const float s0 = row_scale(row_amax[(size_t)((m0 + ar) / rows_div) * EGR_ROW_AMAX_STRIDE]);
    const float s1 = row_scale(row_amax[(size_t)((m0 + ar + 128) / rows_div) * EGR_ROW_AMAX_STRIDE]);
    os_tab[tid] = row_inv(row_amax[(size_t)((m0 + tid) / rows_div) * EGR_ROW_AMAX_STRIDE]);
    const float* ap0 = A + (size_t)(m0 + ar) * K + ah * 8;
    const float* ap1 = ap0 + (size_t)128 * K;
    const int a_slot = ar * 2 + (ah ^ ((ar >> 3) & 1));
    const uint4* bp0; const uint4* bp1; const uint4* bp2; const uint4* bp3;
    int bs0, bs1, bs2, bs3;
#define BSET(I, PTR, SLOT) { const int e = tid + 256 * (I), plane = e >> 9, rem = e & 511, nl = rem >> 1, half = rem & 1; \
        PTR = W2 + ((size_t)plane * N + n0 + nl) * 2 + half; SLOT = plane * 512 + nl * 2 + (half ^ ((nl >> 3) & 1)); }
    BSET(0, bp0, bs0) BSET(1, bp1, bs1) BSET(2, bp2, bs2) BSET(3, bp3, bs3)
#undef BSET
    const size_t b_slab = (size_t)N * 4;         // uint4 per 16-k slab (2 planes x N x 2 halves)
    const int li = lane & 31, lk = lane >> 5;
    const int o_slot = li * 2 + (lk ^ ((li >> 3) & 1));

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    struct SA { float4 a0, a1, a2, a3; };
    struct SB { uint4 b0, b1, b2, b3; };
    const int ns = K / 16;
    auto a_issue = [&](SA& r, int s) {
        const float* p0 = ap0 + s * 16;
        const float* p1 = ap1 + s * 16;
        r.a0 = *(const float4*)p0; r.a1 = *(const float4*)(p0 + 4);
        r.a2 = *(const float4*)p1; r.a3 = *(const float4*)(p1 + 4);
    };
    auto b_issue = [&](SB& r, int s) {
        const size_t o = (size_t)s * b_slab;
        r.b0 = bp0[o]; r.b1 = bp1[o]; r.b2 = bp2[o]; r.b3 = bp3[o];
    };
    auto storeA = [&](const SA& r, int buf) {
        uint4 q0, q1;
        split2h_x8(r.a0, r.a1, s0, q0, q1);
        As[buf][0][a_slot] = q0; As[buf][1][a_slot] = q1;
        split2h_x8(r.a2, r.a3, s1, q0, q1);
        As[buf][0][a_slot + 256] = q0; As[buf][1][a_slot + 256] = q1;
    };
    auto storeB = [&](const SB& r, int buf) {
        Bs[buf][0][bs0] = r.b0; Bs[buf][0][bs1] = r.b1; Bs[buf][0][bs2] = r.b2; Bs[buf][0][bs3] = r.b3;
    };
    // ---- prologue: slabs 0 and 1 in LDS buffers 0 and 1; A operands of slab 0 and B operands of its first column block in registers;
    // the global data of slab 2 in flight
    SA sa, sa2;
    SB sb;
    uint4 a0[4][2], a1[4][2], bq[2][2];
    a_issue(sa, 0); b_issue(sb, 0);
    storeA(sa, 0); storeB(sb, 0);
    if (1 < ns) { a_issue(sa, 1); b_issue(sb, 1); storeA(sa, 1); storeB(sb, 1); }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < 2; ++q) a0[i][q] = As[0][q][(wm0 + i * 32) * 2 + o_slot];
#pragma unroll
    for (int q = 0; q < 2; ++q) bq[0][q] = Bs[0][q][wn0 * 2 + o_slot];
    if (2 < ns) { a_issue(sa, 2); b_issue(sb, 2); }
    if (3 < ns) a_issue(sa2, 3);
    // (A is requested TWO slabs ahead through two register sets -- an HBM / MALL miss has two slab times to land --, the weights, L2
    // hits shared by every row tile, one slab ahead)
    // steady state, slab s (LDS buffers c0 = s % 3, c1 = (s+1) % 3, c2 = (s+2) % 3): A operands of slab s + 1 -> `an`; slab s + 2
    // (registers sa / sb, requested one slab ago) -> buffer c2; request slab s + 3; MFMAs of slab s column block by column block with the
    // B operands one block ahead (the last block fetches the first one of slab s + 1)
    auto step = [&](int s, int c0, int c1, int c2, uint4 (&ac)[4][2], uint4 (&an)[4][2], SA& ra, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        if (FULL || s + 1 < ns) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int q = 0; q < 2; ++q) an[i][q] = As[c1][q][(wm0 + i * 32) * 2 + o_slot];
        }
        if (FULL || s + 2 < ns) { storeB(sb, c2); storeA(ra, c2); }
        if (FULL || s + 3 < ns) b_issue(sb, s + 3);
        if (FULL || s + 4 < ns) a_issue(ra, s + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j + 1 < 4) {
#pragma unroll
                for (int q = 0; q < 2; ++q) bq[(j + 1) & 1][q] = Bs[c0][q][(wn0 + (j + 1) * 32) * 2 + o_slot];
            } else if (FULL || s + 1 < ns) {
#pragma unroll
                for (int q = 0; q < 2; ++q) bq[0][q] = Bs[c1][q][wn0 * 2 + o_slot];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_hf(bq[j & 1][0]), as_hf(ac[i][1]), acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_hf(bq[j & 1][1]), as_hf(ac[i][0]), acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_hf(bq[j & 1][0]), as_hf(ac[i][0]), acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    };
    typedef std::integral_constant<bool, true> FullT;
    typedef std::integral_constant<bool, false> TailT;
    int s = 0;
    for (; s + 10 < ns; s += 6) {                 // ring of three LDS buffers x two A-operand register sets: period 6
        step(s, 0, 1, 2, a0, a1, sa, FullT());
        step(s + 1, 1, 2, 0, a1, a0, sa2, FullT());
        step(s + 2, 2, 0, 1, a0, a1, sa, FullT());
        step(s + 3, 0, 1, 2, a1, a0, sa2, FullT());
        step(s + 4, 1, 2, 0, a0, a1, sa, FullT());
        step(s + 5, 2, 0, 1, a1, a0, sa2, FullT());
    }
    for (; s < ns; s += 2) {
        step(s, s % 3, (s + 1) % 3, (s + 2) % 3, a0, a1, sa, TailT());
        if (s + 1 < ns) step(s + 1, (s + 1) % 3, (s + 2) % 3, s % 3, a1, a0, sa2, TailT());
    }
    // transposed accumulators: register quad g of acc[i][j] = channels n0 + wn0 + 32 j + 8 g + 4 (lane >> 5) .. +3 of pixel lane & 31
    const int px = lane & 31, ch4 = 4 * (lane >> 5);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm0 + i * 32 + px;
        const float os = os_tab[wm0 + i * 32 + px] * out_scale;
        float* row = C + (size_t)m * N + n0 + wn0 + ch4;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *(float4*)(row + j * 32 + 8 * g) = make_float4(acc[i][j][4 * g] * os, acc[i][j][4 * g + 1] * os, acc[i][j][4 * g + 2] * os, acc[i][j][4 * g + 3] * os);
    }
}

// ---- the shipped design as a pure GEMM (k_conv_s3<128, 256, 1, false, 1>'s slab2 loop: 64 x 128 wave tiles, two workgroups per CU), for
// quick experiments on its loop.  V2W: 0 = as shipped (two LDS buffers, A operands read after the barrier); 1 = ring of THREE LDS
// buffers, the A operands and the first B operands of slab s + 1 are read during slab s (their buffer was published by the barrier
// that ended slab s - 1), A requested one slab ahead instead of two to pay for the registers.
#ifndef V2W
#define V2W 0
#endif
__global__ __launch_bounds__(256, 2) void k_gemm_2wg(const float* __restrict__ A, const uint4* __restrict__ W2, float* __restrict__ C, int M, int N, int K,
                                                      const unsigned* __restrict__ row_amax, int rows_div, float out_scale) {
    constexpr int NB = V2W ? 3 : 2;
    __shared__ uint4 As[NB][2][128 * 2];
    __shared__ uint4 Bs[NB][2][256 * 2];
    __shared__ float os_tab[128];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 128;
    const int m0 = blockIdx.x * 128, n0 = blockIdx.y * 256;
    const int ar = tid >> 1, ah = tid & 1;
    const float s0 = row_scale(row_amax[(size_t)((m0 + ar) / rows_div) * EGR_ROW_AMAX_STRIDE]);
    if (tid < 128) os_tab[tid] = row_inv(row_amax[(size_t)((m0 + tid) / rows_div) * EGR_ROW_AMAX_STRIDE]);
    const float* ap0 = A + (size_t)(m0 + ar) * K + ah * 8;
    const int a_slot = ar * 2 + (ah ^ ((ar >> 3) & 1));
    const uint4* bp0; const uint4* bp1; const uint4* bp2; const uint4* bp3;
    int bs0, bs1, bs2, bs3;
#define BSET(I, PTR, SLOT) { const int e = tid + 256 * (I), plane = e >> 9, rem = e & 511, nl = rem >> 1, half = rem & 1; \
        PTR = W2 + ((size_t)plane * N + n0 + nl) * 2 + half; SLOT = plane * 512 + nl * 2 + (half ^ ((nl >> 3) & 1)); }
    BSET(0, bp0, bs0) BSET(1, bp1, bs1) BSET(2, bp2, bs2) BSET(3, bp3, bs3)
#undef BSET
    const unsigned bstep = (unsigned)N * 4;
    const int li = lane & 31, lk = lane >> 5;
    const int o_slot = li * 2 + (lk ^ ((li >> 3) & 1));
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    struct SA { float4 a0, a1; };
    struct SB { uint4 b0, b1, b2, b3; };
    const int ns = K / 16;
    // (running pointers, as the shipped kernel: the slab index `s` of a request is implied by the call order)
    auto a_issue = [&](SA& r, int) { r.a0 = *(const float4*)ap0; r.a1 = *(const float4*)(ap0 + 4); ap0 += 16; };
    auto b_issue = [&](SB& r, int) { r.b0 = *bp0; r.b1 = *bp1; r.b2 = *bp2; r.b3 = *bp3; bp0 += bstep; bp1 += bstep; bp2 += bstep; bp3 += bstep; };
    auto storeA = [&](const SA& r, int buf) { uint4 q0, q1; split2h_x8(r.a0, r.a1, s0, q0, q1); As[buf][0][a_slot] = q0; As[buf][1][a_slot] = q1; };
    auto storeB = [&](const SB& r, int buf) { Bs[buf][0][bs0] = r.b0; Bs[buf][0][bs1] = r.b1; Bs[buf][0][bs2] = r.b2; Bs[buf][0][bs3] = r.b3; };
    typedef std::integral_constant<bool, true> FullT;
    typedef std::integral_constant<bool, false> TailT;
#if V2W == 0
    SA sa0, sa1;
    SB sb;
    auto slab2 = [&](int t, int cur, SA& ra, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        uint4 a[2][2], b[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 2; ++q) a[i][q] = As[cur][q][(wm0 + i * 32) * 2 + o_slot];
#pragma unroll
        for (int q = 0; q < 2; ++q) b[0][q] = Bs[cur][q][wn0 * 2 + o_slot];
        if (FULL || t + 1 < ns) storeB(sb, cur ^ 1);
        if (FULL || t + 2 < ns) b_issue(sb, t + 2);
        __builtin_amdgcn_sched_barrier(0);
        if (FULL || t + 1 < ns) storeA(ra, cur ^ 1);
        if (FULL || t + 3 < ns) a_issue(ra, t + 3);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j + 1 < 4) {
#pragma unroll
                for (int q = 0; q < 2; ++q) b[(j + 1) & 1][q] = Bs[cur][q][(wn0 + (j + 1) * 32) * 2 + o_slot];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_hf(b[j & 1][0]), as_hf(a[i][1]), acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_hf(b[j & 1][1]), as_hf(a[i][0]), acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_hf(b[j & 1][0]), as_hf(a[i][0]), acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    };
    a_issue(sa0, 0); b_issue(sb, 0);
    storeA(sa0, 0); storeB(sb, 0);
    __syncthreads();
    sa1 = sa0;
    if (1 < ns) { b_issue(sb, 1); __builtin_amdgcn_sched_barrier(0); a_issue(sa1, 1); }
    __builtin_amdgcn_sched_barrier(0);
    if (2 < ns) a_issue(sa0, 2);
    int t = 0;
    for (; t + 4 < ns; t += 2) { slab2(t, 0, sa1, FullT()); slab2(t + 1, 1, sa0, FullT()); }
    for (; t < ns; t += 2) { slab2(t, 0, sa1, TailT()); if (t + 1 < ns) slab2(t + 1, 1, sa0, TailT()); }
#else
    SA sa;
    SB sb;
    uint4 a0[2][2], a1[2][2], bq[2][2];
    a_issue(sa, 0); b_issue(sb, 0);
    storeA(sa, 0); storeB(sb, 0);
    if (1 < ns) { a_issue(sa, 1); b_issue(sb, 1); storeA(sa, 1); storeB(sb, 1); }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 2; ++q) a0[i][q] = As[0][q][(wm0 + i * 32) * 2 + o_slot];
#pragma unroll
    for (int q = 0; q < 2; ++q) bq[0][q] = Bs[0][q][wn0 * 2 + o_slot];
    if (2 < ns) { b_issue(sb, 2); a_issue(sa, 2); }
    auto step = [&](int s, int c0, int c1, int c2, uint4 (&ac)[2][2], uint4 (&an)[2][2], auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        if (FULL || s + 2 < ns) storeB(sb, c2);
        if (FULL || s + 3 < ns) b_issue(sb, s + 3);
        __builtin_amdgcn_sched_barrier(0);
        if (FULL || s + 2 < ns) storeA(sa, c2);
        if (FULL || s + 3 < ns) a_issue(sa, s + 3);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j + 1 < 4) {
#pragma unroll
                for (int q = 0; q < 2; ++q) bq[(j + 1) & 1][q] = Bs[c0][q][(wn0 + (j + 1) * 32) * 2 + o_slot];
            } else if (FULL || s + 1 < ns) {     // last column block: the A operands and the first B operands of slab s + 1
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int q = 0; q < 2; ++q) an[i][q] = As[c1][q][(wm0 + i * 32) * 2 + o_slot];
#pragma unroll
                for (int q = 0; q < 2; ++q) bq[0][q] = Bs[c1][q][wn0 * 2 + o_slot];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_hf(bq[j & 1][0]), as_hf(ac[i][1]), acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_hf(bq[j & 1][1]), as_hf(ac[i][0]), acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_hf(bq[j & 1][0]), as_hf(ac[i][0]), acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    };
    int s = 0;
    for (; s + 9 < ns; s += 6) {
        step(s, 0, 1, 2, a0, a1, FullT());
        step(s + 1, 1, 2, 0, a1, a0, FullT());
        step(s + 2, 2, 0, 1, a0, a1, FullT());
        step(s + 3, 0, 1, 2, a1, a0, FullT());
        step(s + 4, 1, 2, 0, a0, a1, FullT());
        step(s + 5, 2, 0, 1, a1, a0, FullT());
    }
    for (; s < ns; s += 2) {
        step(s, s % 3, (s + 1) % 3, (s + 2) % 3, a0, a1, TailT());
        if (s + 1 < ns) step(s + 1, (s + 1) % 3, (s + 2) % 3, s % 3, a1, a0, TailT());
    }
#endif
    const int px = lane & 31, ch4 = 4 * (lane >> 5);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + wm0 + i * 32 + px;
        const float os = os_tab[wm0 + i * 32 + px] * out_scale;
        float* row = C + (size_t)m * N + n0 + wn0 + ch4;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *(float4*)(row + j * 32 + 8 * g) = make_float4(acc[i][j][4 * g] * os, acc[i][j][4 * g + 1] * os, acc[i][j][4 * g + 2] * os, acc[i][j][4 * g + 3] * os);
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char** argv) {
    int M = 26 * 128 * 4, N = 512, K = 1024, reps = 20;
    if (argc >= 4) { M = atoi(argv[1]); N = atoi(argv[2]); K = atoi(argv[3]); }
    if (argc >= 5) reps = atoi(argv[4]);
    const int rows = 26, rows_div = M / rows;
    printf("M %d N %d K %d (rows_div %d)\n", M, N, K, rows_div);
    std::vector<float> hA((size_t)M * K), hW((size_t)K * N);
    srand(1);
    for (auto& v : hA) v = (rand() / (float)RAND_MAX - 0.5f) * 4.f;
    for (int r = 0; r < rows; ++r) { const float lv = powf(10.f, -0.2f * r); for (size_t i = (size_t)r * rows_div * K; i < (size_t)(r + 1) * rows_div * K; ++i) hA[i] *= lv; }
    for (auto& v : hW) v = (rand() / (float)RAND_MAX - 0.5f) * 2.f / sqrtf((float)K);
    // fp32 pack [K/16][N][16]
    std::vector<float> hP((size_t)K * N);
    for (int k = 0; k < K; ++k) for (int n = 0; n < N; ++n) hP[((size_t)(k / 16) * N + n) * 16 + (k % 16)] = hW[(size_t)k * N + n];
    float *dA, *dP, *dC1, *dC2, *dRA; void* dW2;
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dP, hP.size() * 4)); CK(hipMalloc(&dC1, (size_t)M * N * 4)); CK(hipMalloc(&dC2, (size_t)M * N * 4));
    CK(hipMalloc(&dW2, (size_t)K * N * 2 * 2)); CK(hipMalloc(&dRA, (size_t)rows * EGR_ROW_AMAX_STRIDE * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dP, hP.data(), hP.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dRA, 0, (size_t)rows * EGR_ROW_AMAX_STRIDE * 4));
    const float ws = 8192.f;
    if (egr_split2h_pack(dP, dW2, K / 16, N, ws, nullptr)) { printf("pack: %s\n", egr_last_error()); return 1; }
    if (egr_absmax_rows(dA, rows, (int64_t)rows_div * K, 1, 0, dRA, nullptr)) { printf("absmax: %s\n", egr_last_error()); return 1; }
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto ref = [&]() { if (egr_conv_h2(dA, dW2, nullptr, nullptr, nullptr, dC1, M, 1, 1, K, 1, 1, N, 1, 1, 1, 1, 0, 0, 0, 0, 0.f, 1, 1, 0, 0, 1, 1, 1, 0, 0, 0, ws, dRA, rows, nullptr, nullptr)) { printf("conv_h2: %s\n", egr_last_error()); exit(1); } };
    const bool two = getenv("KERN") && atoi(getenv("KERN")) == 2;
    auto neu = [&]() {
        if (two) hipLaunchKernelGGL(k_gemm_2wg, dim3(M / 128, N / 256), dim3(256), 0, 0, dA, (const uint4*)dW2, dC2, M, N, K, (const unsigned*)dRA, rows_div, 1.0f / ws);
        else hipLaunchKernelGGL(k_gemm_w128, dim3(M / 256, N / 256), dim3(256), 0, 0, dA, (const uint4*)dW2, dC2, M, N, K, (const unsigned*)dRA, rows_div, 1.0f / ws);
    };
    ref(); neu(); CK(hipDeviceSynchronize());
    float t1, t2;
    hipEventRecord(e0); for (int i = 0; i < reps; ++i) ref(); hipEventRecord(e1); CK(hipDeviceSynchronize()); hipEventElapsedTime(&t1, e0, e1);
    hipEventRecord(e0); for (int i = 0; i < reps; ++i) neu(); hipEventRecord(e1); CK(hipDeviceSynchronize()); hipEventElapsedTime(&t2, e0, e1);
    const double fl = 2.0 * M * N * K;
    printf("shipped k_conv_s3: %.3f ms  %.1f TF/s-eq   |   experiment: %.3f ms  %.1f TF/s-eq   (x%.2f)\n", t1 / reps, fl / (t1 / reps * 1e-3) / 1e12, t2 / reps,
           fl / (t2 / reps * 1e-3) / 1e12, t1 / t2);
    std::vector<float> c1((size_t)M * N), c2((size_t)M * N);
    CK(hipMemcpy(c1.data(), dC1, c1.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(c2.data(), dC2, c2.size() * 4, hipMemcpyDeviceToHost));
    double md = 0, mx = 0; size_t bad = 0;
    for (size_t i = 0; i < c1.size(); ++i) { md = fmax(md, fabs((double)c1[i] - c2[i])); mx = fmax(mx, fabs((double)c1[i])); if (c1[i] != c2[i]) ++bad; }
    // a few entries against float64
    double me = 0;
    for (int t = 0; t < 64; ++t) {
        const int m = (int)(((long long)t * 7919) % M), n = (t * 131) % N;
        double a = 0; for (int k = 0; k < K; ++k) a += (double)hA[(size_t)m * K + k] * hW[(size_t)k * N + n];
        me = fmax(me, fabs(a - c2[(size_t)m * N + n]) / (fabs(a) + 1e-30));
    }
    printf("max |new - shipped| %.3e of %.3e (%zu entries differ); new vs float64 on 64 entries: max rel %.2e\n", md, mx, bad, me);
    return 0;
}
