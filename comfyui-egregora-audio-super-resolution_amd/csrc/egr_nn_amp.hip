// One anti-aliased-multi-periodicity unit of the vocoder's thin stages as ONE kernel (fp16 operand scheme):
//     y = conv2( snake2( conv1( snake1(x) ) ) ) + x        x, y [B][L][C] channels-last fp32, C = 16 (the 245 760-sample stage)
// -- what flashsr_arch / oracle.flashsr_torch.vocoder run per (kernel size, dilation) of an AMP block, and what egr_flashsr.cpp used to
// enqueue as snake -> k_conv1d_s3 -> snake -> k_conv1d_s3 (+ residual).  At 16 channels those four launches are pure streaming: each
// reads and writes the whole [26][245 760][16] tensor (409 MB) for a few hundred flops per element -- 9 T bytes per unit, 3.3 TB/s
// measured -- so the unit is fused around an L-tile that never leaves the CU:
//   P0  the tile of x with its halo (5 + h1 + 5 + h2 samples per side) -> LDS, clamped at the sequence ends (= the snake's replicate
//       padding), the tile's max |x| on the way;
//   P1  snake1 (2x up FIR, x + sin^2(a x) / b, 2x down FIR: the register-blocked form of k_snake_aa_reg, one thread per (16 rows, channel))
//       -> two fp16 terms of a1 * s1 in LDS, s1 a power of two from a BOUND of the tile (FIR gains x (max |x| + 1 / b)); rows outside
//       the sequence are zero (the convolution's padding);
//   P2  conv1 (k taps, dilation d) on v_mfma_f32_16x16x32_f16 -- N = 16 channels per accumulator, K = 32 = two taps of 16 channels or one
//       of 32, three products per block (a1 b0 + a0 b1 + a0 b0); taps outermost, a wave's row blocks in registers, so a tap's weight
//       fragments are fetched once per wave (L2) -> c1 (fp32) in LDS over the x tile, its maximum on the way;
//   P3  snake2 -> two fp16 terms of a2 * s2 (same bound rule, from max |c1|);
//   P4  conv2 (k taps, dilation 1) + bias + x -> y, coalesced 64-byte row segments.
// Traffic: x read once (+ halo, + once more for the residual: an L2 hit), y written once -- 2 T instead of 9 T.
// The scales are per TILE (a function of that tile of that row alone), so a row's result still depends on that row only.
#include <stdlib.h>

#include "egr_conv.h"
#include "egr_s3_split.h"

namespace egr {

struct AmpP {
    const float* x; float* y;
    const float* alpha1; const float* beta1; const float* alpha2; const float* beta2; const float* filt;   // snake parameters [C], 12-tap FIR
    const uint4* w1; const uint4* w2;      // two fp16 terms of the packed weights x w_scale: [tap][2][C][16] (egr_split2h_pack)
    const float* bias1; const float* bias2;
    float inv_ws1, inv_ws2;                // 1 / w_scale of the two packs
    int B, L, k, d;
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int AMP_SH = 5;                  // rows of input the anti-aliased snake needs on each side of an output row
constexpr int AMP_MAXH1 = 25, AMP_MAXH2 = 5;

template <int C, int TL> struct AmpGeom {
    static constexpr int R0 = TL + 2 * (AMP_SH + AMP_MAXH1 + AMP_SH + AMP_MAXH2);           // x rows (maximum)
    // rows the x / c1 area holds: the clamp-free snake reads rows 16 run .. 16 run + 25 of its source per 16-row run; the source has n + 10
    // rows for n output rows and the last run starts at 16 (ceil(n / 16) - 1) <= n - 1, so it reaches at most 15 rows past the source's last
    // row (its outputs beyond n are discarded).  Those rows exist (16 spare ones) and are zeroed by P0 -- never another phase's planes, never
    // stale LDS.
    static constexpr int XR = R0 + 16;
    static constexpr int RA = ((TL + 2 * (AMP_SH + AMP_MAXH2) + 15) / 16) * 16 + 2 * AMP_MAXH1 + 2;   // operand rows an MFMA row block may touch
    static constexpr int NCH = C / 8;
    static constexpr int LDS = XR * C * 4 + 2 * RA * NCH * 16;
};

// the anti-aliased snake of k_snake_aa_reg on 16 consecutive rows l0 .. l0 + 15 of channel c, source rows in LDS: S[(clamp(l) - row0) * C + c]
// (f2 = 2 f: the up-sampler's gain folded into its taps -- exact, a power of two commutes with every rounding; a_rev = a / 2 pi: v_sin_f32 takes
// revolutions).  EDGE = false: a tile whose whole halo lies inside the sequence -- no clamping anywhere, the 26 window loads are one base
// address + immediates
template <int C, bool EDGE>
__device__ __forceinline__ void amp_snake_run(const float* __restrict__ S, int row0, int nrows, int l0, int c, int L, const float (&f)[12], const float (&f2)[12],
                                              float a_rev, float ib, float (&out)[16]) {
    constexpr int J = 16;
    float xw[J + 10];
    if constexpr (EDGE) {
#pragma unroll
        for (int i = 0; i < J + 10; ++i) {
            int src = l0 - 5 + i;
            src = src < 0 ? 0 : (src > L - 1 ? L - 1 : src);
            int r = src - row0;
            r = r < 0 ? 0 : (r > nrows - 1 ? nrows - 1 : r);   // (rows the tile does not hold belong to outputs that are discarded)
            xw[i] = S[r * C + c];
        }
    } else {
        // (the last run of a phase reaches up to 10 rows past the rows its source holds: AmpGeom::XR keeps those rows inside the x / c1 area,
        // zeroed by P0; the outputs that read them are discarded)
        const float* base = S + (l0 - 5 - row0) * C + c;
#pragma unroll
        for (int i = 0; i < J + 10; ++i) xw[i] = base[i * C];
    }
    float sv[2 * J + 10];
#pragma unroll
    for (int q = 0; q < 2 * J + 10; ++q) {
        float u = 0.f;
#pragma unroll
        for (int j = 0; j < 6; ++j) u += xw[5 + (q >> 1) - j] * f2[(q & 1) + 2 * j];
        const float sn = __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(u * a_rev));      // v_sin_f32 is defined on |revolutions| <= 256: reduce first
        sv[q] = fmaf(ib * sn, sn, u);
    }
    const int i0 = 2 * l0 - 5, L2 = 2 * L;
    if (EDGE && (i0 < 0 || i0 + 2 * J + 9 > L2 - 1)) {          // runs at the sequence ends: up-rate indices clamp to [0, 2L - 1]
        float s_first = 0.f, s_last = 0.f;
#pragma unroll
        for (int q = 0; q < 2 * J + 10; ++q) {
            if (i0 + q == 0) s_first = sv[q];
            if (i0 + q == L2 - 1) s_last = sv[q];
        }
#pragma unroll
        for (int q = 0; q < 2 * J + 10; ++q) {
            if (i0 + q < 0) sv[q] = s_first;
            if (i0 + q > L2 - 1) sv[q] = s_last;
        }
    }
#pragma unroll
    for (int jj = 0; jj < J; ++jj) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 12; ++k) acc += f[k] * sv[2 * jj + k];
        out[jj] = acc;
    }
}

// workgroup maximum of a non-negative value (256 threads; `slot` is LDS scratch of >= 4 floats); every thread receives it
__device__ __forceinline__ float amp_wg_max(float m, float* slot) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) slot[threadIdx.x >> 6] = m;
    __syncthreads();
    const float r = fmaxf(fmaxf(slot[0], slot[1]), fmaxf(slot[2], slot[3]));
    __syncthreads();
    return r;
}

#ifndef AMP_LB16
#define AMP_LB16 2
#endif
#ifndef AMP_TL16
#define AMP_TL16 240
#endif
template <int C, int TL>
__global__ __launch_bounds__(256, C == 16 ? AMP_LB16 : 2) void k_amp_unit(AmpP p) {
    static_assert(C == 16, "K = 32 of v_mfma_f32_16x16x32_f16 = two taps of 16 channels (the 32-channel instantiation of round 5 -- half the "
                           "positions per tile -- lost to the four launches it replaced and was removed in round 6)");
    typedef AmpGeom<C, TL> G;
    constexpr int NCH = G::NCH;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* XS = (float*)smem;                                    // x tile, later c1
    uint4* AP = (uint4*)(smem + (size_t)G::XR * C * 4);          // operand planes [2][RA][NCH]
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, t0 = blockIdx.x * TL;
    const int k = p.k, d = p.d, L = p.L;
    const int h2 = (k - 1) / 2, h1 = d * (k - 1) / 2;
    const int a2_row0 = t0 - h2, R3 = TL + 2 * h2;               // rows of a2 (conv2's operand)
    const int c1_row0 = a2_row0 - AMP_SH, R2 = R3 + 2 * AMP_SH;  // rows of c1
    const int a1_row0 = c1_row0 - h1, R1 = R2 + 2 * h1;          // rows of a1 (conv1's operand)
    const int x_row0 = a1_row0 - AMP_SH, R0 = R1 + 2 * AMP_SH;   // rows of x
    const float* xb = p.x + (size_t)b * L * C;
    const bool interior = x_row0 >= 0 && x_row0 + R0 <= L;       // (uniform over the workgroup) every row any phase touches lies inside the sequence
    float f[12], f2[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) { f[i] = p.filt[i]; f2[i] = 2.0f * f[i]; }
    // gains of the two FIRs (L1 norms): |snake_aa(v)| <= g_dn (g_up max |v| + 1 / b)
    float g_dn = 0.f, g_e = 0.f, g_o = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) { g_dn += fabsf(f[i]); if (i & 1) g_o += fabsf(f[i]); else g_e += fabsf(f[i]); }
    const float g_up = 2.0f * fmaxf(g_e, g_o);

    // ---- P0: x tile -> LDS ----
    float vmax = 0.f;
    for (int e = tid; e < R0 * (C / 4); e += 256) {
        const int row = e / (C / 4), q = e - row * (C / 4);
        int l = x_row0 + row;
        l = l < 0 ? 0 : (l > L - 1 ? L - 1 : l);
        const float4 v = *(const float4*)(xb + (size_t)l * C + 4 * q);
        *(float4*)(XS + row * C + 4 * q) = v;
        vmax = fmaxf(vmax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
    // rows behind the tile: read by the last 16-row run of the snake phases (discarded outputs); zero, never another phase's planes or stale LDS
    for (int e = R0 * (C / 4) + tid; e < G::XR * (C / 4); e += 256) *(float4*)(XS + e * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    const float xmax = amp_wg_max(vmax, red);                    // (the barriers inside also publish the tile)

    // operand scale from a bound of the snake's output, and the snake itself on `nrows_out` rows starting at global row `out_row0`
    auto snake_to_planes = [&](const float* S, int src_row0, int src_rows, int out_row0, int nrows_out, const float* alpha, const float* beta, float smax,
                               float& inv_scale) {
        float ibmax = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) ibmax = fmaxf(ibmax, 1.0f / (__expf(beta[c]) + 1e-9f));
        const float bound = g_dn * (g_up * smax + ibmax);
        const unsigned bb = __float_as_uint(fmaxf(bound, 1e-30f));
        const float s = h2_row_scale(bb);
        inv_scale = h2_row_inv(bb);
        const int nruns = (nrows_out + 15) / 16;
        for (int it = tid; it < nruns * C; it += 256) {
            const int c = it % C, run = it / C;
            const int l0 = out_row0 + run * 16;
            const float a_rev = __expf(alpha[c]) * 0.15915494309189535f, ib = 1.0f / (__expf(beta[c]) + 1e-9f);
            float o[16];
            if (interior) amp_snake_run<C, false>(S, src_row0, src_rows, l0, c, L, f, f2, a_rev, ib, o);
            else amp_snake_run<C, true>(S, src_row0, src_rows, l0, c, L, f, f2, a_rev, ib, o);
            _Float16* P0 = (_Float16*)AP;
            _Float16* P1 = (_Float16*)(AP + (size_t)G::RA * NCH);
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) {
                const int l = l0 + jj, r = run * 16 + jj;
                float v = (interior || (unsigned)l < (unsigned)L) ? o[jj] * s : 0.f;      // outside the sequence: the convolution's zero padding
                const _Float16 hi = (_Float16)v;
                const _Float16 lo = (_Float16)(v - (float)hi);
                if (r < G::RA) { P0[r * C + c] = hi; P1[r * C + c] = lo; }
            }
        }
    };

    // one convolution on the planes: output row r (of `nrows`) = sum over taps j of operand row r + j * dil; weights w [slab][2][C][16].
    // Taps outermost, this wave's row blocks (wave, wave + 4, ...) in registers: the weight fragments of a tap are fetched once per wave.
    const int m16 = lane & 15, kb = lane >> 4;
    constexpr int NH = C / 16, TPM = 32 / C, MAXB = (((TL + 2 * (AMP_SH + AMP_MAXH2) + 15) / 16) + 3) / 4;
    auto conv_planes = [&](const uint4* __restrict__ w, int dil, int nrows, auto&& emit) {
        const int nblk = (nrows + 15) / 16;
        f32x4 acc[MAXB][NH];
#pragma unroll
        for (int i = 0; i < MAXB; ++i)
#pragma unroll
            for (int nh = 0; nh < NH; ++nh) acc[i][nh] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int nsteps = (k + TPM - 1) / TPM;
        for (int s = 0; s < nsteps; ++s) {
            // lane (n = m16 [+ 16 nh], kb): C = 16: tap 2 s + (kb >> 1), channels 8 (kb & 1) ...; C = 32: tap s, channels 8 kb ...
            const int tap = TPM == 2 ? 2 * s + (kb >> 1) : s;
            const bool tv = tap < k;                                  // a missing tap (odd k at C = 16) multiplies by zero
            const int tapc = tv ? tap : k - 1;
            uint4 bw[NH][2];
#pragma unroll
            for (int nh = 0; nh < NH; ++nh)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const size_t idx = C == 16 ? ((size_t)(tapc * 2 + q) * C + m16) * 2 + (kb & 1)
                                               : ((size_t)((tapc * 2 + (kb >> 1)) * 2 + q) * C + nh * 16 + m16) * 2 + (kb & 1);
                    const uint4 v = w[idx];
                    bw[nh][q] = tv ? v : make_uint4(0, 0, 0, 0);
                }
            const int chunk = C == 16 ? (kb & 1) : kb;
#pragma unroll
            for (int i = 0; i < MAXB; ++i) {
                const int rb = wave + 4 * i;
                if (rb < nblk) {
                    const int row = rb * 16 + m16 + tapc * dil;
                    const uint4 a0 = AP[(size_t)row * NCH + chunk];
                    const uint4 a1 = AP[((size_t)G::RA + row) * NCH + chunk];
#pragma unroll
                    for (int nh = 0; nh < NH; ++nh) {
                        acc[i][nh] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_hf(a1), as_hf(bw[nh][0]), acc[i][nh], 0, 0, 0);
                        acc[i][nh] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_hf(a0), as_hf(bw[nh][1]), acc[i][nh], 0, 0, 0);
                        acc[i][nh] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_hf(a0), as_hf(bw[nh][0]), acc[i][nh], 0, 0, 0);
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < MAXB; ++i) {
            const int rb = wave + 4 * i;
            if (rb < nblk) {
#pragma unroll
                for (int nh = 0; nh < NH; ++nh) emit(rb, nh, acc[i][nh]);
            }
        }
    };

    // ---- P1: snake1 -> planes of a1 ----
    float inv_s1;
    snake_to_planes(XS, x_row0, R0, a1_row0, R1, p.alpha1, p.beta1, xmax, inv_s1);
    __syncthreads();

    // ---- P2: conv1 -> c1 (over the x tile) ----
    float cmax_l = 0.f;
    {
        const float sc = inv_s1 * p.inv_ws1;
        conv_planes(p.w1, d, R2, [&](int rb, int nh, const f32x4& acc) {
            const int n = nh * 16 + m16;
            const float bias = p.bias1 ? p.bias1[n] : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = rb * 16 + 4 * kb + i;
                const int l = c1_row0 + r;
                const float v = fmaf(acc[i], sc, bias);
                if (r < R2) {
                    XS[r * C + n] = v;
                    if ((unsigned)l < (unsigned)L) cmax_l = fmaxf(cmax_l, fabsf(v));
                }
            }
        });
    }
    const float cmax = amp_wg_max(cmax_l, red);                  // (its barriers: c1 complete, a1 no longer read)

    // ---- P3: snake2 -> planes of a2 ----
    float inv_s2;
    snake_to_planes(XS, c1_row0, R2, a2_row0, R3, p.alpha2, p.beta2, cmax, inv_s2);
    __syncthreads();

    // ---- P4: conv2 + bias + x -> y ----
    {
        const float sc = inv_s2 * p.inv_ws2;
        float* yb = p.y + (size_t)b * L * C;
        conv_planes(p.w2, 1, TL, [&](int rb, int nh, const f32x4& acc) {
            const int n = nh * 16 + m16;
            const float bias = p.bias2 ? p.bias2[n] : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int l = t0 + rb * 16 + 4 * kb + i;
                if (l < L) yb[(size_t)l * C + n] = fmaf(acc[i], sc, bias) + xb[(size_t)l * C + n];
            }
        });
    }
}

}  // namespace egr

using namespace egr;

// y = conv2(snake2(conv1(snake1(x)))) + x for one (kernel size k, dilation d) unit of a 16-channel AMP block; x, y [B][L][C] fp32 (y != x).
// w1_h2 / w2_h2: egr_split2h_pack of the slab-major packs of conv1 / conv2 ([k][C][16] with K ordered (tap, ci)), scales w1_scale / w2_scale.
extern "C" int egr_amp_unit_h2(const float* x, float* y, int B, int L, int C, int k, int d, const float* alpha1, const float* beta1, const void* w1_h2,
                               float w1_scale, const float* bias1, const float* alpha2, const float* beta2, const void* w2_h2, float w2_scale,
                               const float* bias2, const float* filt, int aa_taps, void* stream) {
    EGR_CHECK(x && y && x != y && alpha1 && beta1 && alpha2 && beta2 && w1_h2 && w2_h2 && filt, EGR_ERR_ARG, "egr_amp_unit_h2: null argument");
    EGR_CHECK(B >= 1 && B <= 65535 && L >= 16 && w1_scale > 0.f && w2_scale > 0.f, EGR_ERR_ARG, "egr_amp_unit_h2: bad shape / scale");
    if (C != 16 || aa_taps != 12 || k < 1 || k > 11 || (k & 1) == 0 || d < 1 || d * (k - 1) / 2 > AMP_MAXH1 || (((uintptr_t)x) & 15) || (((uintptr_t)y) & 15)) {
        set_error("egr_amp_unit_h2: C = %d, k = %d, d = %d, %d FIR taps do not qualify (C = 16, odd k <= 11, d (k - 1) / 2 <= %d, 12 taps)", C, k, d, aa_taps, AMP_MAXH1);
        return EGR_ERR_UNSUPPORTED;
    }
    AmpP p;
    p.x = x; p.y = y; p.alpha1 = alpha1; p.beta1 = beta1; p.alpha2 = alpha2; p.beta2 = beta2; p.filt = filt;
    p.w1 = (const uint4*)w1_h2; p.w2 = (const uint4*)w2_h2; p.bias1 = bias1; p.bias2 = bias2;
    p.inv_ws1 = 1.0f / w1_scale; p.inv_ws2 = 1.0f / w2_scale;
    p.B = B; p.L = L; p.k = k; p.d = d;
    {
        // tile length 240: the snake phases hand out (16-row run, channel) items to 256 threads, and snake2's 240 + 2 h2 <= 250 rows are exactly
        // one round (snake1: two); 256 rows waste half of a second round (+8 ... +17 % per unit), 496 rows halve the occupancy (+10 ... +20 %):
        // profiles/r05/flashsr_kernel_experiments.log item 5
        constexpr int TL = AMP_TL16;
        constexpr size_t lds = (size_t)AmpGeom<16, TL>::LDS;
        hipLaunchKernelGGL((k_amp_unit<16, TL>), dim3((unsigned)((L + TL - 1) / TL), (unsigned)B), dim3(256), lds, (hipStream_t)stream, p);
    }
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}
