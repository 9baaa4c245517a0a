// Shared parameter block of the implicit-GEMM convolution kernels (egr_nn_gemm.hip: v_mfma_f32_32x32x2_f32 operands;
// egr_nn_gemm_s3.hip: exact three-way bf16 split of the same fp32 operands on v_mfma_f32_32x32x16_bf16).
#pragma once
#include "egr_common.h"

namespace egr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { ACT_NONE = 0, ACT_SILU = 1, ACT_TANH = 2, ACT_LEAKY01 = 3, ACT_LOGCLAMP = 4 };

struct ConvP {
    const float* x;        // [B][H][W][Cin]   (physical; logical input is 2H x 2W when up2 != 0)
    const float* w;        // PACKED [ceil(K/16)][Cout][16], K = KH*KW*Cin ordered (ky, kx, ci)
    const float* bias;     // [Cout] or null
    const float* bias_b;   // [B][Cout] or null (time-embedding bias)
    const float* res;      // [M][Cout] or null
    float* y;              // [M][Cout]
    int B, H, W, Cin, OH, OW, Cout, KH, KW, stride, dil, pad_t, pad_l, up2, act;
    int M, K;
    float act_param;
    // output placement: pixel (b, oy, ox) of the OH x OW grid is written at (b, oy*osy + ooy, ox*osx + oox) of an
    // OHF x OWF image (identity by default); lets four 2x2 "phase" convolutions fill a 2x-upsampled output.
    int osy, osx, ooy, oox, OHF, OWF;
    // split-K (small-M layers): blockIdx.z owns slabs [z*kt_per, ...); raw partial tiles go to ws[z][M][Cout]
    int ksplit, kt_per;
    float* ws;
    // independent problems along blockIdx.z (used when ksplit == 1): element offsets added per z
    long long zx, zw, zy;
    // optional fused input transform (GroupNorm of the producer): x' = x*gn_scale[b][c] + gn_shift[b][c], then SiLU;
    // zero padding applies AFTER it (the reference pads the normalised tensor).  VEC path only.
    const float* gn_scale;
    const float* gn_shift;
    int gn_silu;
    const float* zeros;    // >= 64 zero floats: out-of-image / out-of-range rows read from here (no select needed)
    // three-way bf16 split of w: [ceil(K/16)][3][Cout][16] bf16 (egr_split3_pack); zw counts uint4 (8 bf16) here
    const uint4* w3;
    // z-streaming (k_conv_s3 GEMMs with nz > 1): a workgroup walks zs_nzb consecutive z problems of its (M, N) tile in one
    // slab loop (loads of z+1 in flight while z's tile is stored); 0 = one z per workgroup (blockIdx.z)
    int zs_nzb, nz;
    // k_conv_s3: 1 = walk the (row tile, column tile) space XCD-aware -- workgroup L of a z slice sits on XCD L % 8; the
    // workgroups of one XCD take the column tiles of ONE row tile back to back, so the activation tile they share is an L2 hit
    int xcd_remap;
    // operand scheme of the split kernels: 0 = three bf16 terms (w3 = egr_split3_pack), 1 = two fp16 terms of the pre-scaled
    // operands (w3 = egr_split2h_pack(w, w_scale)).  Scheme 1 scales every BATCH ROW of x by its own power of two, derived in the
    // kernel from row_amax[m / rows_div] (bits of that row's max |x|, written by the producer or by k_absmax_rows): the scaled
    // maximum lands in [2^14, 2^15), whatever the level of the row, and the result of a row depends on that row alone.
    // out_scale (1 / w_scale; 1 for every other kernel) and the row's inverse scale multiply the accumulators before bias /
    // residual / activation.
    int sch;
    float out_scale;
    const unsigned* row_amax;
    int rows_div;
    // scheme 1, optional: out_amax[m / rows_div] is raised to max |y| over the GEMM rows of each batch row (after bias / residual /
    // activation) -- the row maxima of y for the NEXT split contraction that reads it, without a pass over y.  Not with split-K.
    unsigned* out_amax;
    // k_conv3x3_is, optional: GroupNorm partial statistics of y -- part[(unit) * (Cout / 4) + quad] = (sum, sum of squares) over the 32
    // pixels of one sub-tile row and the 4 channels of a quad, unit = ((b * H + y) * (W / 32) + x / 32): the statistics of y are then
    // a fixed-order reduction over 1/32 of y's bytes (egr_groupnorm_stats_from_partials with tiles_per_image = H * W / 32)
    float2* gn_part;
};

// scheme 1: the power of two that brings a row whose largest magnitude has the bits `amax_bits` into [2^14, 2^15), and its
// inverse (both normal floats for every row: the biased exponent e of the maximum is clamped to [15, 254], i.e. rows below 2^-112
// are treated as 2^-112 -- they are zero for every purpose of the graph -- and inf / nan maxima get a scale that lets them
// propagate as they would in fp32).  The epilogue applies the inverse and 1 / w_scale as two exact multiplications.
__device__ __forceinline__ float h2_row_scale(unsigned amax_bits) {
    const int e = min(max((int)((amax_bits >> 23) & 0xffu), 15), 254);
    return __uint_as_float((unsigned)(268 - e) << 23);
}
__device__ __forceinline__ float h2_row_inv(unsigned amax_bits) {
    const int e = min(max((int)((amax_bits >> 23) & 0xffu), 15), 254);
    return __uint_as_float((unsigned)(e - 14) << 23);
}

__device__ __forceinline__ float apply_act(float v, int act, float prm) {
    switch (act) {
        case ACT_SILU: return v / (1.0f + __expf(-v));
        case ACT_TANH: return tanhf(v);
        case ACT_LEAKY01: return v > 0.f ? v : 0.1f * v;
        case ACT_LOGCLAMP: return __logf(fmaxf(v, prm));
        default: return v;
    }
}

// Output rows of one thread: (i, r) -> pixel m.  RES is compile-time so that the residual loads of a 32-row sub-tile are
// issued back to back before any of them is consumed (a run-time `if (p.res)` between them serialises the round trips).
// os_tab (OST): per-row output scales of the block tile in LDS, indexed by m - m0 (scheme 1 of the split kernels)
template <int I, int TM, int TN, bool RES, bool OST>
__device__ __forceinline__ void conv_epilogue_rows_i(const ConvP& p, f32x16 (&acc)[TM][TN], const float (&bv)[TN], float* yout,
                                                     int m0, int n0, int wm0, int wn0, const float* os_tab, unsigned* om_tab = nullptr) {
    const int lane = threadIdx.x & 63, col = lane & 31, rhalf = lane >> 5;
    const bool ident = (p.osy == 1 && p.osx == 1 && p.OHF == p.OH && p.OWF == p.OW);
    const bool decode = !ident || p.bias_b;
    const int nb = n0 + wn0 + col;
    auto place = [&](int m, int& b) -> size_t {          // row offset (elements) of pixel m in y / res, and its batch index
        size_t mo = (size_t)m;
        b = 0;
        if (decode) {
            const int ox = m % p.OW, t = m / p.OW, oy = t % p.OH;
            b = t / p.OH;
            if (!ident) mo = ((size_t)b * p.OHF + (size_t)oy * p.osy + p.ooy) * p.OWF + (size_t)ox * p.osx + p.oox;
        }
        return mo * p.Cout + nb;
    };
    {
        constexpr int i = I;                             // compile-time sub-tile row: no unrolling needed to index acc
        float rv[16][TN];
        if (RES) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * rhalf;
                int b;
                const size_t o = place(m < p.M ? m : 0, b);
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const bool ok = m < p.M && nb + j * 32 < p.Cout;
                    rv[r][j] = p.res[ok ? o + j * 32 : 0];
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * rhalf;
            if (m < p.M) {
                int b;
                const size_t o = place(m, b);
                const float* bb = p.bias_b ? p.bias_b + (size_t)b * p.Cout + nb : nullptr;
                const float os = OST ? os_tab[m - m0] : 1.0f;
                float vm = 0.f;
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    if (nb + j * 32 < p.Cout) {
                        float v = fmaf(OST ? acc[i][j][r] * os : acc[i][j][r], p.out_scale, bv[j]);   // out_scale = 1: the plain sum
                        if (bb) v += bb[j * 32];
                        if (RES) v += rv[r][j];
                        v = apply_act(v, p.act, p.act_param);
                        yout[o + j * 32] = v;
                        vm = fmaxf(vm, fabsf(v));
                    }
                if (OST && om_tab) atomicMax(&om_tab[m / p.rows_div - m0 / p.rows_div], __float_as_uint(vm));
            }
        }
    }
}

template <int TM, int TN, bool RES, bool OST>
__device__ __forceinline__ void conv_epilogue_rows(const ConvP& p, f32x16 (&acc)[TM][TN], const float (&bv)[TN], float* yout,
                                                   int m0, int n0, int wm0, int wn0, const float* os_tab, unsigned* om_tab) {
    conv_epilogue_rows_i<0, TM, TN, RES, OST>(p, acc, bv, yout, m0, n0, wm0, wn0, os_tab, om_tab);
    if constexpr (TM > 1) conv_epilogue_rows_i<1, TM, TN, RES, OST>(p, acc, bv, yout, m0, n0, wm0, wn0, os_tab, om_tab);
    if constexpr (TM > 2) conv_epilogue_rows_i<2, TM, TN, RES, OST>(p, acc, bv, yout, m0, n0, wm0, wn0, os_tab, om_tab);
    if constexpr (TM > 3) conv_epilogue_rows_i<3, TM, TN, RES, OST>(p, acc, bv, yout, m0, n0, wm0, wn0, os_tab, om_tab);
    static_assert(TM <= 4, "add more rows");
}

// Accumulator layout of every 32x32 MFMA tile: register r of lane l holds row (r&3) + 8*(r>>2) + 4*(l>>5), column l&31.
// Writes raw split-K partials, or bias + per-row bias + residual + activation at the (possibly strided) output place.
template <int TM, int TN, bool OST = false>
__device__ __forceinline__ void conv_epilogue(const ConvP& p, f32x16 (&acc)[TM][TN], int m0, int n0, int wm0, int wn0,
                                              float* yb = nullptr, const float* os_tab = nullptr, unsigned* om_tab = nullptr) {
    float* const yout = yb ? yb : p.y;
    const int lane = threadIdx.x & 63, col = lane & 31, rhalf = lane >> 5;
    if (p.ksplit > 1) {        // raw partial sums; bias / residual / activation are applied by k_splitk_reduce
        float* wz = p.ws + (size_t)blockIdx.z * p.M * p.Cout;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * rhalf;
                if (m < p.M) {
                    float* row = wz + (size_t)m * p.Cout + n0 + wn0 + col;
                    const float os = OST ? os_tab[m - m0] : 1.0f;
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        if (n0 + wn0 + j * 32 + col < p.Cout) row[j * 32] = (OST ? acc[i][j][r] * os : acc[i][j][r]) * p.out_scale;
                }
            }
        return;
    }
    float bv[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn0 + j * 32 + col;
        bv[j] = (p.bias && n < p.Cout) ? p.bias[n] : 0.f;
    }
    if (p.res) conv_epilogue_rows<TM, TN, true, OST>(p, acc, bv, yout, m0, n0, wm0, wn0, os_tab, om_tab);
    else conv_epilogue_rows<TM, TN, false, OST>(p, acc, bv, yout, m0, n0, wm0, wn0, os_tab, om_tab);
}

// TRANSPOSED accumulator tiles (kernels that issue mfma(weights, activations): D[i][j] = sum_k w[n = i][k] x[m = j][k]): register r of
// lane l holds output channel (r&3) + 8*(r>>2) + 4*(l>>5) of pixel l&31, i.e. four CONSECUTIVE channels of one pixel per register
// quad -- the tile leaves as 16-byte stores (32 per thread at 64 x 128 per wave instead of 128 four-byte ones; bias / residual reads
// are 16 bytes too).  acc[i][j]: pixels wm0 + 32 i .., channels wn0 + 32 j ..  Cout % 4 != 0 falls back to element stores.
template <int TM, int TN>
__device__ __forceinline__ void store_tile_plain_t(f32x16 (&acc)[TM][TN], float* __restrict__ y, int M, int Cout, int m0, int n0, int wm0,
                                                   int wn0, const float* os_tab, float osw);
template <int TM, int TN, bool VEC, bool RES>
__device__ __forceinline__ void conv_epilogue_t_rows(const ConvP& p, f32x16 (&acc)[TM][TN], int m0, int n0, int wm0, int wn0,
                                                     const float* os_tab, unsigned* om_tab, int sub_stride, float2* part = nullptr) {
    const int lane = threadIdx.x & 63, px = lane & 31, ch4 = 4 * (lane >> 5);
    const bool ident = (p.osy == 1 && p.osx == 1 && p.OHF == p.OH && p.OWF == p.OW);
    const bool decode = !ident || p.bias_b;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + ((wm0 >> 5) + i) * sub_stride + px;        // sub_stride 32: consecutive GEMM rows; W: one image row per sub-tile
        if (m < p.M) {
            const float os = os_tab[wm0 + i * 32 + px], osw = p.out_scale;
            size_t mo = (size_t)m;
            int b = 0;
            if (decode) {
                const int ox = m % p.OW, t = m / p.OW, oy = t % p.OH;
                b = t / p.OH;
                if (!ident) mo = ((size_t)b * p.OHF + (size_t)oy * p.osy + p.ooy) * p.OWF + (size_t)ox * p.osx + p.oox;
            }
            const size_t rowo = mo * p.Cout;
            float vm = 0.f;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float4 rv[4];
                if (RES && VEC) {                    // the residual reads of a 32-channel block go out before any is consumed
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int n = n0 + wn0 + j * 32 + 8 * g + ch4;
                        rv[g] = *(const float4*)(p.res + (n < p.Cout ? rowo + n : 0));
                    }
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = n0 + wn0 + j * 32 + 8 * g + ch4;
                    if (VEC) {
                        if (n < p.Cout) {
                            float4 v = make_float4(acc[i][j][4 * g] * os * osw, acc[i][j][4 * g + 1] * os * osw, acc[i][j][4 * g + 2] * os * osw,
                                                   acc[i][j][4 * g + 3] * os * osw);
                            if (p.bias) { const float4 t = *(const float4*)(p.bias + n); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
                            if (p.bias_b) { const float4 t = *(const float4*)(p.bias_b + (size_t)b * p.Cout + n); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
                            if (RES) { v.x += rv[g].x; v.y += rv[g].y; v.z += rv[g].z; v.w += rv[g].w; }
                            v.x = apply_act(v.x, p.act, p.act_param); v.y = apply_act(v.y, p.act, p.act_param);
                            v.z = apply_act(v.z, p.act, p.act_param); v.w = apply_act(v.w, p.act, p.act_param);
                            *(float4*)(p.y + rowo + n) = v;
                            vm = fmaxf(fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))), vm);
                            if (part) {              // (sum, sum of squares) of this quad over the 32 pixels of the sub-tile: lanes px = 0 write
                                float ps = (v.x + v.y) + (v.z + v.w), pq = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
#pragma unroll
                                for (int o = 16; o >= 1; o >>= 1) { ps += __shfl_xor(ps, o); pq += __shfl_xor(pq, o); }
                                if (px == 0) part[((size_t)(m / 32)) * (p.Cout >> 2) + (n >> 2)] = make_float2(ps, pq);
                            }
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (n + e < p.Cout) {
                                float v = acc[i][j][4 * g + e] * os * osw;
                                if (p.bias) v += p.bias[n + e];
                                if (p.bias_b) v += p.bias_b[(size_t)b * p.Cout + n + e];
                                if (RES) v += p.res[rowo + n + e];
                                v = apply_act(v, p.act, p.act_param);
                                p.y[rowo + n + e] = v;
                                vm = fmaxf(vm, fabsf(v));
                            }
                    }
                }
            }
            if (om_tab) atomicMax(&om_tab[m / p.rows_div - m0 / p.rows_div], __float_as_uint(vm));
        }
    }
}

// The common case without a single data-dependent branch: whole tile inside the problem, identity placement, no per-row bias, no
// activation, Cout % 4 == 0 -- the Winograd GEMMs, the phase convolutions and most 1x1 layers.  (The general routine below tests
// bounds, bias, residual and the activation kind per stored quad: ~10 instructions, a branch and a wait each, 128 times per thread;
// measured 17 % of the K = 1024 GEMM.)  BIAS / RES are compile-time; the residual quads of a sub-tile row are requested together.
template <int TM, int TN, bool BIAS, bool RES, bool PART>
__device__ __forceinline__ void conv_epilogue_t_fast(const ConvP& p, f32x16 (&acc)[TM][TN], int m0, int n0, int wm0, int wn0,
                                                     const float* os_tab, unsigned* om_tab, int sub_stride, float2* part) {
    const int lane = threadIdx.x & 63, px = lane & 31, ch4 = 4 * (lane >> 5);
    const float osw = p.out_scale;
    const int nb = n0 + wn0 + ch4;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + ((wm0 >> 5) + i) * sub_stride + px;
        const float os = os_tab[wm0 + i * 32 + px] * osw;         // (two exact powers of two: their product is exact unless it leaves fp32's range)
        float* row = p.y + (size_t)m * p.Cout + nb;
        float4 rv[TN][4];
        if (RES) {
            const float* rr = p.res + (size_t)m * p.Cout + nb;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) rv[j][g] = *(const float4*)(rr + j * 32 + 8 * g);
        }
        float vm = 0.f;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 v = make_float4(acc[i][j][4 * g] * os, acc[i][j][4 * g + 1] * os, acc[i][j][4 * g + 2] * os, acc[i][j][4 * g + 3] * os);
                if (BIAS) { const float4 t = *(const float4*)(p.bias + nb + j * 32 + 8 * g); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }   // (L1 hits; 64 registers if held)
                if (RES) { v.x += rv[j][g].x; v.y += rv[j][g].y; v.z += rv[j][g].z; v.w += rv[j][g].w; }
                *(float4*)(row + j * 32 + 8 * g) = v;
                vm = fmaxf(fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))), vm);
                if (PART) {                      // GroupNorm partials: (sum, sum of squares) of the quad over the sub-tile's 32 pixels
                    float ps = (v.x + v.y) + (v.z + v.w), pq = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
#pragma unroll
                    for (int o = 16; o >= 1; o >>= 1) { ps += __shfl_xor(ps, o); pq += __shfl_xor(pq, o); }
                    if (px == 0) part[((size_t)(m / 32)) * (p.Cout >> 2) + ((nb + j * 32 + 8 * g) >> 2)] = make_float2(ps, pq);
                }
            }
        if (om_tab) atomicMax(&om_tab[m / p.rows_div - m0 / p.rows_div], __float_as_uint(vm));
    }
}

// om_tab (optional, LDS, zeroed, one word per batch row the block tile touches): raised to max |y| per batch row
// sub_stride: GEMM-row distance between the tile's 32-row sub-tiles (32 = a contiguous tile)
template <int TM, int TN>
__device__ __forceinline__ void conv_epilogue_t(const ConvP& p, f32x16 (&acc)[TM][TN], int m0, int n0, int wm0, int wn0,
                                                const float* os_tab, unsigned* om_tab = nullptr, int sub_stride = 32, float2* part = nullptr) {
    const bool vec = (p.Cout & 3) == 0;
    {
        // (m0 + wm0 + ... : the last row this wave touches; sub-tiles of one wave are consecutive)
        const int m_last = m0 + ((wm0 >> 5) + TM - 1) * sub_stride + 31;
        const bool fast = vec && p.ksplit <= 1 && !p.bias_b && p.act == ACT_NONE && p.osy == 1 && p.osx == 1 && p.OHF == p.OH &&
                          p.OWF == p.OW && m_last < p.M && n0 + wn0 + TN * 32 <= p.Cout;
        if (fast && part) {                           // (the input-stationary 3x3 kernel: bias always, residual on the second conv of a block)
            if (p.bias && p.res) { conv_epilogue_t_fast<TM, TN, true, true, true>(p, acc, m0, n0, wm0, wn0, os_tab, om_tab, sub_stride, part); return; }
            if (p.bias && !p.res) { conv_epilogue_t_fast<TM, TN, true, false, true>(p, acc, m0, n0, wm0, wn0, os_tab, om_tab, sub_stride, part); return; }
        } else if (fast) {                            // wave-uniform
            if (p.bias) {
                if (p.res) conv_epilogue_t_fast<TM, TN, true, true, false>(p, acc, m0, n0, wm0, wn0, os_tab, om_tab, sub_stride, nullptr);
                else conv_epilogue_t_fast<TM, TN, true, false, false>(p, acc, m0, n0, wm0, wn0, os_tab, om_tab, sub_stride, nullptr);
            } else {
                if (p.res) conv_epilogue_t_fast<TM, TN, false, true, false>(p, acc, m0, n0, wm0, wn0, os_tab, om_tab, sub_stride, nullptr);
                else conv_epilogue_t_fast<TM, TN, false, false, false>(p, acc, m0, n0, wm0, wn0, os_tab, om_tab, sub_stride, nullptr);
            }
            return;
        }
    }
    if (p.ksplit > 1) {        // raw partial sums; bias / residual / activation are applied by k_splitk_reduce
        store_tile_plain_t<TM, TN>(acc, p.ws + (size_t)blockIdx.z * p.M * p.Cout, p.M, p.Cout, m0, n0, wm0, wn0, os_tab, p.out_scale);
        return;
    }
    if (vec) {
        if (p.res) conv_epilogue_t_rows<TM, TN, true, true>(p, acc, m0, n0, wm0, wn0, os_tab, om_tab, sub_stride, part);
        else conv_epilogue_t_rows<TM, TN, true, false>(p, acc, m0, n0, wm0, wn0, os_tab, om_tab, sub_stride, part);
    } else {
        if (p.res) conv_epilogue_t_rows<TM, TN, false, true>(p, acc, m0, n0, wm0, wn0, os_tab, om_tab, sub_stride);
        else conv_epilogue_t_rows<TM, TN, false, false>(p, acc, m0, n0, wm0, wn0, os_tab, om_tab, sub_stride);
    }
}

// plain transposed tile store (z-streamed GEMMs: no bias / residual / activation / placement)
template <int TM, int TN>
__device__ __forceinline__ void store_tile_plain_t(f32x16 (&acc)[TM][TN], float* __restrict__ y, int M, int Cout, int m0, int n0, int wm0,
                                                   int wn0, const float* os_tab, float osw) {
    const int lane = threadIdx.x & 63, px = lane & 31, ch4 = 4 * (lane >> 5);
    const bool vec = (Cout & 3) == 0;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm0 + i * 32 + px;
        if (m >= M) continue;
        const float os = os_tab[wm0 + i * 32 + px];
        float* row = y + (size_t)m * Cout;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn0 + j * 32 + 8 * g + ch4;
                if (vec) {
                    if (n < Cout) *(float4*)(row + n) = make_float4(acc[i][j][4 * g] * os * osw, acc[i][j][4 * g + 1] * os * osw, acc[i][j][4 * g + 2] * os * osw,
                                                                    acc[i][j][4 * g + 3] * os * osw);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n + e < Cout) row[n + e] = acc[i][j][4 * g + e] * os * osw;
                }
            }
    }
}

// after the epilogue: the block tile's per-batch-row output maxima (LDS, nrows words) go to out_amax[b_first ...] -- one checked
// atomic per batch row the tile touches.  Every thread of the workgroup must arrive.
__device__ __forceinline__ void out_amax_commit(const ConvP& p, const unsigned* om_tab, int m0, int bm) {
    __syncthreads();
    const int b_first = m0 / p.rows_div, b_last = min(m0 + bm - 1, p.M - 1) / p.rows_div;
    for (int t = threadIdx.x; t <= b_last - b_first; t += blockDim.x) {
        const unsigned bits = om_tab[t];
        unsigned* slot = p.out_amax + (size_t)(b_first + t) * EGR_ROW_AMAX_STRIDE;
        if (bits > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, bits);
    }
}

// egr_nn_gemm.hip: 4 KiB of zeros on the current device (created on first use, one per device)
int zero_page(const float** out);

// egr_nn_gemm_s3.hip: launches k_conv_s3<bm, bn> (bm = s3_bm(...), bn in {32, 64, 128}; grid.x = ceil(M / bm));
// p.w3 must be set and Cin % 16 == 0
int s3_bn(int Cout);
int s3_bm(long long M, int Cout, int bn);
int s3_zs_nzb(long long M, int Cout, int bm, int bn, int nz, int K);
// input-stationary stride-1 1-D convolution (k_conv1d_s3); returns false when the shape does not qualify
bool launch_conv1d_s3(const ConvP& p, hipStream_t st);
// input-stationary 3x3 convolution of big images (k_conv3x3_is, scheme 1; optional fused input GroupNorm); false: does not qualify
bool launch_conv3x3_is(const ConvP& p, hipStream_t st);
void launch_conv_s3(int bm, int bn, dim3 grid, hipStream_t st, const ConvP& p);

}  // namespace egr
