// HBM-bound operators of the FlashSR graph (channels-last activations): GroupNorm(+SiLU), LayerNorm, row
// softmax, element-wise glue, GEGLU, channel concat, anti-aliased snake activation (2x up FIR . snake . 2x
// down FIR fused), transposed-conv overlap-add, STFT magnitude frames for the mel front-end, Philox noise.
#include <math.h>
#include <string.h>

#include <map>
#include <mutex>
#include <vector>

#include <stdlib.h>
#include <string.h>

#include "egr_common.h"
#include "egr_rowmax.h"
#include "egr_fft_device.h"
#include "egr_plan.h"

namespace egr {

// ---------------------------------------------------------------- GroupNorm over [B][HW][C], G groups
// pass 1: per (b, g) sum and sum of squares in double (atomics), pass 2: y = x*scale[b][c] + shift[b][c] (+SiLU)
__global__ __launch_bounds__(256) void k_gn_stats(const float* __restrict__ x, int HW, int C, int G,
                                                   double* __restrict__ stats /*[B][G][2]*/, int slab) {
    // block = (slab of positions, b); thread t handles channel (t % C) for C <= 256 strides
    const int b = blockIdx.y;
    const int p0 = blockIdx.x * slab;
    const int p1 = min(HW, p0 + slab);
    const int cpg = C / G;
    const float* xb = x + (size_t)b * HW * C;
    // each thread walks channels tid, tid+256, ... ; positions p0..p1
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float s = 0.f, ss = 0.f;
        for (int p = p0; p < p1; ++p) {
            const float v = xb[(size_t)p * C + c];
            s += v;
            ss += v * v;
        }
        // combine the cpg channels of a group (adjacent lanes) before touching memory when cpg is a power of 2
        double ds = s, dss = ss;
        bool leader = true;
        if (cpg <= 64 && (cpg & (cpg - 1)) == 0 && (C % 64) == 0) {
            for (int o = 1; o < cpg; o <<= 1) {
                ds += __shfl_xor(ds, o);
                dss += __shfl_xor(dss, o);
            }
            leader = (c % cpg) == 0;
        }
        if (leader) {
            atomicAdd(&stats[((size_t)b * G + c / cpg) * 2 + 0], ds);
            atomicAdd(&stats[((size_t)b * G + c / cpg) * 2 + 1], dss);
        }
    }
}

// Vectorised statistics pass (C % 4 == 0, C/4 <= 256, (C/G) % 4 == 0): a thread owns 4 adjacent channels (one float4, all in
// one group) and every NPR-th position of the block's slab; 4 independent float4 loads are in flight per thread.  Partial
// sums are combined per group in LDS (double) and flushed with one pair of atomics per (block, group).
__global__ __launch_bounds__(256) void k_gn_stats_v4(const float* __restrict__ x, int HW, int C, int G,
                                                      double* __restrict__ stats /*[B][G][2]*/, int slab) {
    __shared__ double gs[64], gss[64];
    const int b = blockIdx.y, C4 = C >> 2, NPR = 256 / C4;
    const int c4 = threadIdx.x % C4, pr = threadIdx.x / C4;
    const int p0 = blockIdx.x * slab, p1 = min(HW, p0 + slab);
    const int cpg = C / G, g = (4 * c4) / cpg;
    if (threadIdx.x < 64) { gs[threadIdx.x] = 0.0; gss[threadIdx.x] = 0.0; }
    __syncthreads();
    if (pr < NPR) {
        const float4* xb = (const float4*)(x + (size_t)b * HW * C) + c4;
        float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;
        int p = p0 + pr;
        for (; p + 3 * NPR < p1; p += 4 * NPR) {
            const float4 v0 = xb[(size_t)p * C4], v1 = xb[(size_t)(p + NPR) * C4], v2 = xb[(size_t)(p + 2 * NPR) * C4],
                         v3 = xb[(size_t)(p + 3 * NPR) * C4];
            s0 += (v0.x + v0.y) + (v0.z + v0.w) + (v2.x + v2.y) + (v2.z + v2.w);
            s1 += (v1.x + v1.y) + (v1.z + v1.w) + (v3.x + v3.y) + (v3.z + v3.w);
            q0 += (v0.x * v0.x + v0.y * v0.y) + (v0.z * v0.z + v0.w * v0.w) + (v2.x * v2.x + v2.y * v2.y) + (v2.z * v2.z + v2.w * v2.w);
            q1 += (v1.x * v1.x + v1.y * v1.y) + (v1.z * v1.z + v1.w * v1.w) + (v3.x * v3.x + v3.y * v3.y) + (v3.z * v3.z + v3.w * v3.w);
        }
        for (; p < p1; p += NPR) {
            const float4 v = xb[(size_t)p * C4];
            s0 += (v.x + v.y) + (v.z + v.w);
            q0 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        }
        atomicAdd(&gs[g], (double)s0 + (double)s1);
        atomicAdd(&gss[g], (double)q0 + (double)q1);
    }
    __syncthreads();
    if (threadIdx.x < G) {
        atomicAdd(&stats[((size_t)b * G + threadIdx.x) * 2 + 0], gs[threadIdx.x]);
        atomicAdd(&stats[((size_t)b * G + threadIdx.x) * 2 + 1], gss[threadIdx.x]);
    }
}

__global__ void k_gn_coeff(const double* __restrict__ stats, const float* __restrict__ gamma,
                           const float* __restrict__ beta, float* __restrict__ scale, float* __restrict__ shift,
                           int B, int C, int G, double count, float eps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * C) return;
    const int b = i / C, c = i - b * C, g = c / (C / G);
    const double mean = stats[((size_t)b * G + g) * 2] / count;
    double var = stats[((size_t)b * G + g) * 2 + 1] / count - mean * mean;
    if (var < 0) var = 0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    const double sc = rstd * (double)gamma[c];
    scale[i] = (float)sc;
    shift[i] = (float)((double)beta[c] - mean * sc);
}

// grid (chunks, B): a workgroup loops over the quads of image b = blockIdx.y; row_amax (optional): row_amax[b] raised to max |y|
__global__ __launch_bounds__(256) void k_affine_c(const float* __restrict__ x, const float* __restrict__ scale,
                                                   const float* __restrict__ shift, float* __restrict__ y,
                                                   long long n4_img, int HW, int C, int silu, unsigned* __restrict__ row_amax) {
    // vectorised over 4 channels; C % 4 == 0
    const int b = blockIdx.y;
    float ym = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4_img;
         i += (long long)gridDim.x * blockDim.x) {
        const long long e = ((long long)b * n4_img + i) * 4;
        const int c = (int)(e % C);
        const float4 v = *(const float4*)(x + e);
        const float4 sc = *(const float4*)(scale + (size_t)b * C + c);
        const float4 sh = *(const float4*)(shift + (size_t)b * C + c);
        float4 o = make_float4(v.x * sc.x + sh.x, v.y * sc.y + sh.y, v.z * sc.z + sh.z, v.w * sc.w + sh.w);
        if (silu) {
            o.x = o.x / (1.f + __expf(-o.x)); o.y = o.y / (1.f + __expf(-o.y));
            o.z = o.z / (1.f + __expf(-o.z)); o.w = o.w / (1.f + __expf(-o.w));
        }
        *(float4*)(y + e) = o;
        ym = amax4(o, ym);
    }
    if (row_amax) row_amax_commit_wg(row_amax + (size_t)b * EGR_ROW_AMAX_STRIDE, ym);
}

// out[b] = max_c |scale[b][c]| * x_amax[b] + max_c |shift[b][c]|  >=  max |x * scale + shift| over image b (and >= the SiLU of it):
// the operand bound of a convolution that applies the GroupNorm in its loader (egr_conv_h2_gn).  One workgroup per image.
__global__ __launch_bounds__(256) void k_gn_bound(const float* __restrict__ scale, const float* __restrict__ shift, int C,
                                                   const float* __restrict__ x_amax, float* __restrict__ out) {
    __shared__ float r0[256], r1[256];
    const int b = blockIdx.x;
    float ms = 0.f, mh = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) {
        ms = fmaxf(ms, fabsf(scale[(size_t)b * C + c]));
        mh = fmaxf(mh, fabsf(shift[(size_t)b * C + c]));
    }
    r0[threadIdx.x] = ms; r1[threadIdx.x] = mh;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) { r0[threadIdx.x] = fmaxf(r0[threadIdx.x], r0[threadIdx.x + o]); r1[threadIdx.x] = fmaxf(r1[threadIdx.x], r1[threadIdx.x + o]); }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[(size_t)b * EGR_ROW_AMAX_STRIDE] = fmaf(r0[0], x_amax[(size_t)b * EGR_ROW_AMAX_STRIDE], r1[0]);
}

// ---------------------------------------------------------------- LayerNorm over rows of C (one wave per row)
// grid (chunks, batch rows): the waves of a workgroup loop over the `per_b` token rows of batch row blockIdx.y;
// row_amax (optional): row_amax[batch row] raised to max |y|
__global__ __launch_bounds__(256) void k_layernorm(const float* __restrict__ x, const float* __restrict__ g,
                                                    const float* __restrict__ b, float* __restrict__ y, int per_b,
                                                    int C, float eps, unsigned* __restrict__ row_amax) {
    const int lane = threadIdx.x & 63;
    float ym = 0.f;
    for (int rb = blockIdx.x * 4 + (threadIdx.x >> 6); rb < per_b; rb += gridDim.x * 4) {
    const size_t row = (size_t)blockIdx.y * per_b + rb;
    const float* xr = x + (size_t)row * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += xr[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / C;
    float v = 0.f;
    for (int c = lane; c < C; c += 64) { const float d = xr[c] - mean; v += d * d; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const float rstd = rsqrtf(v / C + eps);
    float* yr = y + (size_t)row * C;
    for (int c = lane; c < C; c += 64) { const float o = (xr[c] - mean) * rstd * g[c] + b[c]; yr[c] = o; ym = fmaxf(ym, fabsf(o)); }
    }
    if (row_amax) row_amax_commit_wg(row_amax + (size_t)blockIdx.y * EGR_ROW_AMAX_STRIDE, ym);
}

// ---------------------------------------------------------------- row softmax in place (one workgroup per row)
__global__ __launch_bounds__(256) void k_softmax(float* __restrict__ x, int cols) {
    __shared__ float red[4];
    float* r = x + (size_t)blockIdx.x * cols;
    float m = -INFINITY;
    for (int c = threadIdx.x; c < cols; c += 256) m = fmaxf(m, r[c]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int c = threadIdx.x; c < cols; c += 256) { const float e = __expf(r[c] - m); r[c] = e; s += e; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    s = red[0] + red[1] + red[2] + red[3];
    const float inv = 1.0f / s;
    for (int c = threadIdx.x; c < cols; c += 256) r[c] *= inv;
}

// The same with the row held in registers (cols <= 256 NI): one read and one write of the row instead of three and two.  Same
// element-to-thread map and the same order of the per-thread sums as k_softmax, so the results are the same bits.
template <int NI>
__global__ __launch_bounds__(256) void k_softmax_reg(float* __restrict__ x, int cols) {
    __shared__ float red[4];
    float* r = x + (size_t)blockIdx.x * cols;
    float v[NI];
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int c = threadIdx.x + 256 * i;
        v[i] = c < cols ? r[c] : -INFINITY;
        m = fmaxf(m, v[i]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i)
        if (threadIdx.x + 256 * i < cols) { v[i] = __expf(v[i] - m); s += v[i]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    s = red[0] + red[1] + red[2] + red[3];
    const float inv = 1.0f / s;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int c = threadIdx.x + 256 * i;
        if (c < cols) r[c] = v[i] * inv;
    }
}

// ---------------------------------------------------------------- element-wise
enum { EW_ADD = 0, EW_AXPBY = 1, EW_SILU = 2, EW_SCALE = 3, EW_COPY = 4, EW_ADD_SCALE = 5 };
// grid (chunks, rows): a workgroup loops over the `n` elements of row blockIdx.y; row_amax (optional): max |y| per row
__global__ __launch_bounds__(256) void k_eltwise(const float* __restrict__ a, const float* __restrict__ b,
                                                  float* __restrict__ y, long long n, int op, float s0, float s1,
                                                  unsigned* __restrict__ row_amax) {
    const size_t ro = (size_t)blockIdx.y * n;
    a += ro; y += ro;
    if (b) b += ro;
    float ym = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        float v;
        switch (op) {
            case EW_ADD: v = a[i] + b[i]; break;
            case EW_AXPBY: v = s0 * a[i] + s1 * b[i]; break;
            case EW_SILU: v = a[i] / (1.f + __expf(-a[i])); break;
            case EW_SCALE: v = s0 * a[i]; break;
            case EW_ADD_SCALE: v = s0 * __fadd_rn(a[i], b[i]); break;      // the sum rounded first, as add followed by scale
            default: v = a[i]; break;
        }
        y[i] = v;
        ym = fmaxf(ym, fabsf(v));
    }
    if (row_amax) row_amax_commit_wg(row_amax + (size_t)blockIdx.y * EGR_ROW_AMAX_STRIDE, ym);
}

// GEGLU: u [rows][2*D] -> y [rows][D] = u[:, :D] * gelu(u[:, D:])   (exact erf GELU)
// grid (chunks, batch rows): `rows` token rows per batch row; row_amax (optional): max |y| per batch row
__global__ __launch_bounds__(256) void k_geglu(const float* __restrict__ u, float* __restrict__ y, long long rows, int D,
                                                unsigned* __restrict__ row_amax) {
    const long long n = rows * D;
    u += (size_t)blockIdx.y * n * 2;
    y += (size_t)blockIdx.y * n;
    float ym = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / D;
        const int d = (int)(i - r * D);
        const float a = u[r * 2 * D + d], g = u[r * 2 * D + D + d];
        const float o = a * (0.5f * g * (1.0f + erff(g * 0.70710678118654752f)));
        y[i] = o;
        ym = fmaxf(ym, fabsf(o));
    }
    if (row_amax) row_amax_commit_wg(row_amax + (size_t)blockIdx.y * EGR_ROW_AMAX_STRIDE, ym);
}

// channel concat: y[m][0..C1) = a[m], y[m][C1..C1+C2) = b[m]
// grid (chunks, batch rows): M pixel rows per batch row; row_amax (optional): max |y| per batch row
__global__ __launch_bounds__(256) void k_concat(const float* __restrict__ a, const float* __restrict__ b,
                                                 float* __restrict__ y, long long M, int C1, int C2, unsigned* __restrict__ row_amax) {
    const int C = C1 + C2;
    const long long n = M * C;
    a += (size_t)blockIdx.y * M * C1;
    b += (size_t)blockIdx.y * M * C2;
    y += (size_t)blockIdx.y * n;
    float ym = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const long long m = i / C;
        const int c = (int)(i - m * C);
        const float o = c < C1 ? a[m * C1 + c] : b[m * C2 + (c - C1)];
        y[i] = o;
        ym = fmaxf(ym, fabsf(o));
    }
    if (row_amax) row_amax_commit_wg(row_amax + (size_t)blockIdx.y * EGR_ROW_AMAX_STRIDE, ym);
}

// [B][C][H][W] <-> [B][H][W][C] style permutes are done by the host once (weights, tiny tensors); the only
// runtime layout change is the latent/noise NCHW<->NHWC, handled by this generic 2-D transpose per batch.
__global__ __launch_bounds__(256) void k_transpose(const float* __restrict__ x, float* __restrict__ y, int R, int Cc) {
    __shared__ float t[32][33];
    const size_t off = (size_t)blockIdx.z * R * Cc;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int k = ty; k < 32; k += 8)
        if (r0 + k < R && c0 + tx < Cc) t[k][tx] = x[off + (size_t)(r0 + k) * Cc + c0 + tx];
    __syncthreads();
    for (int k = ty; k < 32; k += 8)
        if (c0 + k < Cc && r0 + tx < R) y[off + (size_t)(c0 + k) * R + r0 + tx] = t[tx][k];
}

// ---------------------------------------------------------------- anti-aliased snake over [B][L][C]
// u[i] = 2 * sum_n xp[n] f[i+15-2n] (xp = replicate-pad-5 of x), s = u + sin^2(a u)/(b+1e-9),
// y[l] = sum_k f[k] s[clamp(2l+k-5, 0, 2L-1)]          (12-tap kaiser-sinc, see oracle/flashsr_torch._act_aa)
__global__ __launch_bounds__(256) void k_snake_aa(const float* __restrict__ x, const float* __restrict__ alpha,
                                                   const float* __restrict__ beta, const float* __restrict__ filt,
                                                   float* __restrict__ y, int L, int C, int K) {
    __shared__ float f[32];
    if (threadIdx.x < K) f[threadIdx.x] = filt[threadIdx.x];
    __syncthreads();
    const int b = blockIdx.z;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int l = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (c >= C || l >= L) return;
    const float* xb = x + (size_t)b * L * C + c;
    const float a = __expf(alpha[c]), ib = 1.0f / (__expf(beta[c]) + 1e-9f);
    const int pad = K / 2 - 1;                 // 5
    const int pl = pad * 2 + (K - 2) / 2;      // 15
    const int L2 = 2 * L;
    float acc = 0.f;
    for (int k = 0; k < K; ++k) {
        int i = 2 * l + k - (K / 2 - 1);
        i = i < 0 ? 0 : (i > L2 - 1 ? L2 - 1 : i);
        // u[i]: taps n with 0 <= i + pl - 2n <= K-1
        const int t = i + pl;
        int n_lo = (t - (K - 1) + 1) >> 1;     // ceil((t-K+1)/2), t-K+1 may be negative
        if (2 * n_lo < t - (K - 1)) ++n_lo;
        const int n_hi = t >> 1;
        float u = 0.f;
        for (int n = n_lo; n <= n_hi; ++n) {
            int src = n - pad;
            src = src < 0 ? 0 : (src > L - 1 ? L - 1 : src);
            u += xb[(size_t)src * C] * f[t - 2 * n];
        }
        u *= 2.0f;
        const float sn = __sinf(u * a);
        acc += f[k] * (u + ib * sn * sn);
    }
    y[(size_t)b * L * C + (size_t)l * C + c] = acc;
}


// LDS-tiled version (C % CB == 0): a workgroup owns TL output positions x CB channels.  The x tile (+5 halo,
// replicate-clamped), then every 2x-rate sample s[i] it needs (2*TL+10 per channel, ONE sin each), are staged in
// LDS; the down FIR reads them back.  ~2.3 snake evaluations per output instead of 12.
template <int CB, int TL>
__global__ __launch_bounds__(256) void k_snake_aa_tiled(const float* __restrict__ x, const float* __restrict__ alpha,
                                                        const float* __restrict__ beta, const float* __restrict__ filt,
                                                        float* __restrict__ y, int L, int C) {
    constexpr int PL = 256 / CB, K = 12, XT = TL + 10, ST = 2 * TL + 10;
    __shared__ float xs[XT][CB];
    __shared__ float ss[ST][CB];
    __shared__ float f[K];
    if (threadIdx.x < K) f[threadIdx.x] = filt[threadIdx.x];
    const int c = threadIdx.x % CB, pl = threadIdx.x / CB;
    const int c0 = blockIdx.x * CB, l0 = blockIdx.y * TL, b = blockIdx.z;
    const float* xb = x + (size_t)b * L * C + c0 + c;
    for (int p = pl; p < XT; p += PL) {
        int src = l0 - 5 + p;
        src = src < 0 ? 0 : (src > L - 1 ? L - 1 : src);
        xs[p][c] = xb[(size_t)src * C];
    }
    const float a = __expf(alpha[c0 + c]), ib = 1.0f / (__expf(beta[c0 + c]) + 1e-9f);
    __syncthreads();
    const int L2 = 2 * L;
    for (int q = pl; q < ST; q += PL) {
        int i = 2 * l0 - 5 + q;
        i = i < 0 ? 0 : (i > L2 - 1 ? L2 - 1 : i);
        const int t = i + 15;
        const int n_hi = t >> 1;                       // taps n_hi-5 .. n_hi, filter index t - 2n
        float u = 0.f;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int n = n_hi - j;
            int p = n - l0;                            // xs row of xp[n] (tile rows are already clamped copies)
            p = p < 0 ? 0 : (p > XT - 1 ? XT - 1 : p);
            u += xs[p][c] * f[t - 2 * n];
        }
        u *= 2.0f;
        const float sn = __sinf(u * a);
        ss[q][c] = u + ib * sn * sn;
    }
    __syncthreads();
    float* yb = y + (size_t)b * L * C + c0 + c;
    for (int j = pl; j < TL; j += PL) {
        const int l = l0 + j;
        if (l >= L) break;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) acc += f[k] * ss[2 * j + k][c];
        yb[(size_t)l * C] = acc;
    }
}

// Register-blocked version (K = 12): a thread owns one channel and a run of J consecutive output positions.  Its x window
// (J + 10 values, lanes run over channels so every load is a coalesced row segment) and the 2J + 10 up-rate samples it needs
// live in registers with compile-time indices: no LDS, no barriers, (2J+10)/J sin per output (2.6 at J = 16) and 6 + ~6 FMAs
// per up-rate sample.  Halo re-reads (10/J of the tile) hit L1/L2.  Runs at the tensor edges substitute the clamped
// up-rate samples with predicated selects (same code path, same arithmetic).
template <int J>
__global__ __launch_bounds__(256) void k_snake_aa_reg(const float* __restrict__ x, const float* __restrict__ alpha,
                                                      const float* __restrict__ beta, const float* __restrict__ filt,
                                                      float* __restrict__ y, int L, int C, int nruns,
                                                      unsigned* __restrict__ row_amax) {
    // grid (row chunks, B): a workgroup loops over the (run, channel) items of batch row b = blockIdx.y.
    // row_amax (optional): row_amax[b] raised to max |y| of batch row b (the 1-D convolution that reads y scales each row from it)
    const int b = blockIdx.y;
    const long long per_row = (long long)nruns * C;
    float ymax = 0.f;
    for (long long gid = (long long)blockIdx.x * 256 + threadIdx.x; gid < per_row; gid += (long long)gridDim.x * 256) {
    const int c = (int)(gid % C);
    const int run = (int)(gid / C);
    const int l0 = run * J;
    float f[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) f[k] = filt[k];
    const float a = __expf(alpha[c]), ib = 1.0f / (__expf(beta[c]) + 1e-9f);
    const float* xb = x + (size_t)b * L * C + c;
    float* yb = y + (size_t)b * L * C + c;
    const int L2 = 2 * L;
    float xw[J + 10];                            // xw[i] = xp[l0 + i] = x[clamp(l0 - 5 + i)] (replicate padding of 5)
#pragma unroll
    for (int i = 0; i < J + 10; ++i) {
        int src = l0 - 5 + i;
        src = src < 0 ? 0 : (src > L - 1 ? L - 1 : src);
        xw[i] = xb[(size_t)src * C];
    }
    float sv[2 * J + 10];                        // up-rate sample i = 2 l0 - 5 + q: taps xw[5 + (q>>1) - j] * f[(q&1) + 2j]
#pragma unroll
    for (int q = 0; q < 2 * J + 10; ++q) {
        float u = 0.f;
#pragma unroll
        for (int j = 0; j < 6; ++j) u += xw[5 + (q >> 1) - j] * f[(q & 1) + 2 * j];
        u *= 2.0f;
        const float sn = __sinf(u * a);
        sv[q] = u + ib * sn * sn;
    }
    const int i0 = 2 * l0 - 5;
    if (i0 < 0 || i0 + 2 * J + 9 > L2 - 1) {     // runs at the tensor edges: up-rate indices clamp to [0, 2L-1]
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
    for (int jj = 0; jj < J; ++jj) {             // output l0 + jj = sum_k f[k] * s[2 (l0 + jj) - 5 + k]
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 12; ++k) acc += f[k] * sv[2 * jj + k];
        if (l0 + jj < L) { yb[(size_t)(l0 + jj) * C] = acc; ymax = fmaxf(ymax, fabsf(acc)); }
    }
    }
    // (measured inside the forward, 91 launches: 146 us per launch without the maxima; with them 156 us at <= 8192 workgroups and one
    // checked commit per workgroup, 181 us at 32768, 165 - 202 us with per-wave or unchecked commits; one commit per workgroup of a
    // 100 k-workgroup grid: 2.7 ms -- same-line atomics serialise at ~25 ns)
    if (row_amax) row_amax_commit_wg(row_amax + (size_t)b * EGR_ROW_AMAX_STRIDE, ymax);          // (uniform branch: every thread arrives)
}

// ---------------------------------------------------------------- ConvTranspose1d overlap-add (gather form)
// Y [B][Lin][K][Co] (GEMM output), out[b][o][co] = bias[co] + (add ? add[b][o][co] : 0)
//                                              + sum_{k == (o+pad) mod r, i=(o+pad-k)/r in [0,Lin)} Y[b][i][k][co]
__global__ __launch_bounds__(256) void k_col2im1d(const float* __restrict__ Y, const float* __restrict__ bias,
                                                   const float* __restrict__ add, float* __restrict__ out, int Lin,
                                                   int Lout, int Co, int K, int r, int pad, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const int co = (int)(i % Co);
        const long long t = i / Co;
        const int o = (int)(t % Lout);
        const long long b = t / Lout;
        float v = bias ? bias[co] : 0.f;
        if (add) v += add[i];
        const int q = o + pad;
        for (int k = q % r; k < K; k += r) {
            const int ii = (q - k) / r;
            if (q - k >= 0 && ii < Lin) v += Y[(((size_t)b * Lin + ii) * K + k) * Co + co];
        }
        out[i] = v;
    }
}

// ---------------------------------------------------------------- mel front-end: STFT magnitude frames
// x [B][L] -> mag [B][T][ldm] (ldm >= n_fft/2+1, tail zero-filled), reflect padding of `rpad` samples on both
// sides, caller-supplied window; frames t >= t_valid are zero (they become log(floor) after the mel GEMM).
__global__ __launch_bounds__(256) void k_stft_frames(const float* __restrict__ x, int L, int n_fft, int hop, int rpad,
                                                      int T, int t_valid, int ldm, const float* __restrict__ window,
                                                      FftDesc fd, const cplx* __restrict__ tw,
                                                      const cplx* __restrict__ wsplit, float* __restrict__ mag) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int Mh = n_fft / 2;
    cplx* cur = (cplx*)smem;
    cplx* alt = cur + Mh;
    const int t = blockIdx.x, b = blockIdx.y;
    float* o = mag + ((size_t)b * T + t) * ldm;
    if (t >= t_valid) {
        for (int k = threadIdx.x; k < ldm; k += blockDim.x) o[k] = 0.f;
        return;
    }
    const float* xb = x + (size_t)b * L;
    const int s0 = t * hop - rpad;
    for (int e = threadIdx.x; e < Mh; e += blockDim.x) {
        float v[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int i = s0 + 2 * e + h;
            if (i < 0) i = -i;                      // reflect (no edge repeat), as F.pad(mode="reflect")
            if (i > L - 1) i = 2 * (L - 1) - i;
            i = i < 0 ? 0 : (i > L - 1 ? L - 1 : i);
            v[h] = xb[i] * window[2 * e + h];
        }
        cur[e] = make_float2(v[0], v[1]);
    }
    __syncthreads();
    lds_fft<false>(cur, alt, fd, tw, 1, 0, 1, Mh, false);
    for (int k = threadIdx.x; k < ldm; k += blockDim.x) {
        float r = 0.f;
        if (k <= Mh) {
            const cplx Za = cur[k == Mh ? 0 : k];
            const cplx Zb = cur[(k == 0 || k == Mh) ? 0 : Mh - k];
            const cplx E = make_float2(0.5f * (Za.x + Zb.x), 0.5f * (Za.y - Zb.y));
            const cplx O = make_float2(0.5f * (Za.y + Zb.y), -0.5f * (Za.x - Zb.x));
            const cplx X = cadd(E, cmul(wsplit[k], O));
            r = sqrtf(X.x * X.x + X.y * X.y);
        }
        o[k] = r;
    }
}

// ---------------------------------------------------------------- Philox4x32-10 standard normals
// element e of stream (seed, row) is a pure function of (seed, row, e): results do not depend on rank count.
__device__ __forceinline__ void philox_round(unsigned& c0, unsigned& c1, unsigned& c2, unsigned& c3, unsigned k0,
                                             unsigned k1) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
    const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
    const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}

__global__ __launch_bounds__(256) void k_randn(float* __restrict__ out, long long per_row, int rows,
                                                unsigned long long seed, const long long* __restrict__ row_ids, long long id_base) {
    const long long quads = (per_row + 3) / 4;
    const long long n = quads * rows;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / quads);
        const long long q = i - (long long)r * quads;
        const unsigned long long rid = row_ids ? (unsigned long long)row_ids[r] : (unsigned long long)(id_base + r);
        unsigned c0 = (unsigned)q, c1 = (unsigned)(q >> 32), c2 = (unsigned)rid, c3 = (unsigned)(rid >> 32);
        unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
        for (int it = 0; it < 10; ++it) {
            philox_round(c0, c1, c2, c3, k0, k1);
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        // two Box-Muller pairs
        const float u0 = ((float)c0 + 0.5f) * 2.3283064365386963e-10f, u1 = ((float)c1 + 0.5f) * 2.3283064365386963e-10f;
        const float u2 = ((float)c2 + 0.5f) * 2.3283064365386963e-10f, u3 = ((float)c3 + 0.5f) * 2.3283064365386963e-10f;
        const float r0 = sqrtf(-2.0f * logf(u0)), r1 = sqrtf(-2.0f * logf(u2));
        float z[4];
        z[0] = r0 * cosf(6.283185307179586f * u1);
        z[1] = r0 * sinf(6.283185307179586f * u1);
        z[2] = r1 * cosf(6.283185307179586f * u3);
        z[3] = r1 * sinf(6.283185307179586f * u3);
        float* o = out + (size_t)r * per_row + q * 4;
        for (int j = 0; j < 4; ++j)
            if (q * 4 + j < per_row) o[j] = z[j];
    }
}

// ---------------------------------------------------------------- Winograd F(2x2,3x3) transforms (stride-1, pad-1 3x3 convs)
// V[xi][t][c] = (B^T d B)[i][j], xi = 4i+j, d = 4x4 input patch of output tile t = (b, ty, tx) (2x2 outputs), zero padded.
__global__ __launch_bounds__(256) void k_wino_in(const float* __restrict__ x, int B, int H, int W, int C, int TH, int TW,
                                                  const float* __restrict__ gsc, const float* __restrict__ gsh, int gsilu,
                                                  float* __restrict__ V) {
    const int C4 = C >> 2;
    const long long P = (long long)B * TH * TW, total = P * C4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        const long long t = i / C4;
        const int tx = (int)(t % TW), ty = (int)((t / TW) % TH), b = (int)(t / ((long long)TW * TH));
        float4 d[4][4];
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gsc) {
            sc = *(const float4*)(gsc + (size_t)b * C + 4 * c4);
            sh = *(const float4*)(gsh + (size_t)b * C + 4 * c4);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int iy = 2 * ty - 1 + r;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ix = 2 * tx - 1 + q;
                const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                const size_t off = ok ? (((size_t)b * H + iy) * W + ix) * C + 4 * c4 : (size_t)(4 * c4);
                float4 v = *(const float4*)(x + off);
                if (gsc) {      // producer's GroupNorm (+SiLU) fused into the transform; padding stays zero
                    v = make_float4(v.x * sc.x + sh.x, v.y * sc.y + sh.y, v.z * sc.z + sh.z, v.w * sc.w + sh.w);
                    if (gsilu) {
                        v.x = v.x / (1.f + __expf(-v.x)); v.y = v.y / (1.f + __expf(-v.y));
                        v.z = v.z / (1.f + __expf(-v.z)); v.w = v.w / (1.f + __expf(-v.w));
                    }
                }
                d[r][q] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#define F4OP(a, op, b) make_float4(a.x op b.x, a.y op b.y, a.z op b.z, a.w op b.w)
        float4 r_[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {                      // B^T d : combine rows
            r_[0][q] = F4OP(d[0][q], -, d[2][q]);
            r_[1][q] = F4OP(d[1][q], +, d[2][q]);
            r_[2][q] = F4OP(d[2][q], -, d[1][q]);
            r_[3][q] = F4OP(d[1][q], -, d[3][q]);
        }
        float* o = V + (size_t)t * C + 4 * c4;
        const size_t zs = (size_t)P * C;
#pragma unroll
        for (int r = 0; r < 4; ++r) {                      // (.) B : combine columns
            *(float4*)(o + (size_t)(4 * r + 0) * zs) = F4OP(r_[r][0], -, r_[r][2]);
            *(float4*)(o + (size_t)(4 * r + 1) * zs) = F4OP(r_[r][1], +, r_[r][2]);
            *(float4*)(o + (size_t)(4 * r + 2) * zs) = F4OP(r_[r][2], -, r_[r][1]);
            *(float4*)(o + (size_t)(4 * r + 3) * zs) = F4OP(r_[r][1], -, r_[r][3]);
        }
    }
}

// y[b][2ty+a][2tx+bb][n] = act((A^T M A)[a][bb] + bias[n] + res[...]),  M[xi][t][n]
__global__ __launch_bounds__(256) void k_wino_out(const float* __restrict__ Mx, const float* __restrict__ bias,
                                                   const float* __restrict__ res, int B, int H, int W, int N, int TH, int TW,
                                                   int act, float* __restrict__ y) {
    const int N4 = N >> 2;
    const long long P = (long long)B * TH * TW, total = P * N4;
    const size_t zs = (size_t)P * N;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int n4 = (int)(i % N4);
        const long long t = i / N4;
        const int tx = (int)(t % TW), ty = (int)((t / TW) % TH), b = (int)(t / ((long long)TW * TH));
        const float* mi = Mx + (size_t)t * N + 4 * n4;
        float4 m[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) m[r][q] = *(const float4*)(mi + (size_t)(4 * r + q) * zs);
        float4 s0[4], s1[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {                      // A^T m
            s0[q] = F4OP(F4OP(m[0][q], +, m[1][q]), +, m[2][q]);
            s1[q] = F4OP(F4OP(m[1][q], -, m[2][q]), -, m[3][q]);
        }
        float4 o[2][2];
        o[0][0] = F4OP(F4OP(s0[0], +, s0[1]), +, s0[2]);
        o[0][1] = F4OP(F4OP(s0[1], -, s0[2]), -, s0[3]);
        o[1][0] = F4OP(F4OP(s1[0], +, s1[1]), +, s1[2]);
        o[1][1] = F4OP(F4OP(s1[1], -, s1[2]), -, s1[3]);
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias) bv = *(const float4*)(bias + 4 * n4);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int oy = 2 * ty + a, ox = 2 * tx + c;
                if (oy >= H || ox >= W) continue;
                const size_t off = (((size_t)b * H + oy) * W + ox) * N + 4 * n4;
                float4 v = F4OP(o[a][c], +, bv);
                if (res) { const float4 rr = *(const float4*)(res + off); v = F4OP(v, +, rr); }
                if (act == 1) {
                    v.x = v.x / (1.f + __expf(-v.x)); v.y = v.y / (1.f + __expf(-v.y));
                    v.z = v.z / (1.f + __expf(-v.z)); v.w = v.w / (1.f + __expf(-v.w));
                }
                *(float4*)(y + off) = v;
            }
    }
#undef F4OP
}

// ---------------------------------------------------------------- input low-pass (lowpass_input=True)
// cutoff bin per row from STFT magnitudes mag [B][T][ldm]: e[k] = sum_t mag[t][k]; c = cumsum(e); the cutoff is the
// highest bin whose cumulative energy is still below pct * c[nb-1] (scanning down from the top), plus one.
__global__ __launch_bounds__(256) void k_cutoff_bin(const float* __restrict__ mag, int T, int ldm, int nb, float pct,
                                                     int* __restrict__ cut) {
    extern __shared__ float e[];
    const int b = blockIdx.x;
    const float* m = mag + (size_t)b * T * ldm;
    for (int k = threadIdx.x; k < nb; k += blockDim.x) {
        float s = 0.f;
        for (int t = 0; t < T; ++t) s += m[(size_t)t * ldm + k];
        e[k] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double c = 0.0;
        for (int k = 0; k < nb; ++k) c += e[k];
        const double lim = c * (double)pct;
        double run = c;
        int res = 0;
        for (int i = 1; i < nb; ++i) {          // run = cumulative energy up to bin nb-i
            if (run < lim) { res = nb - i; break; }
            run -= e[nb - i];
        }
        cut[b] = res;
    }
}

// zero-phase (forward-backward) Chebyshev-I low-pass amplitude gain 1 / (1 + eps^2 T_n^2(W/Wc)), W = tan(pi f/fs)
__global__ __launch_bounds__(256) void k_cheby_gain(const int* __restrict__ cut, int nb_stft, float sr, int order,
                                                     float eps2, long long nbins, float* __restrict__ gain) {
    const int b = blockIdx.y;
    const float fc = fmaxf(1.0f, fminf((float)cut[b] / (float)(nb_stft - 1), 0.999f) * 0.5f * sr);
    const float wc = tanf(3.14159265358979f * fc / sr);
    float* g = gain + (size_t)b * nbins;
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < nbins; k += (long long)gridDim.x * blockDim.x) {
        const float f = 0.5f * sr * (float)k / (float)(nbins - 1);
        float v = 0.f;
        if (f < 0.4999f * sr) {
            const float xw = tanf(3.14159265358979f * f / sr) / wc;
            const float tn = xw <= 1.f ? cosf(order * acosf(xw)) : coshf(order * acoshf(xw));
            v = 1.0f / (1.0f + eps2 * tn * tn);
        }
        g[k] = v;
    }
}

struct FrameTables { FftDesc fd; cplx *tw, *wsplit; };
static std::mutex g_mu;
static std::map<std::pair<int, int>, FrameTables> g_tabs;

static int frame_tables(int n_fft, FrameTables* out) {
    int dev = 0;
    EGR_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_tabs.find({dev, n_fft});
    if (it != g_tabs.end()) { *out = it->second; return EGR_OK; }
    FrameTables t;
    const int Mh = n_fft / 2;
    EGR_CHECK(make_schedule(Mh, &t.fd, 127), EGR_ERR_UNSUPPORTED, "n_fft=%d: n_fft/2 has a prime factor above 127", n_fft);
    std::vector<float2> h;
    make_twiddles(h, Mh, 1, Mh);
    EGR_HIP(hipMalloc((void**)&t.tw, h.size() * sizeof(float2)));
    EGR_HIP(hipMemcpy(t.tw, h.data(), h.size() * sizeof(float2), hipMemcpyHostToDevice));
    make_twiddles(h, Mh + 1, 1, n_fft);
    EGR_HIP(hipMalloc((void**)&t.wsplit, h.size() * sizeof(float2)));
    EGR_HIP(hipMemcpy(t.wsplit, h.data(), h.size() * sizeof(float2), hipMemcpyHostToDevice));
    g_tabs[{dev, n_fft}] = t;
    *out = t;
    return EGR_OK;
}

// Thin ends of the VAE.  (1) A 3x3 convolution with very few outputs (conv_out: 128 -> 1) as a 1x1 contraction onto the kh*kw*Cout
// per-tap partial products P[pixel][tap*Cout + co] (one ninth of the MFMA work an im2col tile with 31 idle columns needs)
// followed by this gather: y[b,oy,ox,co] = bias[co] + sum over taps of P at the tap's source pixel (outside the image: nothing,
// which is the zero padding).  (2) A convolution with ONE input channel (conv_in: 1 -> 128): K = kh*kw is below one MFMA slab;
// each thread forms 4 output channels of a pixel from the kh*kw input samples on the vector ALU; the layer is bound by its
// output write.
__global__ __launch_bounds__(256) void k_tap_gather(const float* __restrict__ P, const float* __restrict__ bias, float* __restrict__ y,
                                                    int B, int H, int W, int KH, int KW, int Cout, int pad_t, int pad_l) {
    const long long total = (long long)B * H * W * Cout;
    const int ldp = KH * KW * Cout;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int co = (int)(i % Cout);
        const long long pix = i / Cout;
        const int ox = (int)(pix % W), oy = (int)((pix / W) % H);
        const long long b = pix / ((long long)W * H);
        float acc = bias ? bias[co] : 0.f;
        for (int ky = 0; ky < KH; ++ky) {
            const int iy = oy + ky - pad_t;
            if ((unsigned)iy >= (unsigned)H) continue;
            for (int kx = 0; kx < KW; ++kx) {
                const int ix = ox + kx - pad_l;
                if ((unsigned)ix < (unsigned)W) acc += P[((b * H + iy) * W + ix) * ldp + (ky * KW + kx) * Cout + co];
            }
        }
        y[i] = acc;
    }
}

// w: the packed layout of a K = KH*KW (<= 16) weight, [Cout][16] with k contiguous.  A thread keeps ONE quad of output channels
// (its 4 x KH*KW weights live in registers) and walks pixels; the 256 / (Cout/4) pixels of a block iteration are consecutive, the
// lanes of a pixel read the same input samples (one transaction) and store 16 bytes each of one contiguous output row.
template <int KK>
__global__ __launch_bounds__(256) void k_conv_cin1(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                   float* __restrict__ y, int B, int H, int W, int Cout, int KW, int pad_t, int pad_l) {
    const int c4n = Cout / 4, ppi = 256 / c4n;                 // pixels per block iteration
    const int c0 = (threadIdx.x % c4n) * 4, pl = threadIdx.x / c4n;
    float wr[KK][4];
#pragma unroll
    for (int k = 0; k < KK; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) wr[k][j] = w[(c0 + j) * 16 + k];
    const float4 b4 = bias ? *(const float4*)(bias + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
    const unsigned npix = (unsigned)B * H * W;                 // < 2^31 (checked by the launcher): 32-bit index math
    for (unsigned pix = blockIdx.x * ppi + pl; pix < npix; pix += gridDim.x * ppi) {
        const unsigned row = pix / (unsigned)W;
        const int ox = (int)(pix - row * W), oy = (int)(row % (unsigned)H);
        const size_t b = row / (unsigned)H;
        float4 acc = b4;
#pragma unroll
        for (int k = 0; k < KK; ++k) {
            const int ky = k / KW, kx = k - ky * KW;
            const int iy = oy + ky - pad_t, ix = ox + kx - pad_l;
            const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            const float v = ok ? x[(b * H + iy) * W + ix] : 0.f;
            acc.x = fmaf(v, wr[k][0], acc.x);
            acc.y = fmaf(v, wr[k][1], acc.y);
            acc.z = fmaf(v, wr[k][2], acc.z);
            acc.w = fmaf(v, wr[k][3], acc.w);
        }
        *(float4*)(y + (size_t)pix * Cout + c0) = acc;
    }
}

static inline int grid1d(long long n) {
    long long b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace egr

using namespace egr;

extern "C" int egr_groupnorm_nhwc_ra(const float* x, const float* gamma, const float* beta, float* y, int B, int HW, int C,
                                     int G, float eps, int silu, void* workspace, float* row_amax, void* stream);
extern "C" int egr_groupnorm_nhwc(const float* x, const float* gamma, const float* beta, float* y, int B, int HW, int C,
                                  int G, float eps, int silu, void* workspace, void* stream) {
    return egr_groupnorm_nhwc_ra(x, gamma, beta, y, B, HW, C, G, eps, silu, workspace, nullptr, stream);
}

// The `_ra` forms of the element-wise operators: same outputs, and row_amax[b] (optional; floats the caller zeroed) is raised to
// max |y| over batch row b -- the operand maxima of the split contraction that reads y (egr_conv_h2) without a pass over y.
extern "C" int egr_groupnorm_nhwc_ra(const float* x, const float* gamma, const float* beta, float* y, int B, int HW, int C,
                                     int G, float eps, int silu, void* workspace, float* row_amax, void* stream) {
    EGR_CHECK(x && gamma && beta && y && workspace && B <= 65535, EGR_ERR_ARG, "null argument");
    EGR_CHECK(B >= 1 && HW >= 1 && C >= 4 && G >= 1 && C % G == 0 && C % 4 == 0, EGR_ERR_ARG, "bad groupnorm geometry");
    hipStream_t st = (hipStream_t)stream;
    // workspace layout: double stats[B*G*2] | float scale[B*C] | float shift[B*C]
    double* stats = (double*)workspace;
    float* scale = (float*)(stats + (size_t)B * G * 2);
    float* shift = scale + (size_t)B * C;
    EGR_HIP(hipMemsetAsync(stats, 0, sizeof(double) * B * G * 2, st));
    int slab = (HW + 255) / 256;
    if (slab < 16) slab = HW < 16 ? HW : 16;
    const int nslab = (HW + slab - 1) / slab;
    if (C % 4 == 0 && C / 4 <= 256 && (C / G) % 4 == 0 && G <= 64 && (((uintptr_t)x) & 15) == 0) {
        int slab4 = (HW + 511) / 512;                   // >= 512 workgroups per image row, >= 64 positions each
        if (slab4 < 64) slab4 = HW < 64 ? HW : 64;
        hipLaunchKernelGGL(k_gn_stats_v4, dim3((HW + slab4 - 1) / slab4, B), dim3(256), 0, st, x, HW, C, G, stats, slab4);
    } else
        hipLaunchKernelGGL(k_gn_stats, dim3(nslab, B), dim3(C < 256 ? ((C + 63) / 64) * 64 : 256), 0, st, x, HW, C, G, stats, slab);
    hipLaunchKernelGGL(k_gn_coeff, dim3((B * C + 255) / 256), dim3(256), 0, st, stats, gamma, beta, scale, shift, B, C, G,
                       (double)HW * (C / G), eps);
    const long long n4 = (long long)HW * C / 4;
    hipLaunchKernelGGL(k_affine_c, dim3(row_grid_x(n4, B, 8192, 2), (unsigned)B), dim3(256), 0, st, x, scale, shift, y, n4, HW, C, silu,
                       (unsigned*)row_amax);
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}

// statistics + per-(b, c) scale/shift only; the normalisation itself is applied by the consumer
// (egr_conv_nhwc_gn / egr_winograd_input) while it loads the tensor.
extern "C" int egr_groupnorm_coeff(const float* x, const float* gamma, const float* beta, int B, int HW, int C, int G,
                                   float eps, void* workspace, float* scale, float* shift, void* stream) {
    EGR_CHECK(x && gamma && beta && workspace && scale && shift, EGR_ERR_ARG, "null argument");
    EGR_CHECK(B >= 1 && HW >= 1 && C >= 4 && G >= 1 && C % G == 0, EGR_ERR_ARG, "bad groupnorm geometry");
    hipStream_t st = (hipStream_t)stream;
    double* stats = (double*)workspace;
    EGR_HIP(hipMemsetAsync(stats, 0, sizeof(double) * B * G * 2, st));
    int slab = (HW + 255) / 256;
    if (slab < 16) slab = HW < 16 ? HW : 16;
    const int nslab = (HW + slab - 1) / slab;
    if (C % 4 == 0 && C / 4 <= 256 && (C / G) % 4 == 0 && G <= 64 && (((uintptr_t)x) & 15) == 0) {
        int slab4 = (HW + 511) / 512;                   // >= 512 workgroups per image row, >= 64 positions each
        if (slab4 < 64) slab4 = HW < 64 ? HW : 64;
        hipLaunchKernelGGL(k_gn_stats_v4, dim3((HW + slab4 - 1) / slab4, B), dim3(256), 0, st, x, HW, C, G, stats, slab4);
    } else
        hipLaunchKernelGGL(k_gn_stats, dim3(nslab, B), dim3(C < 256 ? ((C + 63) / 64) * 64 : 256), 0, st, x, HW, C, G, stats, slab);
    hipLaunchKernelGGL(k_gn_coeff, dim3((B * C + 255) / 256), dim3(256), 0, st, stats, gamma, beta, scale, shift, B, C, G,
                       (double)HW * (C / G), eps);
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}

// scale / shift from statistics that already exist (stats[B][G][2] double: sum, sum of squares over HW * C/G elements)
extern "C" int egr_groupnorm_coeff_from_stats(const double* stats, const float* gamma, const float* beta, int B, int HW, int C,
                                              int G, float eps, float* scale, float* shift, void* stream) {
    EGR_CHECK(stats && gamma && beta && scale && shift && B >= 1 && HW >= 1 && C >= 1 && G >= 1 && C % G == 0, EGR_ERR_ARG,
              "bad argument");
    hipLaunchKernelGGL(k_gn_coeff, dim3((B * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, stats, gamma, beta, scale, shift,
                       B, C, G, (double)HW * (C / G), eps);
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}

// bound[b] (row_amax layout) = max_c |scale[b][c]| * x_row_amax[b] + max_c |shift[b][c]|: an upper bound of max |GroupNorm(x)| per
// image from the per-(image, channel) coefficients and the row maxima of x -- no pass over x.  Feeds egr_conv_h2_gn.
extern "C" int egr_gn_operand_bound(const float* scale, const float* shift, int B, int C, const float* x_row_amax, float* bound, void* stream) {
    EGR_CHECK(scale && shift && x_row_amax && bound && B >= 1 && C >= 1, EGR_ERR_ARG, "bad argument");
    hipLaunchKernelGGL(k_gn_bound, dim3(B), dim3(256), 0, (hipStream_t)stream, scale, shift, C, x_row_amax, bound);
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}

extern "C" size_t egr_groupnorm_workspace_bytes(int B, int C, int G) {
    return sizeof(double) * (size_t)B * G * 2 + sizeof(float) * 2 * (size_t)B * C + 64;
}

// batch_rows equal consecutive groups of the `rows` token rows (1: no grouping)
extern "C" int egr_layernorm_rows_ra(const float* x, const float* gamma, const float* beta, float* y, int64_t rows, int C,
                                     float eps, int batch_rows, float* row_amax, void* stream) {
    EGR_CHECK(x && gamma && beta && y && rows >= 1 && C >= 1 && batch_rows >= 1 && batch_rows <= 65535 && rows % batch_rows == 0, EGR_ERR_ARG,
              "bad argument");
    const long long per_b = rows / batch_rows;
    long long nx = (per_b + 3) / 4, cap = std::max<long long>(1, 16384 / batch_rows);
    if (nx > cap) nx = cap;
    hipLaunchKernelGGL(k_layernorm, dim3((unsigned)nx, (unsigned)batch_rows), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y,
                       (int)per_b, C, eps, (unsigned*)row_amax);
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}

extern "C" int egr_layernorm_rows(const float* x, const float* gamma, const float* beta, float* y, int64_t rows, int C,
                                  float eps, void* stream) {
    // (as many groups as keep the grid fine: the grouping does not change any value)
    int g = 1;
    for (int c : {64, 32, 16, 8, 4, 2}) if (rows % c == 0) { g = c; break; }
    return egr_layernorm_rows_ra(x, gamma, beta, y, rows, C, eps, g, nullptr, stream);
}

extern "C" int egr_softmax_rows(float* x, int64_t rows, int cols, void* stream) {
    EGR_CHECK(x && rows >= 1 && cols >= 1, EGR_ERR_ARG, "bad argument");
    const dim3 grid((unsigned)rows), blk(256);
    if (cols <= 1024) hipLaunchKernelGGL(k_softmax_reg<4>, grid, blk, 0, (hipStream_t)stream, x, cols);
    else if (cols <= 2048) hipLaunchKernelGGL(k_softmax_reg<8>, grid, blk, 0, (hipStream_t)stream, x, cols);
    else if (cols <= 4096) hipLaunchKernelGGL(k_softmax_reg<16>, grid, blk, 0, (hipStream_t)stream, x, cols);
    else hipLaunchKernelGGL(k_softmax, grid, blk, 0, (hipStream_t)stream, x, cols);
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}

// n elements = batch_rows equal consecutive runs
extern "C" int egr_eltwise_ra(const float* a, const float* b, float* y, int64_t n, int op, float s0, float s1, int batch_rows,
                              float* row_amax, void* stream) {
    EGR_CHECK(a && y && n >= 0 && op >= 0 && op <= 5 && batch_rows >= 1 && batch_rows <= 65535 && n % batch_rows == 0, EGR_ERR_ARG, "bad argument");
    EGR_CHECK(b || (op != EW_ADD && op != EW_AXPBY && op != EW_ADD_SCALE), EGR_ERR_ARG, "binary op needs b");
    if (n == 0) return EGR_OK;
    const long long per = n / batch_rows;
    hipLaunchKernelGGL(k_eltwise, dim3(row_grid_x(per, batch_rows, 8192, 4), (unsigned)batch_rows), dim3(256), 0, (hipStream_t)stream, a, b, y, per,
                       op, s0, s1, (unsigned*)row_amax);
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}

extern "C" int egr_eltwise(const float* a, const float* b, float* y, int64_t n, int op, float s0, float s1, void* stream) {
    int g = 1;
    for (int c : {64, 32, 16, 8, 4, 2}) if (n % c == 0) { g = c; break; }
    return egr_eltwise_ra(a, b, y, n, op, s0, s1, g, nullptr, stream);
}

// rows token rows = batch_rows equal consecutive groups
extern "C" int egr_geglu_ra(const float* u, float* y, int64_t rows, int D, int batch_rows, float* row_amax, void* stream) {
    EGR_CHECK(u && y && rows >= 1 && D >= 1 && batch_rows >= 1 && batch_rows <= 65535 && rows % batch_rows == 0, EGR_ERR_ARG, "bad argument");
    const long long per = rows / batch_rows;
    hipLaunchKernelGGL(k_geglu, dim3(row_grid_x(per * D, batch_rows, 8192, 4), (unsigned)batch_rows), dim3(256), 0, (hipStream_t)stream, u, y, per, D,
                       (unsigned*)row_amax);
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}

extern "C" int egr_geglu(const float* u, float* y, int64_t rows, int D, void* stream) {
    int g = 1;
    for (int c : {64, 32, 16, 8, 4, 2}) if (rows % c == 0) { g = c; break; }
    return egr_geglu_ra(u, y, rows, D, g, nullptr, stream);
}

extern "C" int egr_tap_gather(const float* P, const float* bias, float* y, int B, int H, int W, int KH, int KW, int Cout, int pad_t,
                              int pad_l, void* stream) {
    EGR_CHECK(P && y && B >= 1 && H >= 1 && W >= 1 && KH >= 1 && KW >= 1 && Cout >= 1, EGR_ERR_ARG, "bad argument");
    hipLaunchKernelGGL(k_tap_gather, dim3(grid1d((long long)B * H * W * Cout)), dim3(256), 0, (hipStream_t)stream, P, bias, y, B, H, W,
                       KH, KW, Cout, pad_t, pad_l);
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}

extern "C" int egr_conv_cin1(const float* x, const float* w_packed, const float* bias, float* y, int B, int H, int W, int Cout, int KH,
                             int KW, int pad_t, int pad_l, void* stream) {
    EGR_CHECK(x && w_packed && y && B >= 1 && H >= 1 && W >= 1 && Cout >= 4 && Cout % 4 == 0 && 256 % (Cout / 4) == 0, EGR_ERR_ARG,
              "bad argument (Cout / 4 must divide 256)");
    EGR_CHECK((KH == 3 && KW == 3) || (KH == 1 && KW == 1), EGR_ERR_UNSUPPORTED, "one-input-channel conv: 3x3 or 1x1 only");
    EGR_CHECK((long long)B * H * W < (1ll << 31), EGR_ERR_ARG, "image too large for 32-bit pixel indices");
    EGR_CHECK((((uintptr_t)y) & 15) == 0 && (!bias || (((uintptr_t)bias) & 15) == 0), EGR_ERR_ARG, "y / bias need 16-byte alignment");
    const int ppi = 256 / (Cout / 4);
    long long nb = ((long long)B * H * W + ppi - 1) / ppi;
    if (nb > 8192) nb = 8192;
    if (KH == 3) hipLaunchKernelGGL((k_conv_cin1<9>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, w_packed, bias, y, B, H, W,
                                    Cout, KW, pad_t, pad_l);
    else hipLaunchKernelGGL((k_conv_cin1<1>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, w_packed, bias, y, B, H, W, Cout,
                            KW, pad_t, pad_l);
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}

// M pixel rows = batch_rows equal consecutive groups
extern "C" int egr_concat_channels_ra(const float* a, const float* b, float* y, int64_t M, int C1, int C2, int batch_rows, float* row_amax,
                                      void* stream) {
    EGR_CHECK(a && b && y && M >= 1 && C1 >= 1 && C2 >= 1 && batch_rows >= 1 && batch_rows <= 65535 && M % batch_rows == 0, EGR_ERR_ARG, "bad argument");
    const long long per = M / batch_rows;
    hipLaunchKernelGGL(k_concat, dim3(row_grid_x(per * (C1 + C2), batch_rows, 8192, 4), (unsigned)batch_rows), dim3(256), 0, (hipStream_t)stream, a, b, y,
                       per, C1, C2, (unsigned*)row_amax);
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}

extern "C" int egr_concat_channels(const float* a, const float* b, float* y, int64_t M, int C1, int C2, void* stream) {
    int g = 1;
    for (int c : {64, 32, 16, 8, 4, 2}) if (M % c == 0) { g = c; break; }
    return egr_concat_channels_ra(a, b, y, M, C1, C2, g, nullptr, stream);
}

extern "C" int egr_transpose_batched(const float* x, float* y, int batch, int R, int Cc, void* stream) {
    EGR_CHECK(x && y && batch >= 1 && batch <= 65535 && R >= 1 && Cc >= 1, EGR_ERR_ARG, "bad argument");
    hipLaunchKernelGGL(k_transpose, dim3((Cc + 31) / 32, (R + 31) / 32, batch), dim3(256), 0, (hipStream_t)stream, x, y, R,
                       Cc);
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}

extern "C" int egr_snake_aa_ra(const float* x, const float* alpha, const float* beta, const float* filt, float* y, int B,
                               int L, int C, int K, float* row_amax, void* stream);
extern "C" int egr_snake_aa(const float* x, const float* alpha, const float* beta, const float* filt, float* y, int B,
                            int L, int C, int K, void* stream) {
    return egr_snake_aa_ra(x, alpha, beta, filt, y, B, L, C, K, nullptr, stream);
}

// same, and row_amax[b] (optional; floats the caller zeroed) is raised to max |y| of batch row b
extern "C" int egr_snake_aa_ra(const float* x, const float* alpha, const float* beta, const float* filt, float* y, int B,
                               int L, int C, int K, float* row_amax, void* stream) {
    EGR_CHECK(x && alpha && beta && filt && y && B >= 1 && B <= 65535 && L >= 1 && C >= 1 && K >= 2 && K <= 32 && K % 2 == 0,
              EGR_ERR_ARG, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    static const bool tiled = getenv("EGR_SNAKE") && !strcmp(getenv("EGR_SNAKE"), "tiled");
    if (K == 12 && !tiled) {
        constexpr int J = 16;
        const int nruns = (L + J - 1) / J;
        static const int max_wg = getenv("EGR_SNAKE_WG") ? atoi(getenv("EGR_SNAKE_WG")) : 8192;
        hipLaunchKernelGGL((k_snake_aa_reg<J>), dim3(row_grid_x((long long)nruns * C, B, max_wg), (unsigned)B), dim3(256), 0, st, x, alpha, beta, filt, y, L, C,
                           nruns, (unsigned*)row_amax);
        row_amax = nullptr;                      // done in the kernel
    } else if (K == 12 && C % 64 == 0)
        hipLaunchKernelGGL((k_snake_aa_tiled<64, 32>), dim3(C / 64, (L + 31) / 32, B), dim3(256), 0, st, x, alpha, beta, filt, y, L, C);
    else if (K == 12 && C % 32 == 0)
        hipLaunchKernelGGL((k_snake_aa_tiled<32, 64>), dim3(C / 32, (L + 63) / 64, B), dim3(256), 0, st, x, alpha, beta, filt, y, L, C);
    else if (K == 12 && C % 16 == 0)
        hipLaunchKernelGGL((k_snake_aa_tiled<16, 128>), dim3(C / 16, (L + 127) / 128, B), dim3(256), 0, st, x, alpha, beta, filt, y, L, C);
    else
        hipLaunchKernelGGL(k_snake_aa, dim3((C + 63) / 64, (L + 3) / 4, B), dim3(256), 0, st, x, alpha, beta, filt, y, L, C, K);
    EGR_HIP(hipGetLastError());
    if (row_amax) return egr_absmax_rows(y, B, (int64_t)L * C, 1, 0, row_amax, stream);      // the other kernels: a pass over y
    return EGR_OK;
}

extern "C" int egr_col2im_convtr1d(const float* Y, const float* bias, const float* add, float* out, int B, int Lin,
                                   int Lout, int Co, int K, int r, int pad, void* stream) {
    EGR_CHECK(Y && out && B >= 1 && Lin >= 1 && Lout >= 1 && Co >= 1 && K >= 1 && r >= 1 && pad >= 0, EGR_ERR_ARG,
              "bad argument");
    const long long n = (long long)B * Lout * Co;
    hipLaunchKernelGGL(k_col2im1d, dim3(grid1d(n)), dim3(256), 0, (hipStream_t)stream, Y, bias, add, out, Lin, Lout, Co, K,
                       r, pad, n);
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}

extern "C" int egr_stft_frames(const float* x, int B, int L, int n_fft, int hop, int rpad, int T, int t_valid, int ldm,
                               const float* window, float* mag, void* stream) {
    EGR_CHECK(x && window && mag && B >= 1 && B <= 65535 && L >= 2 && hop >= 1 && T >= 1 && t_valid >= 0 && t_valid <= T,
              EGR_ERR_ARG, "bad argument");
    EGR_CHECK(n_fft >= 4 && n_fft % 2 == 0 && n_fft <= 8192 && ldm >= n_fft / 2 + 1 && rpad < L, EGR_ERR_UNSUPPORTED,
              "bad STFT geometry");
    FrameTables t;
    int rc = frame_tables(n_fft, &t);
    if (rc) return rc;
    const size_t lds = (size_t)2 * (n_fft / 2) * sizeof(float2);
    hipLaunchKernelGGL(k_stft_frames, dim3(T, B), dim3(256), lds, (hipStream_t)stream, x, L, n_fft, hop, rpad, T, t_valid,
                       ldm, window, t.fd, t.tw, t.wsplit, mag);
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}

// row r of out is a function of (seed, id of row r) only: id = row_ids[r] (device array) or, without one, id_base + r
extern "C" int egr_randn_base(float* out, int64_t per_row, int rows, uint64_t seed, const int64_t* row_ids, int64_t id_base, void* stream) {
    EGR_CHECK(out && per_row >= 1 && rows >= 1, EGR_ERR_ARG, "bad argument");
    const long long n = ((per_row + 3) / 4) * rows;
    hipLaunchKernelGGL(k_randn, dim3(grid1d(n)), dim3(256), 0, (hipStream_t)stream, out, (long long)per_row, rows,
                       (unsigned long long)seed, (const long long*)row_ids, (long long)id_base);
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}

extern "C" int egr_randn(float* out, int64_t per_row, int rows, uint64_t seed, const int64_t* row_ids, void* stream) {
    return egr_randn_base(out, per_row, rows, seed, row_ids, 0, stream);
}

extern "C" int egr_lowpass_gain(const float* mag, int B, int T, int ldm, int nb, float pct, float sr, int order,
                                float ripple_db, int64_t nbins, int* cut_out, float* gain, void* stream) {
    EGR_CHECK(mag && cut_out && gain && B >= 1 && B <= 65535 && T >= 1 && nb >= 2 && nb <= ldm && nbins >= 2 && order >= 1,
              EGR_ERR_ARG, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_cutoff_bin, dim3(B), dim3(256), nb * sizeof(float), st, mag, T, ldm, nb, pct, cut_out);
    const float eps2 = powf(10.0f, ripple_db / 10.0f) - 1.0f;
    hipLaunchKernelGGL(k_cheby_gain, dim3(grid1d(nbins), B), dim3(256), 0, st, cut_out, nb, sr, order, eps2, (long long)nbins,
                       gain);
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}

extern "C" int egr_winograd_input(const float* x, const float* gn_scale, const float* gn_shift, int gn_silu, int B, int H,
                                  int W, int C, float* V, void* stream) {
    EGR_CHECK(x && V && B >= 1 && H >= 1 && W >= 1 && C >= 4 && C % 4 == 0 && (!gn_scale || gn_shift), EGR_ERR_ARG,
              "bad argument");
    const int TH = (H + 1) / 2, TW = (W + 1) / 2;
    const long long n = (long long)B * TH * TW * (C / 4);
    hipLaunchKernelGGL(k_wino_in, dim3(grid1d(n)), dim3(256), 0, (hipStream_t)stream, x, B, H, W, C, TH, TW, gn_scale,
                       gn_shift, gn_silu, V);
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}

extern "C" int egr_winograd_output(const float* M, const float* bias, const float* res, float* y, int B, int H, int W, int N,
                                   int act, void* stream) {
    EGR_CHECK(M && y && B >= 1 && H >= 1 && W >= 1 && N >= 4 && N % 4 == 0 && (act == 0 || act == 1), EGR_ERR_ARG,
              "bad argument");
    const int TH = (H + 1) / 2, TW = (W + 1) / 2;
    const long long n = (long long)B * TH * TW * (N / 4);
    hipLaunchKernelGGL(k_wino_out, dim3(grid1d(n)), dim3(256), 0, (hipStream_t)stream, M, bias, res, B, H, W, N, TH, TW, act, y);
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}
