// Winograd F(4x4,3x3) transforms for the stride-1 pad-1 3x3 convolutions of the FlashSR graph (4x fewer multiplies than
// the direct form, 36 z-batched GEMMs of [tiles][Cin] x [Cin][Cout] in between).  Both kernels are HBM-bound: the input
// transform reads the activation once and writes 36/16 = 2.25x its size, the output transform reads 2.25x the output and
// writes it once (F(2x2,3x3): 4x each).
//
// Cook-Toom points 0, +-a, +-b, inf with a = 3/4, b = 3/2 instead of the textbook 0, +-1, +-2: the error of the scheme is
// the GEMMs' fp32 accumulation error amplified by max|M| / max|y|, and the smaller points cut that factor from ~50 to
// ~12 (tools/probe_wino4_error.py; measured end-to-end error vs float64 in tests/test_gpu_flashsr.py).  Every constant
// below (a^2, b^2, a^2 b^2, a b^2, a^2 b, a^3, b^3) is exact in fp32, so B^T, G and A^T agree exactly.
//   B^T row p=0 : M0(x) = (x^2-a^2)(x^2-b^2)      rows +-a : x (x+-a)(x^2-b^2)      rows +-b : x (x+-b)(x^2-a^2)
//   B^T row inf : x (x^2-a^2)(x^2-b^2)            (coefficients of x^0..x^5)
//   G row p     : [1, p, p^2] / N_p,  N_0 = a^2 b^2, N_+-a = 2 a^2 (a^2-b^2), N_+-b = 2 b^2 (b^2-a^2);  G row inf : [0, 0, 1]
//   A^T[i][p]   : p^i (i < 4), plus 1 at [3][inf]
#include "egr_common.h"
#include "egr_rowmax.h"

namespace egr {

#define W4_A 0.75
#define W4_B 1.5
static constexpr float kA = (float)W4_A, kB = (float)W4_B, kA2 = kA * kA, kB2 = kB * kB, kA3 = kA2 * kA, kB3 = kB2 * kB;
static constexpr float kC0 = kA2 * kB2, kC2 = -(kA2 + kB2);

#define F4A(a, b) make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w)
#define F4S(a, b) make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w)
#define F4M(s, a) make_float4((s) * a.x, (s) * a.y, (s) * a.z, (s) * a.w)
#define F4FMA(s, a, b) make_float4(fmaf((s), a.x, b.x), fmaf((s), a.y, b.y), fmaf((s), a.z, b.z), fmaf((s), a.w, b.w))

// V / M are written once and read once by the next kernel: optionally bypass the caches on the way out (EGR_W4_NT)
#ifdef EGR_W4_NT
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_f4(float* p, const float4& v) {
    f4v t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, (f4v*)p);
}
#else
__device__ __forceinline__ void st_f4(float* p, const float4& v) { *(float4*)p = v; }
#endif

// t = B^T d for one 6-vector; rows ordered 0, +a, -a, +b, -b, inf
__device__ __forceinline__ void bt6(const float4 (&d)[6], float4 (&t)[6]) {
    const float4 ea = F4FMA(-kB2, d[2], d[4]);              // even part of the +-a rows: d4 - b^2 d2
    const float4 oa = F4M(kA, F4FMA(-kB2, d[1], d[3]));     // odd part: a (d3 - b^2 d1)
    const float4 eb = F4FMA(-kA2, d[2], d[4]);              // +-b rows: d4 - a^2 d2
    const float4 ob = F4M(kB, F4FMA(-kA2, d[1], d[3]));     //           b (d3 - a^2 d1)
    t[0] = F4FMA(kC0, d[0], F4FMA(kC2, d[2], d[4]));
    t[1] = F4A(ea, oa);
    t[2] = F4S(ea, oa);
    t[3] = F4A(eb, ob);
    t[4] = F4S(eb, ob);
    t[5] = F4FMA(kC0, d[1], F4FMA(kC2, d[3], d[5]));
}

// s = A^T m for one 6-vector
__device__ __forceinline__ void at6(const float4 (&m)[6], float4 (&s)[4]) {
    const float4 pa = F4A(m[1], m[2]), ma = F4S(m[1], m[2]), pb = F4A(m[3], m[4]), mb = F4S(m[3], m[4]);
    s[0] = F4A(F4A(m[0], pa), pb);
    s[1] = F4FMA(kA, ma, F4M(kB, mb));
    s[2] = F4FMA(kA2, pa, F4M(kB2, pb));
    s[3] = F4A(F4FMA(kA3, ma, F4M(kB3, mb)), m[5]);
}

// V[6 i + j][t][c] = (B^T d B)[i][j] of the 6x6 input tile whose origin is (4 ty - 1, 4 tx - 1); t = (b, ty, tx).
// GN / SILU are compile-time so that the 36 tile loads are issued back to back (a run-time branch between them made the
// compiler wait for every load separately: 36 serialized round trips per thread).
// row_amax (optional): row_amax[b] is raised to the bits of max |V| over image b's tiles and all 36 components -- the operand
// scale of the GEMMs that read V (scheme 1 of csrc/egr_nn_gemm_s3.hip) without another pass over V.
template <bool GN, bool SILU>
__global__ __launch_bounds__(256) void k_wino4_in(const float* __restrict__ x, int B, int H, int W, int C, int TH, int TW,
                                                   const float* __restrict__ gsc, const float* __restrict__ gsh,
                                                   float* __restrict__ V, unsigned* __restrict__ row_amax) {
    const int C4 = C >> 2;
    const long long P = (long long)B * TH * TW;
    const size_t zs = (size_t)P * C;
    // grid (row chunks, B): a workgroup loops over the (tile, channel quad) items of image b = blockIdx.y
    const int b = blockIdx.y;
    const long long per_img = (long long)TH * TW * C4;
    float vmax = 0.f;
    for (long long ii = (long long)blockIdx.x * blockDim.x + threadIdx.x; ii < per_img; ii += (long long)gridDim.x * blockDim.x) {
        {
        const int c4 = (int)(ii % C4);
        const int tl = (int)(ii / C4);
        const int tx = tl % TW, ty = tl / TW;
        const long long t = (long long)b * TH * TW + tl;
        float4 d[6][6];                                   // d[r][q]: row r, column q of the tile
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const int iy = 4 * ty - 1 + r;
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const int ix = 4 * tx - 1 + q;
                const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                const size_t off = ok ? (((size_t)b * H + iy) * W + ix) * C + 4 * c4 : (size_t)(4 * c4);
                d[r][q] = *(const float4*)(x + off);
            }
        }
        // interior tiles (the vast majority) need no masking; edge tiles zero their out-of-image taps after the affine
        const bool interior = ty > 0 && tx > 0 && 4 * ty + 4 < H && 4 * tx + 4 < W;
        if (GN) {      // producer's GroupNorm (+SiLU) fused into the transform; padding stays zero
            const float4 sc = *(const float4*)(gsc + (size_t)b * C + 4 * c4);
            const float4 sh = *(const float4*)(gsh + (size_t)b * C + 4 * c4);
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    float4 v = d[r][q];
                    v = make_float4(v.x * sc.x + sh.x, v.y * sc.y + sh.y, v.z * sc.z + sh.z, v.w * sc.w + sh.w);
                    if (SILU) {
                        v.x = v.x / (1.f + __expf(-v.x)); v.y = v.y / (1.f + __expf(-v.y));
                        v.z = v.z / (1.f + __expf(-v.z)); v.w = v.w / (1.f + __expf(-v.w));
                    }
                    d[r][q] = v;
                }
        }
        if (!interior) {
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                const int iy = 4 * ty - 1 + r;
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    const int ix = 4 * tx - 1 + q;
                    if (!((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)) d[r][q] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) {                     // B^T d : transform the columns in place
            float4 col[6], o6[6];
#pragma unroll
            for (int r = 0; r < 6; ++r) col[r] = d[r][q];
            bt6(col, o6);
#pragma unroll
            for (int r = 0; r < 6; ++r) d[r][q] = o6[r];
        }
        float* o = V + (size_t)t * C + 4 * c4;
#pragma unroll
        for (int r = 0; r < 6; ++r) {                     // (.) B == B^T applied along the row
            float4 row[6];
            bt6(d[r], row);
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                st_f4(o + (size_t)(6 * r + q) * zs, row[q]);
                vmax = amax4(row[q], vmax);
            }
        }
        }
    }
    if (row_amax) row_amax_commit(row_amax + (size_t)b * EGR_ROW_AMAX_STRIDE, vmax);                  // once per wave, every lane arrives together
}

// y[b][4ty+a][4tx+c][n] = act((A^T M A)[a][c] + bias[n] + res[...]),  M[6 i + j][t][n].  RES / SILU are compile-time (loads
// of the residual are then hoisted above the arithmetic instead of being waited for one by one).
// STATS: each thread also writes (sum, sum of squares) of its 64 outputs to part[t][n4] -- the GroupNorm statistics of y are
// then a reduction over 1/32 of the tensor's bytes instead of another pass over y (egr_groupnorm_coeff_from_partials).
// row_amax (optional): row_amax[b] raised to max |y| of image b (the tensor feeds 1x1 / strided / phase convolutions directly).
template <bool RES, bool SILU, bool STATS>
__global__ __launch_bounds__(256) void k_wino4_out(const float* __restrict__ Mx, const float* __restrict__ bias,
                                                    const float* __restrict__ res, int B, int H, int W, int N, int TH, int TW,
                                                    float* __restrict__ y, float2* __restrict__ part, unsigned* __restrict__ row_amax) {
    const int N4 = N >> 2;
    const long long P = (long long)B * TH * TW;
    const size_t zs = (size_t)P * N;
    // grid (row chunks, B): a workgroup loops over the (tile, channel quad) items of image b = blockIdx.y
    const int b = blockIdx.y;
    const long long per_img = (long long)TH * TW * N4;
    float ymax = 0.f;
    for (long long ii = (long long)blockIdx.x * blockDim.x + threadIdx.x; ii < per_img; ii += (long long)gridDim.x * blockDim.x) {
        {
        const int n4 = (int)(ii % N4);
        const int tl = (int)(ii / N4);
        const int tx = tl % TW, ty = tl / TW;
        const long long t = (long long)b * TH * TW + tl;
        const long long i = t * N4 + n4;
        const float* mi = Mx + (size_t)t * N + 4 * n4;
        float4 s[4][6];                                   // s[a][q] = (A^T m)[a][q]
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            float4 m[6];
#pragma unroll
            for (int r = 0; r < 6; ++r) m[r] = *(const float4*)(mi + (size_t)(6 * r + q) * zs);
            float4 col[4];
            at6(m, col);
#pragma unroll
            for (int a = 0; a < 4; ++a) s[a][q] = col[a];
        }
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias) bv = *(const float4*)(bias + 4 * n4);
        float ps = 0.f, pq = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const size_t off = (((size_t)b * H + 4 * ty + a) * W + 4 * tx) * N + 4 * n4;
            float4 rr[4];
            if (RES) {
#pragma unroll
                for (int c = 0; c < 4; ++c) rr[c] = *(const float4*)(res + off + (size_t)c * N);
            }
            float4 o[4];
            at6(s[a], o);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float4 v = F4A(o[c], bv);
                if (RES) v = F4A(v, rr[c]);
                if (SILU) {
                    v.x = v.x / (1.f + __expf(-v.x)); v.y = v.y / (1.f + __expf(-v.y));
                    v.z = v.z / (1.f + __expf(-v.z)); v.w = v.w / (1.f + __expf(-v.w));
                }
                *(float4*)(y + off + (size_t)c * N) = v;
                ymax = amax4(v, ymax);
                if (STATS) {
                    ps += (v.x + v.y) + (v.z + v.w);
                    pq += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
                }
            }
        }
        if (STATS) part[i] = make_float2(ps, pq);
        }
    }
    if (row_amax) row_amax_commit(row_amax + (size_t)b * EGR_ROW_AMAX_STRIDE, ymax);
}

// stats[b][g] = (sum, sum of squares) over image b's tiles and group g's channel quads of part[t][n4]; one workgroup per (g, b),
// fixed summation order (deterministic), double accumulation.
__global__ __launch_bounds__(256) void k_gn_from_partials(const float2* __restrict__ part, int tiles_per_image, int N4, int q4,
                                                           double* __restrict__ stats) {
    __shared__ double rs[256], rq[256];
    const int g = blockIdx.x, b = blockIdx.y, G = gridDim.x;
    const float2* pb = part + (size_t)b * tiles_per_image * N4 + (size_t)g * q4;
    double s = 0.0, q = 0.0;
    const long long n = (long long)tiles_per_image * q4;
    for (long long e = threadIdx.x; e < n; e += 256) {
        const long long t = e / q4;
        const int j = (int)(e - t * q4);
        const float2 v = pb[t * N4 + j];
        s += (double)v.x;
        q += (double)v.y;
    }
    rs[threadIdx.x] = s;
    rq[threadIdx.x] = q;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) { rs[threadIdx.x] += rs[threadIdx.x + o]; rq[threadIdx.x] += rq[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        stats[((size_t)b * G + g) * 2 + 0] = rs[0];
        stats[((size_t)b * G + g) * 2 + 1] = rq[0];
    }
}

static inline int grid1d(long long n) {
    long long b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

}  // namespace egr

using namespace egr;

// G (6x3, rows 0, +a, -a, +b, -b, inf) of the scheme above, in double: the host forms U = G g G^T with it
extern "C" int egr_winograd4_g(double* g18) {
    EGR_CHECK(g18, EGR_ERR_ARG, "null g18");
    const double a = W4_A, b = W4_B, n0 = a * a * b * b, na = 2 * a * a * (a * a - b * b), nb = 2 * b * b * (b * b - a * a);
    const double pts[5] = {0.0, a, -a, b, -b}, nn[5] = {n0, na, na, nb, nb};
    for (int j = 0; j < 5; ++j) {
        g18[3 * j + 0] = 1.0 / nn[j];
        g18[3 * j + 1] = pts[j] / nn[j];
        g18[3 * j + 2] = pts[j] * pts[j] / nn[j];
    }
    g18[15] = 0.0; g18[16] = 0.0; g18[17] = 1.0;
    return EGR_OK;
}

extern "C" int egr_winograd4_input_ra(const float* x, const float* gn_scale, const float* gn_shift, int gn_silu, int B, int H,
                                      int W, int C, float* V, float* row_amax, void* stream);
extern "C" int egr_winograd4_input(const float* x, const float* gn_scale, const float* gn_shift, int gn_silu, int B, int H,
                                   int W, int C, float* V, void* stream) {
    return egr_winograd4_input_ra(x, gn_scale, gn_shift, gn_silu, B, H, W, C, V, nullptr, stream);
}

// same, and row_amax[b] (optional; bits of a non-negative float, zeroed by the caller) is raised to max |V| of image b
extern "C" int egr_winograd4_input_ra(const float* x, const float* gn_scale, const float* gn_shift, int gn_silu, int B, int H,
                                      int W, int C, float* V, float* row_amax, void* stream) {
    EGR_CHECK(x && V && B >= 1 && H >= 4 && W >= 4 && H % 4 == 0 && W % 4 == 0 && C >= 4 && C % 4 == 0 &&
                  (!gn_scale || gn_shift), EGR_ERR_ARG, "F(4x4,3x3) input transform needs H, W, C multiples of 4");
    const int TH = H / 4, TW = W / 4;
    EGR_CHECK(B <= 65535, EGR_ERR_ARG, "too many images");
    const dim3 g(row_grid_x((long long)TH * TW * (C / 4), B), (unsigned)B), blk(256);
    hipStream_t st = (hipStream_t)stream;
    unsigned* ra = (unsigned*)row_amax;
    if (!gn_scale) hipLaunchKernelGGL((k_wino4_in<false, false>), g, blk, 0, st, x, B, H, W, C, TH, TW, gn_scale, gn_shift, V, ra);
    else if (gn_silu) hipLaunchKernelGGL((k_wino4_in<true, true>), g, blk, 0, st, x, B, H, W, C, TH, TW, gn_scale, gn_shift, V, ra);
    else hipLaunchKernelGGL((k_wino4_in<true, false>), g, blk, 0, st, x, B, H, W, C, TH, TW, gn_scale, gn_shift, V, ra);
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}

static int wino4_out_launch(const float* M, const float* bias, const float* res, float* y, int B, int H, int W, int N, int act,
                            float2* part, unsigned* row_amax, hipStream_t st) {
    const int TH = H / 4, TW = W / 4;
    EGR_CHECK(B <= 65535, EGR_ERR_ARG, "too many images");
    const dim3 g(row_grid_x((long long)TH * TW * (N / 4), B), (unsigned)B), blk(256);
#define W4O(R_, S_, T_) hipLaunchKernelGGL((k_wino4_out<R_, S_, T_>), g, blk, 0, st, M, bias, res, B, H, W, N, TH, TW, y, part, row_amax)
    if (part) {
        if (res && act) W4O(true, true, true); else if (res) W4O(true, false, true);
        else if (act) W4O(false, true, true); else W4O(false, false, true);
    } else {
        if (res && act) W4O(true, true, false); else if (res) W4O(true, false, false);
        else if (act) W4O(false, true, false); else W4O(false, false, false);
    }
#undef W4O
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}

extern "C" int egr_winograd4_output(const float* M, const float* bias, const float* res, float* y, int B, int H, int W, int N,
                                    int act, void* stream) {
    EGR_CHECK(M && y && B >= 1 && H >= 4 && W >= 4 && H % 4 == 0 && W % 4 == 0 && N >= 4 && N % 4 == 0 &&
                  (act == 0 || act == 1), EGR_ERR_ARG, "F(4x4,3x3) output transform needs H, W, N multiples of 4");
    return wino4_out_launch(M, bias, res, y, B, H, W, N, act, nullptr, nullptr, (hipStream_t)stream);
}

// same, and part[B*(H/4)*(W/4)][N/4] float2 receives every thread's (sum, sum of squares) of its 4 x 4 x 4 outputs
extern "C" int egr_winograd4_output_stats(const float* M, const float* bias, const float* res, float* y, int B, int H, int W,
                                          int N, int act, void* part, void* stream) {
    EGR_CHECK(M && y && part && B >= 1 && H >= 4 && W >= 4 && H % 4 == 0 && W % 4 == 0 && N >= 4 && N % 4 == 0 &&
                  (act == 0 || act == 1), EGR_ERR_ARG, "F(4x4,3x3) output transform needs H, W, N multiples of 4");
    return wino4_out_launch(M, bias, res, y, B, H, W, N, act, (float2*)part, nullptr, (hipStream_t)stream);
}

// both output transforms with row_amax[b] (optional; floats the caller zeroed) raised to max |y| of image b; part may be null
extern "C" int egr_winograd4_output_ra(const float* M, const float* bias, const float* res, float* y, int B, int H, int W, int N,
                                       int act, void* part, float* row_amax, void* stream) {
    EGR_CHECK(M && y && B >= 1 && H >= 4 && W >= 4 && H % 4 == 0 && W % 4 == 0 && N >= 4 && N % 4 == 0 &&
                  (act == 0 || act == 1), EGR_ERR_ARG, "F(4x4,3x3) output transform needs H, W, N multiples of 4");
    return wino4_out_launch(M, bias, res, y, B, H, W, N, act, (float2*)part, (unsigned*)row_amax, (hipStream_t)stream);
}

// stats[B][G][2] (double: sum, sum of squares per image and group) from the partials of egr_winograd4_output_stats;
// requires (C / G) % 4 == 0.  Feed egr_groupnorm_coeff_from_stats.
extern "C" int egr_groupnorm_stats_from_partials(const void* part, int B, int tiles_per_image, int C, int G, double* stats,
                                                 void* stream) {
    EGR_CHECK(part && stats && B >= 1 && B <= 65535 && tiles_per_image >= 1 && C >= 4 && G >= 1 && C % G == 0 && (C / G) % 4 == 0,
              EGR_ERR_ARG, "bad argument");
    hipLaunchKernelGGL(k_gn_from_partials, dim3(G, B), dim3(256), 0, (hipStream_t)stream, (const float2*)part, tiles_per_image, C / 4,
                       (C / G) / 4, stats);
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}
