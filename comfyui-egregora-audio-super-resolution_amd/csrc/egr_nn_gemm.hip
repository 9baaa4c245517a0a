// Dense contractions of the FlashSR graph on the gfx950 matrix cores, exact fp32:
//   k_conv_igemm : implicit-GEMM convolution over channels-last activations (2-D 3x3 / 1x1 / strided /
//                  nearest-2x-upsampled input, 1-D dilated), fused bias / per-row channel bias / residual /
//                  activation epilogue.  Also serves every Linear layer (1x1, H=W=1).
//   k_bgemm      : strided batched GEMM (QK^T with B transposed, PV) for the attention blocks.
// Both use v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate: bitwise an fmaf chain, MI355X_MICROARCH.md
// "Matrix cores"), a 128 x BN x 16 block tile staged through double-buffered row-major LDS tiles so an MFMA
// operand fetch is two ds_read_b128 per 32x16 sub-tile, 4 waves per workgroup each owning TM x TN 32x32 accumulators.
// At 64 cycles per MFMA the pipe, not LDS or HBM, is the bound for Cin*Cout >= 128*128; the roofline for these
// kernels is the 157.3 TFLOP/s f32 matrix peak.
#include <string.h>

#include <chrono>
#include <map>
#include <mutex>

#include "egr_conv.h"

namespace egr {

#define BM 128
#define BK 16
#define LROW (BK + 4)   // LDS row pitch in floats: 80 B keeps ds_read_b128 of 32 consecutive rows conflict-free

// One BK=16 slab is consumed as two half-slabs of 4 MFMA k-steps.  LDS tiles are row-major [row][k] (k contiguous):
// lane (li = lane&31, lk = lane>>5) fetches, for half hs, the 4 k-values k = 8*lk + 4*hs .. +3 of its row with ONE
// ds_read_b128; MFMA step j then multiplies k = 4*hs + j (lanes 0-31) and k = 8 + 4*hs + j (lanes 32-63).  The k order
// inside a slab is irrelevant to the sum as long as A and B agree.
template <int TM, int TN>
struct HalfOps {
    float4 a[TM], b[TN];
};

template <int TM, int TN>
__device__ __forceinline__ void read_half(const float* __restrict__ As, const float* __restrict__ Bs, int wm0, int wn0, int hs,
                                          HalfOps<TM, TN>& o) {
    const int lane = threadIdx.x & 63, li = lane & 31, lk = lane >> 5;
#pragma unroll
    for (int i = 0; i < TM; ++i) o.a[i] = *(const float4*)(As + (wm0 + i * 32 + li) * LROW + lk * 8 + hs * 4);
#pragma unroll
    for (int j = 0; j < TN; ++j) o.b[j] = *(const float4*)(Bs + (wn0 + j * 32 + li) * LROW + lk * 8 + hs * 4);
}

template <int TM, int TN>
__device__ __forceinline__ void mma_half(const HalfOps<TM, TN>& o, f32x16 (&acc)[TM][TN]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const float av = ((const float*)&o.a[i])[s];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const float bv = ((const float*)&o.b[j])[s];
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
            }
        }
    }
}

// whole slab in one go (used by k_bgemm)
template <int TM, int TN>
__device__ __forceinline__ void mma_slab(const float* __restrict__ As, const float* __restrict__ Bs, int wm0, int wn0,
                                         f32x16 (&acc)[TM][TN]) {
    HalfOps<TM, TN> h0, h1;
    read_half<TM, TN>(As, Bs, wm0, wn0, 0, h0);
    read_half<TM, TN>(As, Bs, wm0, wn0, 1, h1);
    mma_half<TM, TN>(h0, acc);
    mma_half<TM, TN>(h1, acc);
}

// BN = 128: waves 2x2, each 64x64 (TM=2,TN=2); BN = 64: waves 2x2, each 64x32 (2,1); BN = 32: waves 4x1, each 32x32.
template <int BN> struct TileCfg;
template <> struct TileCfg<128> { static constexpr int WM = 2, WN = 2, TM = 2, TN = 2; };
template <> struct TileCfg<64> { static constexpr int WM = 2, WN = 2, TM = 2, TN = 1; };
template <> struct TileCfg<32> { static constexpr int WM = 4, WN = 1, TM = 1, TN = 1; };

template <int TM, int TN>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[TM][TN]) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
}

// Weights arrive PRE-PACKED as [ceil(K/16)][Cout][16] (slab-major, k contiguous per output channel): a B tile is
// then read exactly like an A tile (one float4 along k per thread) and needs no transposition in LDS.
template <int BN, bool VEC, bool GN = false>
__global__ __launch_bounds__(256) void k_conv_igemm(ConvP p) {
    typedef TileCfg<BN> TC;
    __shared__ __attribute__((aligned(16))) float As[2][BM * LROW];
    __shared__ __attribute__((aligned(16))) float Bs[2][BN * LROW];
    const int tid = threadIdx.x, wave = tid >> 6;
    const int wm0 = (wave / TC::WN) * (BM / TC::WM), wn0 = (wave % TC::WN) * (BN / TC::WN);
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int LH = p.up2 ? 2 * p.H : p.H, LW = p.up2 ? 2 * p.W : p.W;   // logical input extent
    if (p.ksplit <= 1 && gridDim.z > 1) {
        p.x += (size_t)blockIdx.z * p.zx;
        p.w += (size_t)blockIdx.z * p.zw;
        p.y += (size_t)blockIdx.z * p.zy;
    }

    f32x16 acc[TC::TM][TC::TN];
    zero_acc<TC::TM, TC::TN>(acc);

    const int ktiles_all = (p.K + BK - 1) / BK;
    const int kt_begin = p.ksplit > 1 ? blockIdx.z * p.kt_per : 0;
    const int kt_end = p.ksplit > 1 ? min(ktiles_all, kt_begin + p.kt_per) : ktiles_all;
    const int kq = (tid & 3) * 4;             // this thread's k-quad inside a slab
    const int r0 = tid >> 2;                  // tile rows r0 and r0 + 64

    // ---- A rows owned by this thread (VEC path): decode (b, oy, ox) once ----
    int a_iy0[2], a_ix0[2], a_b[2];
    size_t a_base[2];
    bool a_ok[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int m = m0 + r0 + 64 * h;
        a_ok[h] = m < p.M;
        const int mm = a_ok[h] ? m : 0;
        const int ox = mm % p.OW, t = mm / p.OW, oy = t % p.OH, b = t / p.OH;
        a_iy0[h] = oy * p.stride - p.pad_t;
        a_ix0[h] = ox * p.stride - p.pad_l;
        a_base[h] = (size_t)b * p.H * p.W;
        a_b[h] = b;
    }
    // ---- B rows (output channels) owned by this thread ----
    constexpr int BQ = BN * 4;                // float4 slots in a B tile
    bool b_ok[2];
    size_t b_off[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int slot = tid + 256 * h;
        const int n = n0 + (slot >> 2);
        b_ok[h] = slot < BQ && n < p.Cout;
        b_off[h] = (size_t)(b_ok[h] ? n : 0) * BK + kq;
    }
    const size_t b_slab = (size_t)p.Cout * BK;

    float4 ra0, ra1, rb0, rb1;                // staged tile registers (scalars: arrays here ended up in scratch)
    ra0 = ra1 = rb0 = rb1 = make_float4(0.f, 0.f, 0.f, 0.f);
    float rs[8];
    int tap = 0, c0 = 0;                      // VEC: current (tap, first channel) of the slab being loaded
    if (VEC && kt_begin > 0) {
        const int tpt = p.Cin / BK;
        tap = kt_begin / tpt;
        c0 = (kt_begin - tap * tpt) * BK;
    }
    // VEC loader state (plain scalars on purpose: arrays captured by the lambdas below end up in scratch): per owned
    // row a pointer to channel kq of the current tap's pixel, or into the zero page when the tap falls outside the
    // image / the row is past M.  Recomputed only when the tap changes; a steady-state slab is one 16-byte load per row.
    const float* aptr0 = p.zeros;
    const float* aptr1 = p.zeros;
    int astep0 = 0, astep1 = 0;               // 1 when the row is real (advance by c0), 0 when it reads the zero page
#define EGR_SET_TAP(TP)                                                                                              \
    {                                                                                                                \
        const int ky_ = (TP) / p.KW, kx_ = (TP) - ky_ * p.KW;                                                        \
        {                                                                                                            \
            const int iy = a_iy0[0] + ky_, ix = a_ix0[0] + kx_ * p.dil;                                              \
            const bool ok = a_ok[0] && (unsigned)iy < (unsigned)LH && (unsigned)ix < (unsigned)LW;                   \
            const int py = p.up2 ? iy >> 1 : iy, px = p.up2 ? ix >> 1 : ix;                                          \
            aptr0 = ok ? p.x + (a_base[0] + (size_t)py * p.W + px) * p.Cin + kq : p.zeros + kq;                      \
            astep0 = ok ? 1 : 0;                                                                                     \
        }                                                                                                            \
        {                                                                                                            \
            const int iy = a_iy0[1] + ky_, ix = a_ix0[1] + kx_ * p.dil;                                              \
            const bool ok = a_ok[1] && (unsigned)iy < (unsigned)LH && (unsigned)ix < (unsigned)LW;                   \
            const int py = p.up2 ? iy >> 1 : iy, px = p.up2 ? ix >> 1 : ix;                                          \
            aptr1 = ok ? p.x + (a_base[1] + (size_t)py * p.W + px) * p.Cin + kq : p.zeros + kq;                      \
            astep1 = ok ? 1 : 0;                                                                                     \
        }                                                                                                            \
    }
    if (VEC) EGR_SET_TAP(tap)
    const float* bptr0 = b_ok[0] ? p.w + (size_t)kt_begin * b_slab + b_off[0] : p.zeros + kq;
    const float* bptr1 = b_ok[1] ? p.w + (size_t)kt_begin * b_slab + b_off[1] : p.zeros + kq;
    const size_t bstep0 = b_ok[0] ? b_slab : 0, bstep1 = b_ok[1] ? b_slab : 0;

    auto load_tile = [&](int kt) {
        if (VEC) {
            float4 v0 = *(const float4*)(aptr0 + astep0 * c0);
            float4 v1 = *(const float4*)(aptr1 + astep1 * c0);
            if (GN) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    float4& v = h ? v1 : v0;
                    const size_t go = (size_t)a_b[h] * p.Cin + c0 + kq;
                    const float4 sc = *(const float4*)(p.gn_scale + go), sh = *(const float4*)(p.gn_shift + go);
                    v = make_float4(v.x * sc.x + sh.x, v.y * sc.y + sh.y, v.z * sc.z + sh.z, v.w * sc.w + sh.w);
                    if (p.gn_silu) {
                        v.x = v.x / (1.f + __expf(-v.x)); v.y = v.y / (1.f + __expf(-v.y));
                        v.z = v.z / (1.f + __expf(-v.z)); v.w = v.w / (1.f + __expf(-v.w));
                    }
                    if (!(h ? astep1 : astep0)) v = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            ra0 = v0;
            ra1 = v1;
            c0 += BK;
            if (c0 >= p.Cin) { c0 = 0; ++tap; EGR_SET_TAP(tap) }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int e = tid + 256 * i, ml = e & (BM - 1), kk = e >> 7;
                const int kg = kt * BK + kk, m = m0 + ml;
                float v = 0.f;
                if (kg < p.K && m < p.M) {
                    const int tp = kg / p.Cin, ci = kg - tp * p.Cin;
                    const int ky = tp / p.KW, kx = tp - ky * p.KW;
                    const int ox = m % p.OW, t = m / p.OW, oy = t % p.OH, b = t / p.OH;
                    const int iy = oy * p.stride + ky - p.pad_t, ix = ox * p.stride + kx * p.dil - p.pad_l;
                    if (iy >= 0 && iy < LH && ix >= 0 && ix < LW) {
                        const int py = p.up2 ? iy >> 1 : iy, px = p.up2 ? ix >> 1 : ix;
                        v = p.x[(((size_t)b * p.H + py) * p.W + px) * p.Cin + ci];
                    }
                }
                rs[i] = v;
            }
        }
        rb0 = *(const float4*)bptr0;
        bptr0 += bstep0;
        if (256 < BQ) {
            rb1 = *(const float4*)bptr1;
            bptr1 += bstep1;
        }
    };
    auto store_tile = [&](int buf) {
        if (VEC) {
            *(float4*)&As[buf][r0 * LROW + kq] = ra0;
            *(float4*)&As[buf][(r0 + 64) * LROW + kq] = ra1;
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int e = tid + 256 * i;
                As[buf][(e & (BM - 1)) * LROW + (e >> 7)] = rs[i];
            }
        }
        if (tid < BQ) *(float4*)&Bs[buf][(tid >> 2) * LROW + kq] = rb0;
        if (tid + 256 < BQ) *(float4*)&Bs[buf][((tid + 256) >> 2) * LROW + kq] = rb1;
    };

    // Software pipeline (one barrier per slab, placed MID-slab):
    //   start of iteration kt : tile kt+1 (global loads issued one iteration ago) -> LDS buf[cur^1]; issue loads of tile kt+2
    //   first half            : MFMAs on the operands fetched during the previous iteration || LDS reads of the second half
    //   barrier               : (RAW) tile kt+1 visible; (WAR) nobody reads buf[cur] first halves / buf[cur^1] any more
    //   second half           : MFMAs || LDS reads of slab kt+1's first half from buf[cur^1]
    // so the matrix pipe never waits on an LDS fetch issued after a barrier.
    HalfOps<TC::TM, TC::TN> opA, opB;
    load_tile(kt_begin);
    store_tile(0);
    __syncthreads();
    read_half<TC::TM, TC::TN>(As[0], Bs[0], wm0, wn0, 0, opA);
    if (kt_begin + 1 < kt_end) load_tile(kt_begin + 1);
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int cur = (kt - kt_begin) & 1;
        if (kt + 1 < kt_end) store_tile(cur ^ 1);
        if (kt + 2 < kt_end) load_tile(kt + 2);
        read_half<TC::TM, TC::TN>(As[cur], Bs[cur], wm0, wn0, 1, opB);
        mma_half<TC::TM, TC::TN>(opA, acc);
        __syncthreads();
        if (kt + 1 < kt_end) read_half<TC::TM, TC::TN>(As[cur ^ 1], Bs[cur ^ 1], wm0, wn0, 0, opA);
        mma_half<TC::TM, TC::TN>(opB, acc);
    }

    // ---- epilogue ----
    const int lane = tid & 63, col = lane & 31, rhalf = lane >> 5;
    if (p.ksplit > 1) {        // raw partial sums; bias / residual / activation are applied by k_splitk_reduce
        float* wz = p.ws + (size_t)blockIdx.z * p.M * p.Cout;
#pragma unroll
        for (int i = 0; i < TC::TM; ++i)
#pragma unroll
            for (int j = 0; j < TC::TN; ++j) {
                const int n = n0 + wn0 + j * 32 + col;
                if (n >= p.Cout) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * rhalf;
                    if (m < p.M) wz[(size_t)m * p.Cout + n] = acc[i][j][r];
                }
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < TC::TM; ++i)
#pragma unroll
        for (int j = 0; j < TC::TN; ++j) {
            const int n = n0 + wn0 + j * 32 + col;
            if (n >= p.Cout) continue;
            const float bv = p.bias ? p.bias[n] : 0.f;
            const bool ident = (p.osy == 1 && p.osx == 1 && p.OHF == p.OH && p.OWF == p.OW);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * rhalf;
                if (m >= p.M) continue;
                size_t mo = (size_t)m;
                if (!ident) {
                    const int ox = m % p.OW, t = m / p.OW, oy = t % p.OH, b = t / p.OH;
                    mo = ((size_t)b * p.OHF + (size_t)oy * p.osy + p.ooy) * p.OWF + (size_t)ox * p.osx + p.oox;
                }
                float v = acc[i][j][r] + bv;
                if (p.bias_b) v += p.bias_b[(size_t)(m / (p.OH * p.OW)) * p.Cout + n];
                if (p.res) v += p.res[mo * p.Cout + n];
                p.y[mo * p.Cout + n] = apply_act(v, p.act, p.act_param);
            }
        }
}

// y[place(m)][n] = act(sum_z ws[z][m][n] + bias[n] + bias_b[b][n] + res[place(m)][n])   (fixed summation order)
// AMX: p.out_amax[m / rows_div] is raised to max |y| per batch row (LDS table of up to 1024 batch rows per workgroup, then one
// checked atomic per batch row the workgroup touched)
template <bool AMX>
__global__ __launch_bounds__(256) void k_splitk_reduce(ConvP p) {
    __shared__ unsigned om[AMX ? 1024 : 1];
    const int nbr = AMX ? (p.M + p.rows_div - 1) / p.rows_div : 0;
    if (AMX) {
        for (int t = threadIdx.x; t < nbr; t += 256) om[t] = 0u;
        __syncthreads();
    }
    const long long total = (long long)p.M * p.Cout;
    const bool ident = (p.osy == 1 && p.osx == 1 && p.OHF == p.OH && p.OWF == p.OW);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(i % p.Cout);
        const int m = (int)(i / p.Cout);
        float v = 0.f;
        for (int z = 0; z < p.ksplit; ++z) v += p.ws[(size_t)z * total + i];
        if (p.bias) v += p.bias[n];
        size_t mo = (size_t)m;
        const int ox = m % p.OW, t = m / p.OW, oy = t % p.OH, b = t / p.OH;
        if (!ident) mo = ((size_t)b * p.OHF + (size_t)oy * p.osy + p.ooy) * p.OWF + (size_t)ox * p.osx + p.oox;
        if (p.bias_b) v += p.bias_b[(size_t)b * p.Cout + n];
        if (p.res) v += p.res[mo * p.Cout + n];
        v = apply_act(v, p.act, p.act_param);
        p.y[mo * p.Cout + n] = v;
        if (AMX) atomicMax(&om[m / p.rows_div], __float_as_uint(fabsf(v)));
    }
    if (AMX) {
        __syncthreads();
        for (int t = threadIdx.x; t < nbr; t += 256) {
            const unsigned bits = om[t];
            unsigned* slot = p.out_amax + (size_t)t * EGR_ROW_AMAX_STRIDE;
            if (bits > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, bits);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// strided batched GEMM: C[b] = alpha * A[b] (MxK, row-major, lda) * op(B[b]); op = B (KxN, ldb) or B^T (B is NxK)
// batch index b = (b1, b2) with separate strides so heads can live inside a [B][T][C] tensor.
// ------------------------------------------------------------------------------------------------
struct GemmP {
    const float* a; const float* b; float* c;
    int M, N, K, lda, ldb, ldc, nb2, transB;
    long long sa1, sa2, sb1, sb2, sc1, sc2;
    float alpha;
};

template <int BN>
__global__ __launch_bounds__(256) void k_bgemm(GemmP p) {
    typedef TileCfg<BN> TC;
    __shared__ __attribute__((aligned(16))) float As[2][BM * LROW];
    __shared__ __attribute__((aligned(16))) float Bs[2][BN * LROW];
    const int tid = threadIdx.x, wave = tid >> 6;
    const int wm0 = (wave / TC::WN) * (BM / TC::WM), wn0 = (wave % TC::WN) * (BN / TC::WN);
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int b1 = blockIdx.z / p.nb2, b2 = blockIdx.z - b1 * p.nb2;
    const float* A = p.a + b1 * p.sa1 + b2 * p.sa2;
    const float* Bm = p.b + b1 * p.sb1 + b2 * p.sb2;
    float* Cm = p.c + b1 * p.sc1 + b2 * p.sc2;

    f32x16 acc[TC::TM][TC::TN];
    zero_acc<TC::TM, TC::TN>(acc);

    const int ktiles = (p.K + BK - 1) / BK;
    float ra[8], rbq[8];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {                    // A tile: 128 rows x 16 k ; k fastest across lanes
            const int e = tid + 256 * i, kk = e & (BK - 1), ml = e >> 4;
            const int kg = kt * BK + kk, m = m0 + ml;
            const bool ok = kg < p.K && m < p.M;
            const float v = A[ok ? (size_t)m * p.lda + kg : 0];
            ra[i] = ok ? v : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = tid + 256 * i;
            float v = 0.f;
            if (e < BK * BN) {
                if (p.transB) {                          // B^T stored [N][K]: k fastest
                    const int kk = e & (BK - 1), nl = e >> 4;
                    const int kg = kt * BK + kk, n = n0 + nl;
                    const bool ok = kg < p.K && n < p.N;
                    v = Bm[ok ? (size_t)n * p.ldb + kg : 0];
                    v = ok ? v : 0.f;
                } else {                                  // B stored [K][N]: n fastest
                    const int nl = e % BN, kk = e / BN;
                    const int kg = kt * BK + kk, n = n0 + nl;
                    const bool ok = kg < p.K && n < p.N;
                    v = Bm[ok ? (size_t)kg * p.ldb + n : 0];
                    v = ok ? v : 0.f;
                }
            }
            rbq[i] = v;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = tid + 256 * i;
            As[buf][(e >> 4) * LROW + (e & (BK - 1))] = ra[i];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = tid + 256 * i;
            if (e < BK * BN) {
                if (p.transB) Bs[buf][(e >> 4) * LROW + (e & (BK - 1))] = rbq[i];
                else Bs[buf][(e % BN) * LROW + e / BN] = rbq[i];
            }
        }
    };
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < ktiles; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < ktiles) load_tile(kt + 1);
        mma_slab<TC::TM, TC::TN>(As[cur], Bs[cur], wm0, wn0, acc);
        if (kt + 1 < ktiles) store_tile(cur ^ 1);
        __syncthreads();
    }
    const int lane = tid & 63, col = lane & 31, rhalf = lane >> 5;
#pragma unroll
    for (int i = 0; i < TC::TM; ++i)
#pragma unroll
        for (int j = 0; j < TC::TN; ++j) {
            const int n = n0 + wn0 + j * 32 + col;
            if (n >= p.N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * rhalf;
                if (m < p.M) Cm[(size_t)m * p.ldc + n] = p.alpha * acc[i][j][r];
            }
        }
}

// per-device page of zeros the loaders read for padded / out-of-range rows (shared with egr_nn_gemm_s3.hip)
int zero_page(const float** out) {
    static std::mutex mu;
    static std::map<int, float*> pages;
    int dev = 0;
    EGR_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    auto it = pages.find(dev);
    if (it == pages.end()) {
        float* ptr = nullptr;
        EGR_HIP(hipMalloc((void**)&ptr, 4096));
        EGR_HIP(hipMemset(ptr, 0, 4096));
        it = pages.emplace(dev, ptr).first;
    }
    *out = it->second;
    return EGR_OK;
}

// scratch for split-K partials, one buffer per (device, stream): launches on one stream are ordered, so reuse within a stream
// is safe; row groups of one forward that run on different streams (flashsr_engine.infer_rows) must not share it
static int splitk_workspace(size_t bytes, hipStream_t st, float** out) {
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, std::pair<float*, size_t>> bufs;
    int dev = 0;
    EGR_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    auto& b = bufs[std::make_pair(dev, st)];
    if (b.second < bytes) {
        if (b.first) { EGR_HIP(hipDeviceSynchronize()); EGR_HIP(hipFree(b.first)); }
        size_t want = bytes < (64u << 20) ? (64u << 20) : bytes;
        EGR_HIP(hipMalloc((void**)&b.first, want));
        b.second = want;
    }
    *out = b.first;
    return EGR_OK;
}

// busy-waits `ticks` of the 100 MHz wall clock (bounded by an iteration cap); egr_streams_overlap_us
__global__ void k_spin(long long ticks, unsigned* sink) {
    const long long t0 = wall_clock64();
    unsigned n = 0;
    while (wall_clock64() - t0 < ticks && n < (1u << 24)) ++n;
    if (sink && n == 0xffffffffu) *sink = n;
}

}  // namespace egr

using namespace egr;

extern "C" int egr_conv_nhwc_placed(const float* x, const float* w, const float* bias, const float* bias_b,
                                    const float* res, float* y, int B, int H, int W, int Cin, int OH, int OW, int Cout,
                                    int KH, int KW, int stride, int dil, int pad_t, int pad_l, int up2, int act,
                                    float act_param, int osy, int osx, int ooy, int oox, int OHF, int OWF, void* stream);
static int conv_launch(const float* x, const float* w, const float* bias, const float* bias_b, const float* res, float* y,
                       int B, int H, int W, int Cin, int OH, int OW, int Cout, int KH, int KW, int stride, int dil, int pad_t,
                       int pad_l, int up2, int act, float act_param, int osy, int osx, int ooy, int oox, int OHF, int OWF,
                       int nz, long long zx, long long zw, long long zy, const float* gn_scale, const float* gn_shift,
                       int gn_silu, void* stream, const void* w3 = nullptr, int sch = 0, float w_scale = 1.f,
                       const float* row_amax = nullptr, int batch_rows = 0, float* out_amax = nullptr, void* gn_part = nullptr);

extern "C" int egr_conv_nhwc(const float* x, const float* w, const float* bias, const float* bias_b, const float* res,
                             float* y, int B, int H, int W, int Cin, int OH, int OW, int Cout, int KH, int KW,
                             int stride, int dil, int pad_t, int pad_l, int up2, int act, float act_param,
                             void* stream) {
    return egr_conv_nhwc_placed(x, w, bias, bias_b, res, y, B, H, W, Cin, OH, OW, Cout, KH, KW, stride, dil, pad_t, pad_l,
                                up2, act, act_param, 1, 1, 0, 0, OH, OW, stream);
}

extern "C" int egr_conv_nhwc_placed(const float* x, const float* w, const float* bias, const float* bias_b,
                                    const float* res, float* y, int B, int H, int W, int Cin, int OH, int OW, int Cout,
                                    int KH, int KW, int stride, int dil, int pad_t, int pad_l, int up2, int act,
                                    float act_param, int osy, int osx, int ooy, int oox, int OHF, int OWF, void* stream) {
    return conv_launch(x, w, bias, bias_b, res, y, B, H, W, Cin, OH, OW, Cout, KH, KW, stride, dil, pad_t, pad_l, up2, act,
                       act_param, osy, osx, ooy, oox, OHF, OWF, 1, 0, 0, 0, nullptr, nullptr, 0, stream);
}

// egr_conv_nhwc with the producer's GroupNorm (+SiLU) applied to the input while it is loaded.
extern "C" int egr_conv_nhwc_gn(const float* x, const float* gn_scale, const float* gn_shift, int gn_silu, const float* w,
                                const float* bias, const float* res, float* y, int B, int H, int W, int Cin, int OH, int OW,
                                int Cout, int KH, int KW, int stride, int pad_t, int pad_l, int act, void* stream) {
    EGR_CHECK(gn_scale && gn_shift, EGR_ERR_ARG, "null scale/shift");
    return conv_launch(x, w, bias, nullptr, res, y, B, H, W, Cin, OH, OW, Cout, KH, KW, stride, 1, pad_t, pad_l, 0, act, 0.f, 1,
                       1, 0, 0, OH, OW, 1, 0, 0, 0, gn_scale, gn_shift, gn_silu, stream);
}

// nz independent GEMMs y[z] = x[z] * w[z] ([rows][Cin] x [Cin][Cout], weights packed per z), offsets in elements.
extern "C" int egr_gemm_zbatched(const float* x, const float* w, float* y, int nz, int rows, int Cin, int Cout, int64_t zx,
                                 int64_t zw, int64_t zy, void* stream) {
    EGR_CHECK(nz >= 1 && nz <= 65535, EGR_ERR_ARG, "bad nz");
    return conv_launch(x, w, nullptr, nullptr, nullptr, y, rows, 1, 1, Cin, 1, 1, Cout, 1, 1, 1, 1, 0, 0, 0, 0, 0.f, 1, 1, 0, 0,
                       1, 1, nz, zx, zw, zy, nullptr, nullptr, 0, stream);
}

// Same contraction on the bf16 matrix pipe: w3 = egr_split3_pack(w) (csrc/egr_nn_gemm_s3.hip).  nz > 1: independent
// problems along z with element offsets zx / zy and a weight offset of zw3 16-byte units.
extern "C" int egr_conv_s3(const float* x, const void* w3, const float* bias, const float* bias_b, const float* res, float* y,
                           int B, int H, int W, int Cin, int OH, int OW, int Cout, int KH, int KW, int stride, int dil,
                           int pad_t, int pad_l, int up2, int act, float act_param, int osy, int osx, int ooy, int oox,
                           int OHF, int OWF, int nz, int64_t zx, int64_t zw3, int64_t zy, void* stream) {
    EGR_CHECK(w3, EGR_ERR_ARG, "null w3");
    EGR_CHECK(nz >= 1 && nz <= 65535, EGR_ERR_ARG, "bad nz");
    return conv_launch(x, nullptr, bias, bias_b, res, y, B, H, W, Cin, OH, OW, Cout, KH, KW, stride, dil, pad_t, pad_l, up2, act,
                       act_param, osy, osx, ooy, oox, OHF, OWF, nz, zx, zw3, zy, nullptr, nullptr, 0, stream, w3);
}

// The same on two fp16 terms per operand (csrc/egr_nn_gemm_s3.hip, scheme 1): w2 = egr_split2h_pack(w, w_scale).  The GEMM rows
// (B * OH * OW of them) belong to `batch_rows` equal consecutive groups -- the rows of the batch -- and every group is scaled by its
// own power of two, derived in the kernel from row_amax[group] (bits of the group's max |x|: egr_absmax_rows, or the producer of
// x); nothing about the scale travels through the host.  out_amax (optional, [batch_rows] floats the caller zeroed): raised to
// max |y| per batch row by the epilogue (not when the launch takes the split-K path: the caller checks with egr_conv_h2_splits_k).
extern "C" int egr_conv_h2(const float* x, const void* w2, const float* bias, const float* bias_b, const float* res, float* y,
                           int B, int H, int W, int Cin, int OH, int OW, int Cout, int KH, int KW, int stride, int dil,
                           int pad_t, int pad_l, int up2, int act, float act_param, int osy, int osx, int ooy, int oox,
                           int OHF, int OWF, int nz, int64_t zx, int64_t zw2, int64_t zy, float w_scale, const float* row_amax,
                           int batch_rows, float* out_amax, void* stream) {
    EGR_CHECK(w2 && row_amax, EGR_ERR_ARG, "null w2 / row_amax");
    EGR_CHECK(nz >= 1 && nz <= 65535, EGR_ERR_ARG, "bad nz");
    return conv_launch(x, nullptr, bias, bias_b, res, y, B, H, W, Cin, OH, OW, Cout, KH, KW, stride, dil, pad_t, pad_l, up2, act,
                       act_param, osy, osx, ooy, oox, OHF, OWF, nz, zx, zw2, zy, nullptr, nullptr, 0, stream, w2, 1, w_scale, row_amax,
                       batch_rows, out_amax);
}

// 3x3 stride-1 pad-1 convolution on two fp16 terms with the producer's GroupNorm (+ SiLU) applied to x while it is loaded:
// x' = x * gn_scale[b][c] + gn_shift[b][c] (then SiLU), zero padding after it.  Only the input-stationary kernel (k_conv3x3_is)
// serves it: H %% 4 == 0, W %% 32 == 0, Cin %% 32 == 0, >= 512 tiles of 4 x 32 pixels, otherwise EGR_ERR_UNSUPPORTED (nothing is
// launched).  row_amax[b] must bound max |x'| of image b from above (egr_gn_operand_bound); it need not be tight.
// gn_part (optional, [B * H * W / 32][Cout / 4] float2): GroupNorm partial statistics of y, (sum, sum of squares) per 32-pixel row
// segment and channel quad -- egr_groupnorm_stats_from_partials(part, B, H * W / 32, Cout, G) then gives the statistics of y
// without a pass over it ((Cout / G) % 4 == 0).
extern "C" int egr_conv_h2_gn(const float* x, const float* gn_scale, const float* gn_shift, int gn_silu, const void* w2, const float* bias,
                              const float* res, float* y, int B, int H, int W, int Cin, int Cout, int act, float w_scale, const float* row_amax,
                              float* out_amax, void* gn_part, void* stream) {
    EGR_CHECK(w2 && row_amax && gn_scale && gn_shift, EGR_ERR_ARG, "null w2 / row_amax / gn_scale / gn_shift");
    return conv_launch(x, nullptr, bias, nullptr, res, y, B, H, W, Cin, H, W, Cout, 3, 3, 1, 1, 1, 1, 0, act, 0.f, 1, 1, 0, 0, H, W, 1, 0, 0, 0,
                       gn_scale, gn_shift, gn_silu, stream, w2, 1, w_scale, row_amax, B, out_amax, gn_part);
}

static int conv_launch(const float* x, const float* w, const float* bias, const float* bias_b, const float* res, float* y,
                       int B, int H, int W, int Cin, int OH, int OW, int Cout, int KH, int KW, int stride, int dil, int pad_t,
                       int pad_l, int up2, int act, float act_param, int osy, int osx, int ooy, int oox, int OHF, int OWF,
                       int nz, long long zx, long long zw, long long zy, const float* gn_scale, const float* gn_shift,
                       int gn_silu, void* stream, const void* w3, int sch, float w_scale, const float* row_amax, int batch_rows,
                       float* out_amax, void* gn_part) {
    EGR_CHECK(x && (w || w3) && y, EGR_ERR_ARG, "null x/w/y");
    EGR_CHECK(sch == 0 || (w3 && w_scale > 0.f && row_amax && batch_rows >= 1 && ((long long)B * OH * OW) % batch_rows == 0), EGR_ERR_ARG,
              "bad operand scheme: needs w_scale > 0, row_amax and a batch row count that divides the GEMM rows");
    EGR_CHECK(!w3 || ((Cin % BK) == 0 && (((uintptr_t)x) & 15) == 0 && (((uintptr_t)w3) & 15) == 0 && (!gn_scale || sch == 1)), EGR_ERR_ARG,
              "split conv needs Cin %% 16 == 0, 16-byte aligned x / w3 and no fused input affine (scheme 1: only the input-stationary 3x3 kernel fuses one)");
    EGR_CHECK(!gn_scale || (gn_shift && (Cin % BK) == 0 && (((uintptr_t)x) & 15) == 0), EGR_ERR_ARG,
              "fused input affine needs Cin %% 16 == 0 and a 16-byte aligned input");
    EGR_CHECK(B >= 1 && H >= 1 && W >= 1 && Cin >= 1 && OH >= 1 && OW >= 1 && Cout >= 1 && KH >= 1 && KW >= 1 &&
                  stride >= 1 && dil >= 1,
              EGR_ERR_ARG, "bad conv geometry");
    const long long M = (long long)B * OH * OW;
    EGR_CHECK(M < (1LL << 31) && (long long)KH * KW * Cin < (1LL << 31), EGR_ERR_ARG, "conv too large for 32-bit indexing");
    ConvP p;
    memset(&p, 0, sizeof(p));
    p.x = x; p.w = w; p.w3 = (const uint4*)w3; p.bias = bias; p.bias_b = bias_b; p.res = res; p.y = y;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.OH = OH; p.OW = OW; p.Cout = Cout; p.KH = KH; p.KW = KW;
    p.stride = stride; p.dil = dil; p.pad_t = pad_t; p.pad_l = pad_l; p.up2 = up2; p.act = act; p.act_param = act_param;
    p.M = (int)M; p.K = KH * KW * Cin;
    p.sch = sch; p.out_scale = sch ? 1.0f / w_scale : 1.0f; p.row_amax = sch ? (const unsigned*)row_amax : nullptr;
    p.rows_div = sch ? (int)(M / batch_rows) : 1;
    p.out_amax = (sch && nz == 1) ? (unsigned*)out_amax : nullptr;
    p.gn_part = (float2*)gn_part;
    EGR_CHECK(osy >= 1 && osx >= 1 && ooy >= 0 && oox >= 0 && (OH - 1) * osy + ooy < OHF && (OW - 1) * osx + oox < OWF,
              EGR_ERR_ARG, "bad output placement");
    p.osy = osy; p.osx = osx; p.ooy = ooy; p.oox = oox; p.OHF = OHF; p.OWF = OWF;
    const bool vec = (Cin % BK) == 0 && (((uintptr_t)x) & 15) == 0;
    int bn = w3 ? s3_bn(Cout) : (Cout > 64 ? 128 : (Cout > 32 ? 64 : 32));
    // short-K layers that cannot use split-K (the transformer blocks' linears: a few thousand rows, K <= 496): narrower column
    // tiles until the grid covers the 256 CUs
    static const bool narrow = !(getenv("EGR_S3_NARROW") && atoi(getenv("EGR_S3_NARROW")) == 0);
    if (w3 && narrow && nz <= 1 && (KH * KW * Cin + BK - 1) / BK < 32)
        while (bn > 64 && ((M + 127) / 128) * ((Cout + bn - 1) / bn) < 256) bn >>= 1;
    // (fp16 terms: the 256 x 128 tile loses to 128 x 128 on every layer that took it -- short-K streaming GEMMs with 128 outputs, 5 per
    // forward: 5.1 -> 3.1 ms, profiles/r05/flashsr_kernel_experiments.log item 6 -- half the registers per workgroup, twice the workgroups in flight)
    int bm = w3 ? (sch ? 128 : s3_bm(M, Cout, bn)) : BM;
    dim3 grid((unsigned)((M + bm - 1) / bm), (unsigned)((Cout + bn - 1) / bn));
    hipStream_t st = (hipStream_t)stream;
    // split-K when the output tiles alone cannot fill the chip and the K loop is long (deep UNet / latent layers)
    const int tiles = (int)(grid.x * grid.y), ktiles = (p.K + BK - 1) / BK;
    p.ksplit = 1; p.kt_per = ktiles; p.ws = nullptr;
    p.zx = zx; p.zw = zw; p.zy = zy;
    p.gn_scale = gn_scale; p.gn_shift = gn_shift; p.gn_silu = gn_silu;
    { int zrc = zero_page(&p.zeros); if (zrc) return zrc; }
    p.nz = nz; p.zs_nzb = 0;
    if (nz > 1) grid.z = nz;
    if (w3 && nz > 1 && KH == 1 && KW == 1 && H == 1 && W == 1 && stride == 1 && zw == (long long)(p.K / BK) * Cout * (sch ? 4 : 6) &&
        !bias && !bias_b && !res && act == 0 && osy == 1 && osx == 1 && OHF == OH && OWF == OW) {
        p.zs_nzb = s3_zs_nzb(M, Cout, 128, bn, nz, p.K);          // Winograd GEMMs: stream several z per workgroup
        if (p.zs_nzb > 0) {                                       // (128-row tiles: the 256-row variant would spill)
            bm = 128;
            grid.x = (unsigned)((M + bm - 1) / bm);
            grid.z = (nz + p.zs_nzb - 1) / p.zs_nzb;
        }
    }
    if (nz == 1 && tiles < 192 && ktiles >= 32) {
        int S = (768 + tiles - 1) / tiles;
        if (S > ktiles / 8) S = ktiles / 8;
        if (S > 64) S = 64;
        if (S >= 2) {
            const int per = (ktiles + S - 1) / S;
            S = (ktiles + per - 1) / per;
            float* ws = nullptr;
            int rc = splitk_workspace((size_t)S * M * Cout * sizeof(float), st, &ws);
            if (rc) return rc;
            p.ksplit = S; p.kt_per = per; p.ws = ws;
            grid.z = S;
            // (with out_amax: the reduction kernel writes y and tracks the row maxima)
            EGR_CHECK(!p.out_amax || batch_rows <= 1024, EGR_ERR_UNSUPPORTED, "out_amax with split-K serves up to 1024 batch rows");
        }
    }
#define LAUNCH(BN_, V_) hipLaunchKernelGGL((k_conv_igemm<BN_, V_>), grid, dim3(256), 0, st, p)
    if (w3 && launch_conv3x3_is(p, st)) {}
    else if (w3 && gn_scale) {
        set_error("egr_conv_h2_gn: the shape does not qualify for the input-stationary 3x3 kernel (stride 1, pad 1, H %% 4 == 0, W %% 32 == 0, "
                  "Cin %% 32 == 0, >= 512 tiles, one batch row per image)");
        return EGR_ERR_UNSUPPORTED;
    }
    else if (w3 && launch_conv1d_s3(p, st)) {}
    else if (w3) launch_conv_s3(bm, bn, grid, st, p);
    else if (gn_scale) {        // fused-GroupNorm loader: separate instantiations so the plain kernels pay nothing for it
        if (bn == 128) hipLaunchKernelGGL((k_conv_igemm<128, true, true>), grid, dim3(256), 0, st, p);
        else if (bn == 64) hipLaunchKernelGGL((k_conv_igemm<64, true, true>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((k_conv_igemm<32, true, true>), grid, dim3(256), 0, st, p);
    } else
    if (bn == 128) { if (vec) LAUNCH(128, true); else LAUNCH(128, false); }
    else if (bn == 64) { if (vec) LAUNCH(64, true); else LAUNCH(64, false); }
    else { if (vec) LAUNCH(32, true); else LAUNCH(32, false); }
#undef LAUNCH
    if (p.ksplit > 1) {
        long long nb = (M * Cout + 255) / 256;
        if (nb > 2048) nb = 2048;
        if (p.out_amax) hipLaunchKernelGGL(k_splitk_reduce<true>, dim3((unsigned)nb), dim3(256), 0, st, p);
        else hipLaunchKernelGGL(k_splitk_reduce<false>, dim3((unsigned)nb), dim3(256), 0, st, p);
    }
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}

extern "C" int egr_bgemm(const float* a, const float* b, float* c, int nb1, int nb2, int M, int N, int K, int lda,
                         int ldb, int ldc, int64_t sa1, int64_t sa2, int64_t sb1, int64_t sb2, int64_t sc1,
                         int64_t sc2, int transB, float alpha, void* stream) {
    EGR_CHECK(a && b && c && nb1 >= 1 && nb2 >= 1 && M >= 1 && N >= 1 && K >= 1, EGR_ERR_ARG, "bad gemm argument");
    EGR_CHECK((long long)nb1 * nb2 <= 65535, EGR_ERR_ARG, "too many batches");
    GemmP p;
    p.a = a; p.b = b; p.c = c; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.nb2 = nb2;
    p.transB = transB; p.sa1 = sa1; p.sa2 = sa2; p.sb1 = sb1; p.sb2 = sb2; p.sc1 = sc1; p.sc2 = sc2; p.alpha = alpha;
    const int bn = N > 64 ? 128 : (N > 32 ? 64 : 32);
    dim3 grid((M + BM - 1) / BM, (N + bn - 1) / bn, nb1 * nb2);
    hipStream_t st = (hipStream_t)stream;
    if (bn == 128) hipLaunchKernelGGL((k_bgemm<128>), grid, dim3(256), 0, st, p);
    else if (bn == 64) hipLaunchKernelGGL((k_bgemm<64>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((k_bgemm<32>), grid, dim3(256), 0, st, p);
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}

// Wall time (us) of two `spin_us` busy kernels launched back to back on streams a and b: about spin_us when the streams sit on
// different hardware queues, about twice that when the runtime maps them onto the same one (HIP multiplexes its streams onto a few
// hardware queues; which pairs collide is not visible through the API).
extern "C" int egr_streams_overlap_us(void* a, void* b, int spin_us, double* elapsed_us) {
    EGR_CHECK(elapsed_us && spin_us >= 10 && spin_us <= 5000, EGR_ERR_ARG, "bad argument");
    hipStream_t sa = (hipStream_t)a, sb = (hipStream_t)b;
    const long long ticks = (long long)spin_us * 100;
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, sa, 100ll, (unsigned*)nullptr);     // warm both queues
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, sb, 100ll, (unsigned*)nullptr);
    EGR_HIP(hipStreamSynchronize(sa));
    EGR_HIP(hipStreamSynchronize(sb));
    const auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, sa, ticks, (unsigned*)nullptr);
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, sb, ticks, (unsigned*)nullptr);
    EGR_HIP(hipStreamSynchronize(sa));
    EGR_HIP(hipStreamSynchronize(sb));
    *elapsed_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}
