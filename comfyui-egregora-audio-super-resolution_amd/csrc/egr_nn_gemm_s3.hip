// fp32 implicit-GEMM convolution on the bf16 matrix pipe of gfx950 — same function as k_conv_igemm, 16x the MFMA rate.
//
// Every fp32 operand is split exactly into three bf16 terms  x = x0 + x1 + x2  (x0 = bf16(x), x1 = bf16(x - x0),
// x2 = bf16(x - x0 - x1); the residuals are exact in fp32).  A bf16 x bf16 product is exact in fp32 and
// v_mfma_f32_32x32x16_bf16 accumulates in fp32, so the six partial products
//     a0 b0 + a0 b1 + a1 b0 + a1 b1 + a0 b2 + a2 b0
// reproduce the fp32 product up to the dropped terms a1 b2 + a2 b1 + a2 b2 < 3 * 2^-24 |a b| — the rounding an IEEE
// fp32 multiply makes anyway.  Measured against float64 the result is MORE accurate than the fmaf chain of
// v_mfma_f32_32x32x2_f32 (fewer roundings per accumulator: one per 16 k instead of one per k); see
// tests/test_gpu_flashsr.py::test_split3_conv_error_vs_float64.  Six bf16 MFMAs of 32 cycles replace eight f32 MFMAs of
// 64 cycles per 32x32x16 block: 2.67x the f32 matrix peak (2.5 PFLOP/s / 6 = 417 TFLOP/s of fp32-equivalent work).
//
// Weights are split once on the device (egr_split3_pack -> [slab][3][Cout][16] bf16); activations are split by the
// loader on their way into LDS (v_cvt_pk_bf16_f32 + two subtractions per term).  LDS tiles are per plane
// [row][2 chunks of 8 bf16] with the chunk index XOR-ed by bit 3 of the row, which makes both the loader's
// ds_write_b128 and the MFMA operand ds_read_b128 (lane l: row l&31, k-half l>>5) conflict-free at a 32-byte pitch.
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "egr_conv.h"
#include "egr_s3_split.h"

namespace egr {

// packed fp32 weights -> [slab][2][Cout][16] f16 terms of w * scale
__global__ __launch_bounds__(256) void k_split2h_pack(const float* __restrict__ w, uint4* __restrict__ w2, long long nslabs, int Cout,
                                                      float scale) {
    const long long total = nslabs * Cout * 2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int half = (int)(i & 1);
        const long long rn = i >> 1;
        const long long slab = rn / Cout;
        const int n = (int)(rn - slab * Cout);
        const float4* src = (const float4*)(w + (rn * 16 + half * 8));
        uint4 q0, q1;
        split2h_x8(src[0], src[1], scale, q0, q1);
        uint4* dst = w2 + ((slab * 2) * Cout + n) * 2 + half;
        dst[0] = q0;
        dst[(size_t)Cout * 2] = q1;
    }
}

// max |x| over n floats into *slot (bits of a non-negative float); the host zeroes the slot
__global__ __launch_bounds__(256) void k_absmax(const float* __restrict__ x, long long n, unsigned* __restrict__ slot) {
    float m = 0.f;
    const long long n4 = n >> 2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const float4 v = ((const float4*)x)[i];
        m = fmaxf(fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))), m);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = fmaxf(m, fabsf(x[(n4 << 2) + threadIdx.x]));
    amax_commit(slot, m);
}

// out[r] (bits of a non-negative float, zeroed by the caller) raised to max |x| over row r of x: nz segments of `per_row` floats
// at x + z * zx + r * per_row (nz = 1: one contiguous run; the Winograd V tensor: one segment per component).  grid (blocks, R, nz)
__global__ __launch_bounds__(256) void k_absmax_rows(const float* __restrict__ x, long long per_row, long long zx, unsigned* __restrict__ out) {
    const float* xr = x + (size_t)blockIdx.z * zx + (size_t)blockIdx.y * per_row;
    float m = 0.f;
    const long long n4 = per_row >> 2;
    // four independent 16-byte loads in flight per thread
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long st = (long long)gridDim.x * blockDim.x;
    for (; i + 3 * st < n4; i += 4 * st) {
        const float4 a = ((const float4*)xr)[i], b = ((const float4*)xr)[i + st], c = ((const float4*)xr)[i + 2 * st], d = ((const float4*)xr)[i + 3 * st];
        m = fmaxf(fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))), m);
        m = fmaxf(fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w))), m);
        m = fmaxf(fmaxf(fmaxf(fabsf(c.x), fabsf(c.y)), fmaxf(fabsf(c.z), fabsf(c.w))), m);
        m = fmaxf(fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), fmaxf(fabsf(d.z), fabsf(d.w))), m);
    }
    for (; i < n4; i += st) {
        const float4 a = ((const float4*)xr)[i];
        m = fmaxf(fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))), m);
    }
    if (blockIdx.x == 0 && threadIdx.x < (per_row & 3)) m = fmaxf(m, fabsf(xr[(n4 << 2) + threadIdx.x]));
    amax_commit(out + (size_t)blockIdx.y * EGR_ROW_AMAX_STRIDE, m);
}

// the same for rows that are not 16-byte aligned (element loads)
__global__ __launch_bounds__(256) void k_absmax_rows_scalar(const float* __restrict__ x, long long per_row, long long zx, unsigned* __restrict__ out) {
    const float* xr = x + (size_t)blockIdx.z * zx + (size_t)blockIdx.y * per_row;
    float m = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < per_row; i += (long long)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(xr[i]));
    amax_commit(out + (size_t)blockIdx.y * EGR_ROW_AMAX_STRIDE, m);
}

__global__ __launch_bounds__(256) void k_split3_pack(const float* __restrict__ w, uint4* __restrict__ w3, long long nslabs,
                                                     int Cout) {
    // one thread per (slab, n, half): 8 consecutive k of one output channel
    const long long total = nslabs * Cout * 2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int half = (int)(i & 1);
        const long long rn = i >> 1;
        const long long slab = rn / Cout;
        const int n = (int)(rn - slab * Cout);
        const float4* src = (const float4*)(w + (rn * 16 + half * 8));
        uint4 q0, q1, q2;
        split3_x8(src[0], src[1], q0, q1, q2);
        uint4* dst = w3 + ((slab * 3) * Cout + n) * 2 + half;
        dst[0] = q0;
        dst[(size_t)Cout * 2] = q1;
        dst[(size_t)Cout * 4] = q2;
    }
}

// plain tile store (z-streamed GEMMs have no bias / residual / activation / placement)
template <int TM, int TN, bool OST = false>
__device__ __forceinline__ void store_tile_plain(f32x16 (&acc)[TM][TN], float* __restrict__ y, int M, int Cout, int m0, int n0,
                                                 int wm0, int wn0, float os_u, const float* os_tab = nullptr) {
    const int lane = threadIdx.x & 63, col = lane & 31, rhalf = lane >> 5;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * rhalf;
            if (m < M) {
                float* row = y + (size_t)m * Cout + n0 + wn0 + col;
                const float os = OST ? os_tab[m - m0] * os_u : os_u;
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    if (n0 + wn0 + j * 32 + col < Cout) row[j * 32] = acc[i][j][r] * os;
            }
        }
}

template <int BM, int BN, int PF, bool ZS = false, int SCH = 0>
__global__ __launch_bounds__(256, 2) void k_conv_s3(ConvP p) {
    typedef S3Cfg<BM, BN> TC;
    constexpr int TM = TC::TM, TN = TC::TN;
    constexpr int AP = BM / 128;                                  // A chunks (8 k of one row) per thread per slab
    constexpr int NP = SCH ? 2 : 3;                               // operand planes (terms of the split)
    __shared__ uint4 As[2][NP][BM * 2];
    __shared__ uint4 Bs[2][NP][BN * 2];
    __shared__ float os_tab[SCH ? BM : 1];                        // scheme 1: output scale of every row of the block tile
    __shared__ unsigned om_tab[SCH ? BM : 1];                     // scheme 1: max |y| per batch row of the block tile (p.out_amax)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm0 = (wave / TC::WN) * (BM / TC::WM), wn0 = (wave % TC::WN) * (BN / TC::WN);
    int bx = blockIdx.x, by = blockIdx.y;
    if (p.xcd_remap) {                     // launcher guarantees gridDim.x * gridDim.y % 8 == 0 and gridDim.y > 1
        const int lin = by * gridDim.x + bx, per = (gridDim.x * gridDim.y) >> 3;
        const int j = (lin & 7) * per + (lin >> 3);
        bx = j / gridDim.y;
        by = j - bx * gridDim.y;
    }
    const int m0 = bx * BM, n0 = by * BN;
    const int LH = p.up2 ? 2 * p.H : p.H, LW = p.up2 ? 2 * p.W : p.W;
    // z-streaming: this workgroup owns z problems [z0, z0 + nzl) of its tile; otherwise blockIdx.z is the z problem
    const int z0 = ZS ? blockIdx.z * p.zs_nzb : blockIdx.z;
    const int nzl = ZS ? min(p.zs_nzb, p.nz - z0) : 1;
    if (p.ksplit <= 1 && (gridDim.z > 1 || ZS)) {
        p.x += (size_t)z0 * p.zx;
        p.w3 += (size_t)z0 * p.zw;
        p.y += (size_t)z0 * p.zy;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int ktiles_all = p.K / S3_BK;
    const int kt_begin = p.ksplit > 1 ? blockIdx.z * p.kt_per : 0;
    const int kt_end = ZS ? nzl * ktiles_all : (p.ksplit > 1 ? min(ktiles_all, kt_begin + p.kt_per) : ktiles_all);

    // ---- A: thread owns 8 consecutive k (half ah) of tile rows ar (and ar + 128 when BM = 256) ----
    // (plain scalars on purpose: arrays captured by the lambdas below end up in scratch)
    const int ar = tid >> 1, ah = tid & 1;
    float a_scale0 = 1.f, a_scale1 = 1.f;                         // scheme 1: the operand scales of this thread's rows
    if constexpr (SCH == 1) {
        for (int r = tid; r < BM; r += 256) {
            os_tab[r] = h2_row_inv(p.row_amax[(size_t)(min(m0 + r, p.M - 1) / p.rows_div) * EGR_ROW_AMAX_STRIDE]);
            om_tab[r] = 0u;
        }
        a_scale0 = h2_row_scale(p.row_amax[(size_t)(min(m0 + ar, p.M - 1) / p.rows_div) * EGR_ROW_AMAX_STRIDE]);
        if (AP > 1) a_scale1 = h2_row_scale(p.row_amax[(size_t)(min(m0 + ar + 128, p.M - 1) / p.rows_div) * EGR_ROW_AMAX_STRIDE]);
    }
    int a_iy0, a_ix0, a_iy1 = 0, a_ix1 = 0;
    size_t a_base0, a_base1 = 0;
    bool a_ok0, a_ok1 = false;
    {
        const int m = m0 + ar;
        a_ok0 = m < p.M;
        const int mm = a_ok0 ? m : 0;
        const int ox = mm % p.OW, t = mm / p.OW, oy = t % p.OH, b = t / p.OH;
        a_iy0 = oy * p.stride - p.pad_t;
        a_ix0 = ox * p.stride - p.pad_l;
        a_base0 = (size_t)b * p.H * p.W;
    }
    if (AP > 1) {
        const int m = m0 + ar + 128;
        a_ok1 = m < p.M;
        const int mm = a_ok1 ? m : 0;
        const int ox = mm % p.OW, t = mm / p.OW, oy = t % p.OH, b = t / p.OH;
        a_iy1 = oy * p.stride - p.pad_t;
        a_ix1 = ox * p.stride - p.pad_l;
        a_base1 = (size_t)b * p.H * p.W;
    }
    int tap = 0, c0 = 0;
    {
        const int tpt = p.Cin / S3_BK;
        tap = kt_begin / tpt;
        c0 = (kt_begin - tap * tpt) * S3_BK;
    }
    const float* aptr0 = p.zeros;
    const float* aptr1 = p.zeros;
    int astep0 = 0, astep1 = 0;
    auto set_tap = [&](int tp) {
        // z-streaming GEMMs (1x1 taps): the "tap" counter walks the z problems, each zx floats further
        const int ky = ZS ? 0 : tp / p.KW, kx = ZS ? 0 : tp - ky * p.KW;
        const size_t zoff = ZS ? (size_t)tp * p.zx : 0;
        {
            const int iy = a_iy0 + ky, ix = a_ix0 + kx * p.dil;
            const bool ok = a_ok0 && (unsigned)iy < (unsigned)LH && (unsigned)ix < (unsigned)LW;
            const int py = p.up2 ? iy >> 1 : iy, px = p.up2 ? ix >> 1 : ix;
            aptr0 = ok ? p.x + zoff + (a_base0 + (size_t)py * p.W + px) * p.Cin + ah * 8 : p.zeros + ah * 8;
            astep0 = ok ? 1 : 0;
        }
        if (AP > 1) {
            const int iy = a_iy1 + ky, ix = a_ix1 + kx * p.dil;
            const bool ok = a_ok1 && (unsigned)iy < (unsigned)LH && (unsigned)ix < (unsigned)LW;
            const int py = p.up2 ? iy >> 1 : iy, px = p.up2 ? ix >> 1 : ix;
            aptr1 = ok ? p.x + zoff + (a_base1 + (size_t)py * p.W + px) * p.Cin + ah * 8 : p.zeros + ah * 8;
            astep1 = ok ? 1 : 0;
        }
    };
    set_tap(tap);
    const int a_slot = ar * 2 + (ah ^ ((ar >> 3) & 1));          // row ar + 128 lands 256 slots further

    // ---- B: 2*NP*BN 16-byte chunks per slab (NP planes x BN channels x 2 halves), up to 6 per thread ----
    constexpr int NBQ = 2 * NP * BN;
    const size_t b_slab = (size_t)p.Cout * 2 * NP;               // uint4 per slab
#define S3_BSETUP(I, PTR, STEP, SLOT)                                                                                \
    const uint4* PTR;                                                                                                \
    unsigned STEP;                                                                                                   \
    int SLOT;                                                                                                        \
    {                                                                                                                \
        const int e = tid + 256 * (I);                                                                               \
        const int plane = e / (2 * BN), rem = e - plane * 2 * BN, nl = rem >> 1, half = rem & 1;                     \
        const bool ok = e < NBQ && n0 + nl < p.Cout;                                                                 \
        SLOT = plane * (BN * 2) + nl * 2 + (half ^ ((nl >> 3) & 1));                                                 \
        PTR = ok ? p.w3 + (size_t)kt_begin * b_slab + ((size_t)plane * p.Cout + n0 + nl) * 2 + half                  \
                 : (const uint4*)p.zeros;                                                                            \
        STEP = ok ? (unsigned)b_slab : 0u;                                                                                    \
    }
    S3_BSETUP(0, bptr0, bstep0, bslot0)
    S3_BSETUP(1, bptr1, bstep1, bslot1)
    S3_BSETUP(2, bptr2, bstep2, bslot2)
    S3_BSETUP(3, bptr3, bstep3, bslot3)
    S3_BSETUP(4, bptr4, bstep4, bslot4)
    S3_BSETUP(5, bptr5, bstep5, bslot5)
#undef S3_BSETUP

    // staged tile registers: one set per slab in flight (PF = 2 sets: a load has two slab-times to land)
    struct Stage {
        float4 a0, a1, a2, a3;
        uint4 b0, b1, b2, b3, b4, b5;
    };
    auto load_tile_issue = [&](Stage& r) {            // the global loads of the next tile (no control flow)
        {
            const float* s = aptr0 + astep0 * c0;
            r.a0 = *(const float4*)s;
            r.a1 = *(const float4*)(s + 4);
        }
        if (AP > 1) {
            const float* s = aptr1 + astep1 * c0;
            r.a2 = *(const float4*)s;
            r.a3 = *(const float4*)(s + 4);
        }
        r.b0 = *bptr0;
        if (256 < NBQ) r.b1 = *bptr1;
        if (512 < NBQ) r.b2 = *bptr2;
        if (768 < NBQ) r.b3 = *bptr3;
        if (1024 < NBQ) { r.b4 = *bptr4; r.b5 = *bptr5; }
    };
    auto load_tile_advance = [&]() {                  // pointer bookkeeping; the tap change is the only branch
        bptr0 += bstep0;
        if (256 < NBQ) bptr1 += bstep1;
        if (512 < NBQ) bptr2 += bstep2;
        if (768 < NBQ) bptr3 += bstep3;
        if (1024 < NBQ) { bptr4 += bstep4; bptr5 += bstep5; }
        c0 += S3_BK;
        if (c0 >= p.Cin) { c0 = 0; ++tap; set_tap(tap); }
    };
    auto load_tile = [&](Stage& r) {
        load_tile_issue(r);
        load_tile_advance();
    };
    auto store_tile = [&](const Stage& r, int buf) {
        uint4 q[3];
        split_x8<SCH>(r.a0, r.a1, a_scale0, q);
        {
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) As[buf][pl][a_slot] = q[pl];
            if (AP > 1) {
                split_x8<SCH>(r.a2, r.a3, a_scale1, q);
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) As[buf][pl][a_slot + 256] = q[pl];
            }
            if (tid < NBQ) Bs[buf][0][bslot0] = r.b0;
            if (tid + 256 < NBQ) Bs[buf][0][bslot1] = r.b1;
            if (tid + 512 < NBQ) Bs[buf][0][bslot2] = r.b2;
            if (768 < NBQ) Bs[buf][0][bslot3] = r.b3;
            if (1024 < NBQ) {
                Bs[buf][0][bslot4] = r.b4;
                Bs[buf][0][bslot5] = r.b5;
            }
        }
    };

    // operand fetch: lane (li = lane&31, lk = lane>>5) reads chunk lk of row li (+32 i) — k 0..7 on lanes 0-31, 8..15 on 32-63
    const int li = lane & 31, lk = lane >> 5;
    const int o_slot = li * 2 + (lk ^ ((li >> 3) & 1));

    // one slab: MFMAs on LDS buffer `cur`; tile kt+1 (staged in SN) -> LDS buffer cur^1; tile kt+1+PF -> SN
    // FULL: a middle slab (tile kt+1 is stored, tile kt+1+PF is loaded, unconditionally): its body up to the pointer
    // bookkeeping is ONE basic block, which lets the scheduler interleave the split VALU / LDS stores with the MFMAs
    auto slab = [&](int kt, int cur, Stage& sn, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        constexpr int TH = TM > 2 ? 1 : TM;                       // A sub-tiles in flight (register budget)
        constexpr int TNH = TN > 2 ? 1 : TN;                      // B sub-tiles in flight
#pragma unroll
        for (int j0 = 0; j0 < TN; j0 += TNH) {
            uint4 b[TNH][3];
#pragma unroll
            for (int j = 0; j < TNH; ++j)
#pragma unroll
                for (int q = 0; q < NP; ++q) b[j][q] = Bs[cur][q][(wn0 + (j0 + j) * 32) * 2 + o_slot];
#pragma unroll
            for (int i0 = 0; i0 < TM; i0 += TH) {
                uint4 a[TH][3];
#pragma unroll
                for (int i = 0; i < TH; ++i)
#pragma unroll
                    for (int q = 0; q < NP; ++q) a[i][q] = As[cur][q][(wm0 + (i0 + i) * 32) * 2 + o_slot];
                if (i0 == 0 && j0 == 0) {
                    if (FULL) {
                        store_tile(sn, cur ^ 1);
                        load_tile_issue(sn);
                    } else {
                        if (kt + 1 < kt_end) store_tile(sn, cur ^ 1);
                        if (kt + 1 + PF < kt_end) load_tile(sn);
                    }
                    // keeps the global loads above the MFMAs (see the fp16 path); the bf16 kernels spill with it (36 bytes per lane
                    // at 128 x 256) and stay as they were
                    if constexpr (SCH == 1) __builtin_amdgcn_sched_barrier(0);
                }
                // SCH 0: the six bf16 products term by term over the sub-tiles (smallest terms first); SCH 1: three f16 products
                if constexpr (SCH == 0) {
#define S3_MMA(QA, QB)                                                                                                   \
    _Pragma("unroll") for (int i = 0; i < TH; ++i) _Pragma("unroll") for (int j = 0; j < TNH; ++j) acc[i0 + i][j0 + j] =  \
        __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(a[i][QA]), as_bf(b[j][QB]), acc[i0 + i][j0 + j], 0, 0, 0);
                    S3_MMA(2, 0)
                    S3_MMA(0, 2)
                    S3_MMA(1, 1)
                    S3_MMA(1, 0)
                    S3_MMA(0, 1)
                    S3_MMA(0, 0)
#undef S3_MMA
                } else {
#define S3_MMA(QA, QB)                                                                                                   \
    _Pragma("unroll") for (int i = 0; i < TH; ++i) _Pragma("unroll") for (int j = 0; j < TNH; ++j) acc[i0 + i][j0 + j] =  \
        __builtin_amdgcn_mfma_f32_32x32x16_f16(as_hf(a[i][QA]), as_hf(b[j][QB]), acc[i0 + i][j0 + j], 0, 0, 0);
                    S3_MMA(1, 0)
                    S3_MMA(0, 1)
                    S3_MMA(0, 0)
#undef S3_MMA
                }
            }
        }
        if (FULL) load_tile_advance();
        if (ZS) {                                   // last slab of a z problem: store its tile, restart the accumulators
            const int done = kt + 1;
            const int zl = done / ktiles_all;
            if (done - zl * ktiles_all == 0) {
                store_tile_plain<TM, TN, SCH == 1>(acc, p.y + (size_t)(zl - 1) * p.zy, p.M, p.Cout, m0, n0, wm0, wn0, p.out_scale, os_tab);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            }
        }
        __syncthreads();
    };

    if constexpr (SCH == 1 && TM <= 2 && BM == 128) {
        // fp16 scheme, 128-row tiles: with half the matrix work per slab the loop was bound by the latency of the activation loads
        // (every A line is read by one workgroup only -- an HBM / MALL miss -- and had one slab to arrive).  Here the A tile is
        // requested TWO slabs ahead through two alternating register sets (8 registers each) and the weight tile, an L2 hit shared
        // by every row tile, one slab ahead; B is issued before A so that the wait for B (vmcnt counts in order) leaves the newer A
        // request in flight.
        struct SA { float4 a0, a1; };
        struct SB { uint4 b0, b1, b2, b3; };
        static_assert(NBQ <= 1024, "four weight chunks per thread");
        SA sa0, sa1;
        SB sb;
        sb.b0 = sb.b1 = sb.b2 = sb.b3 = make_uint4(0, 0, 0, 0);
        auto a_issue = [&](SA& r) {
            const float* sp = aptr0 + astep0 * c0;
            r.a0 = *(const float4*)sp;
            r.a1 = *(const float4*)(sp + 4);
        };
        auto a_adv = [&]() {
            c0 += S3_BK;
            if (c0 >= p.Cin) { c0 = 0; ++tap; set_tap(tap); }
        };
        auto b_issue = [&](SB& r) {
            r.b0 = *bptr0;
            if (256 < NBQ) r.b1 = *bptr1;
            if (512 < NBQ) r.b2 = *bptr2;
            if (768 < NBQ) r.b3 = *bptr3;
        };
        auto b_adv = [&]() {
            bptr0 += bstep0;
            if (256 < NBQ) bptr1 += bstep1;
            if (512 < NBQ) bptr2 += bstep2;
            if (768 < NBQ) bptr3 += bstep3;
        };
        auto storeA = [&](const SA& ra, int buf) {
            uint4 q[3];
            split_x8<1>(ra.a0, ra.a1, a_scale0, q);
            {
            As[buf][0][a_slot] = q[0];
            As[buf][1][a_slot] = q[1];
            }
        };
        auto storeB = [&](const SB& rb, int buf) {
            {
            if (NBQ >= 256 || tid < NBQ) Bs[buf][0][bslot0] = rb.b0;
            if (NBQ >= 512 || tid + 256 < NBQ) Bs[buf][0][bslot1] = rb.b1;
            if (NBQ >= 768 || tid + 512 < NBQ) Bs[buf][0][bslot2] = rb.b2;
            if (768 < NBQ) Bs[buf][0][bslot3] = rb.b3;
            }
        };
        // slab t (local index): MFMAs on LDS buffer cur; tile t+1 = (ra, sb) -> LDS buffer cur^1; B(t+2) -> sb, A(t+3) -> ra
        auto slab2 = [&](int t, int cur, SA& ra, auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
            const int n = kt_end - kt_begin;
            // the wave's A operands are read ONCE per slab and stay in registers, the B operands of column sub-tile j + 1 are fetched
            // while j multiplies (12 ds_read_b128 per wave and slab at TN = 4; the generic loop below re-reads A per sub-tile: 24).
            // The weight tile is the FIRST MFMA operand: accumulator registers run along the output channels (conv_epilogue_t)
            uint4 a[TM][2], b[2][2];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int q = 0; q < 2; ++q) a[i][q] = As[cur][q][(wm0 + i * 32) * 2 + o_slot];
#pragma unroll
            for (int q = 0; q < 2; ++q) b[0][q] = Bs[cur][q][wn0 * 2 + o_slot];
            // The weight tile needs no arithmetic: it goes to LDS and its successor is requested before anything else (pinned: the
            // machine scheduler otherwise sinks the requests below the MFMAs).  The activation tile's split, its LDS stores and the
            // request for A(t+3) are left to the scheduler to spread over the MFMAs: that request has two slabs to land, so it may
            // sit anywhere in this one, and it stays behind B's in program order (the next slab's wait for B leaves it in flight).
            if (FULL) {
                storeB(sb, cur ^ 1);
                b_issue(sb);
                __builtin_amdgcn_sched_barrier(0);
                storeA(ra, cur ^ 1);
                a_issue(ra);
            } else {
                if (t + 1 < n) storeB(sb, cur ^ 1);
                if (t + 2 < n) b_issue(sb);
                __builtin_amdgcn_sched_barrier(0);
                if (t + 1 < n) storeA(ra, cur ^ 1);
                if (t + 3 < n) a_issue(ra);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (j + 1 < TN) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) b[(j + 1) & 1][q] = Bs[cur][q][(wn0 + (j + 1) * 32) * 2 + o_slot];
                }
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_hf(b[j & 1][0]), as_hf(a[i][1]), acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_hf(b[j & 1][1]), as_hf(a[i][0]), acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_hf(b[j & 1][0]), as_hf(a[i][0]), acc[i][j], 0, 0, 0);
            }
            if (FULL || t + 2 < n) b_adv();
            if (FULL || t + 3 < n) a_adv();
            if (ZS) {                                   // last slab of a z problem: store its tile, restart the accumulators
                const int done = kt_begin + t + 1;
                const int zl = done / ktiles_all;
                if (done - zl * ktiles_all == 0) {
                    store_tile_plain_t<TM, TN>(acc, p.y + (size_t)(zl - 1) * p.zy, p.M, p.Cout, m0, n0, wm0, wn0, os_tab, p.out_scale);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
                }
            }
            __syncthreads();
        };
        typedef std::integral_constant<bool, true> FullT;
        typedef std::integral_constant<bool, false> TailT;
        const int n = kt_end - kt_begin;
        a_issue(sa0); a_adv();
        b_issue(sb); b_adv();
        storeA(sa0, 0);
        storeB(sb, 0);
        __syncthreads();
        sa1 = sa0;
        if (1 < n) { b_issue(sb); b_adv(); __builtin_amdgcn_sched_barrier(0); a_issue(sa1); a_adv(); }
        __builtin_amdgcn_sched_barrier(0);
        if (2 < n) { a_issue(sa0); a_adv(); }
        int t = 0;
        for (; t + 4 < n; t += 2) {                  // tile t+1 lives in sa1 for even t, in sa0 for odd t
            slab2(t, 0, sa1, FullT());
            slab2(t + 1, 1, sa0, FullT());
        }
        if (t < n) { slab2(t, 0, sa1, TailT()); ++t; }
        if (t < n) { slab2(t, 1, sa0, TailT()); ++t; }
        if (t < n) { slab2(t, 0, sa1, TailT()); ++t; }
        if (t < n) { slab2(t, 1, sa0, TailT()); ++t; }
        unsigned* const om = (!ZS && p.out_amax && p.ksplit <= 1) ? om_tab : nullptr;
        if (!ZS) conv_epilogue_t<TM, TN>(p, acc, m0, n0, wm0, wn0, os_tab, om);
        if (om) out_amax_commit(p, om_tab, m0, BM);
        return;
    }
    Stage s0, s1;
    s0.a2 = s0.a3 = s1.a0 = s1.a1 = s1.a2 = s1.a3 = make_float4(0.f, 0.f, 0.f, 0.f);
    s0.b0 = s0.b1 = s0.b2 = s1.b0 = s1.b1 = s1.b2 = make_uint4(0, 0, 0, 0);
    s0.b3 = s0.b4 = s0.b5 = s1.b3 = s1.b4 = s1.b5 = make_uint4(0, 0, 0, 0);
    load_tile(s0);
    store_tile(s0, 0);
    __syncthreads();
    typedef std::integral_constant<bool, true> FullT;
    typedef std::integral_constant<bool, false> TailT;
    if (PF == 1) {
        if (kt_begin + 1 < kt_end) load_tile(s0);
        int kt = kt_begin;
        for (; kt + 2 < kt_end; ++kt) slab(kt, (kt - kt_begin) & 1, s0, FullT());
        for (; kt < kt_end; ++kt) slab(kt, (kt - kt_begin) & 1, s0, TailT());
    } else {
        if (kt_begin + 1 < kt_end) load_tile(s0);
        if (kt_begin + 2 < kt_end) load_tile(s1);
        int kt = kt_begin;
        for (; kt + 1 < kt_end; kt += 2) {        // buffers and stage sets alternate with the parity of kt - kt_begin
            slab(kt, 0, s0, TailT());
            slab(kt + 1, 1, s1, TailT());
        }
        if (kt < kt_end) slab(kt, 0, s0, TailT());
    }
    unsigned* const om = (SCH == 1 && !ZS && p.out_amax && p.ksplit <= 1) ? om_tab : nullptr;
    if (!ZS) conv_epilogue<TM, TN, SCH == 1>(p, acc, m0, n0, wm0, wn0, nullptr, os_tab, om);
    if (om) out_amax_commit(p, om_tab, m0, BM);
}

// block-tile height for a problem: 256 rows when that still leaves >= 2 full rounds of workgroups on the 256 CUs
// output-channel tile: 256 wide when Cout is a multiple of 256 (half the activation loads / splits per MFMA: +20 %)
int s3_bn(int Cout) {
    static const bool no256 = getenv("EGR_S3_BN256") && atoi(getenv("EGR_S3_BN256")) == 0;
    if (!no256 && Cout >= 256 && Cout % 256 == 0) return 256;
    return Cout > 64 ? 128 : (Cout > 32 ? 64 : 32);
}

int s3_bm(long long M, int Cout, int bn) {
    if (bn != 128) return 128;
    const long long tiles256 = ((M + 255) / 256) * ((Cout + 127) / 128);
    return tiles256 >= 1024 ? 256 : 128;
}

// z problems per workgroup for z-streamed GEMMs: as many as keeps >= ~2048 workgroups in the grid (0: no streaming)
int s3_zs_nzb(long long M, int Cout, int bm, int bn, int nz, int K) {
    static const bool off = getenv("EGR_S3_ZS") && atoi(getenv("EGR_S3_ZS")) == 0;
    // the 256-wide streamed variant spills a little (in-loop tile store): it pays only where the K loop is short
    if (off || nz < 2 || bn < 128 || K % S3_BK || (bn == 256 && K > 256)) return 0;
    const long long tiles = ((M + bm - 1) / bm) * ((Cout + bn - 1) / bn);
    long long groups = (2048 + tiles - 1) / tiles;
    if (groups < 1) groups = 1;
    if (groups > nz) groups = nz;
    const int nzb = (int)((nz + groups - 1) / groups);
    return nzb >= 2 ? nzb : 0;
}

void launch_conv_s3(int bm, int bn, dim3 grid, hipStream_t st, const ConvP& p_in) {
    static const int pf = getenv("EGR_S3_PF") ? atoi(getenv("EGR_S3_PF")) : 1;
    static const bool remap = !(getenv("EGR_S3_XCD") && atoi(getenv("EGR_S3_XCD")) == 0);
    ConvP p = p_in;
    p.xcd_remap = (remap && grid.y > 1 && ((grid.x * grid.y) & 7) == 0) ? 1 : 0;
    if (p.sch) {                                  // two-term fp16 scheme (PF = 1 only)
        if (p.zs_nzb > 0) {
            if (bn == 256) hipLaunchKernelGGL((k_conv_s3<128, 256, 1, true, 1>), grid, dim3(256), 0, st, p);
            else hipLaunchKernelGGL((k_conv_s3<128, 128, 1, true, 1>), grid, dim3(256), 0, st, p);
        } else if (bn == 256) hipLaunchKernelGGL((k_conv_s3<128, 256, 1, false, 1>), grid, dim3(256), 0, st, p);
        else if (bn == 128) hipLaunchKernelGGL((k_conv_s3<128, 128, 1, false, 1>), grid, dim3(256), 0, st, p);
        else if (bn == 64) hipLaunchKernelGGL((k_conv_s3<128, 64, 1, false, 1>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((k_conv_s3<128, 32, 1, false, 1>), grid, dim3(256), 0, st, p);
        return;
    }
    if (p.zs_nzb > 0) {
        if (bn == 256) hipLaunchKernelGGL((k_conv_s3<128, 256, 1, true>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((k_conv_s3<128, 128, 1, true>), grid, dim3(256), 0, st, p);
        return;
    }
    if (bn == 256) {
        hipLaunchKernelGGL((k_conv_s3<128, 256, 1>), grid, dim3(256), 0, st, p);
        return;
    }
    if (bm == 256) hipLaunchKernelGGL((k_conv_s3<256, 128, 1>), grid, dim3(256), 0, st, p);
    else if (bn == 128) { if (pf == 2) hipLaunchKernelGGL((k_conv_s3<128, 128, 2>), grid, dim3(256), 0, st, p); else hipLaunchKernelGGL((k_conv_s3<128, 128, 1>), grid, dim3(256), 0, st, p); }
    else if (bn == 64) { if (pf == 2) hipLaunchKernelGGL((k_conv_s3<128, 64, 2>), grid, dim3(256), 0, st, p); else hipLaunchKernelGGL((k_conv_s3<128, 64, 1>), grid, dim3(256), 0, st, p); }
    else { if (pf == 2) hipLaunchKernelGGL((k_conv_s3<128, 32, 2>), grid, dim3(256), 0, st, p); else hipLaunchKernelGGL((k_conv_s3<128, 32, 1>), grid, dim3(256), 0, st, p); }
}

}  // namespace egr

using namespace egr;

extern "C" int egr_split2h_pack(const float* w_packed, void* w2, int64_t nslabs, int Cout, float scale, void* stream) {
    EGR_CHECK(w_packed && w2 && nslabs >= 1 && Cout >= 1 && scale > 0.f, EGR_ERR_ARG, "bad split2h pack argument");
    EGR_CHECK((((uintptr_t)w_packed) & 15) == 0 && (((uintptr_t)w2) & 15) == 0, EGR_ERR_ARG, "split2h pack needs 16-byte alignment");
    long long nb = (nslabs * Cout * 2 + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(k_split2h_pack, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, w_packed, (uint4*)w2,
                       (long long)nslabs, Cout, scale);
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}

extern "C" int egr_absmax(const float* x, int64_t n, float* slot, void* stream) {
    EGR_CHECK(x && slot && n >= 1 && (((uintptr_t)x) & 15) == 0, EGR_ERR_ARG, "bad absmax argument");
    long long nb = (n / 4 + 255) / 256;
    if (nb > 2048) nb = 2048;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(k_absmax, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, (long long)n, (unsigned*)slot);
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}

// row_amax[r] = bits of max |x| over batch row r (rows x per_row floats, contiguous; nz > 1: nz such blocks zx floats apart, the
// maximum taken over all of them).  row_amax must be zeroed by the caller (maxima of several tensors may share it).  16-byte
// aligned rows (per_row % 4 == 0, zx % 4 == 0) take the vector kernel.  Feeds egr_conv_h2.
extern "C" int egr_absmax_rows(const float* x, int rows, int64_t per_row, int nz, int64_t zx, float* row_amax, void* stream) {
    EGR_CHECK(x && row_amax && rows >= 1 && rows <= 65535 && per_row >= 1 && nz >= 1 && nz <= 65535, EGR_ERR_ARG, "bad absmax_rows argument");
    const bool vec = (((uintptr_t)x) & 15) == 0 && per_row % 4 == 0 && (nz == 1 || zx % 4 == 0);
    long long nb = vec ? (per_row / 4 + 1023) / 1024 : (per_row + 1023) / 1024;              // 4 loads per thread per round
    const long long cap = std::max(1LL, 4096LL / ((long long)rows * nz));
    if (nb > cap) nb = cap;
    if (nb < 1) nb = 1;
    if (vec)
        hipLaunchKernelGGL(k_absmax_rows, dim3((unsigned)nb, (unsigned)rows, (unsigned)nz), dim3(256), 0, (hipStream_t)stream, x, (long long)per_row,
                           (long long)zx, (unsigned*)row_amax);
    else
        hipLaunchKernelGGL(k_absmax_rows_scalar, dim3((unsigned)nb, (unsigned)rows, (unsigned)nz), dim3(256), 0, (hipStream_t)stream, x,
                           (long long)per_row, (long long)zx, (unsigned*)row_amax);
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}

extern "C" int egr_split3_pack(const float* w_packed, void* w3, int64_t nslabs, int Cout, void* stream) {
    EGR_CHECK(w_packed && w3 && nslabs >= 1 && Cout >= 1, EGR_ERR_ARG, "bad split3 pack argument");
    EGR_CHECK((((uintptr_t)w_packed) & 15) == 0 && (((uintptr_t)w3) & 15) == 0, EGR_ERR_ARG, "split3 pack needs 16-byte alignment");
    long long nb = (nslabs * Cout * 2 + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(k_split3_pack, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, w_packed, (uint4*)w3,
                       (long long)nslabs, Cout);
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Stride-1 1-D convolution (the vocoder's AMP blocks: k = 3 / 7 / 11, dilation 1 / 3 / 5, 16..256 channels), input-stationary:
// the implicit-GEMM loader above re-loads and re-splits every activation once per tap; here a workgroup splits its
// (128 + dil (k-1)) x CC halo tile ONCE into LDS and every tap reads its MFMA operands from it at a row offset, so the
// split work and the L2 traffic drop by the tap count.  Weights stream through the same double-buffered B tiles.
// LDS halo tile per plane: [row][CC/8 chunks of 8 bf16], chunk index XOR-ed with (row / (16 / NCH)) & (NCH - 1): conflict-free
// ds_read_b128 for any row offset (brute-forced over all offsets for NCH = 2, 4).
namespace egr {

template <int BN, int CC, int SCH = 0>
__global__ __launch_bounds__(256, 2) void k_conv1d_s3(ConvP p) {
    typedef S3Cfg<128, BN> TC;
    constexpr int TM = TC::TM, TN = TC::TN, NCH = CC / 8, NSL = CC / 16, RMAX = 128 + 50;
    constexpr int NP = SCH ? 2 : 3;
    __shared__ uint4 As[NP][RMAX * NCH];
    __shared__ uint4 Bs[2][NP][BN * 2];
    __shared__ float os_tab[SCH ? 128 : 1];
    __shared__ unsigned om_tab[1];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm0 = (wave / TC::WN) * (128 / TC::WM), wn0 = (wave % TC::WN) * (BN / TC::WN);
    const int L = p.W, b = blockIdx.z, l0 = blockIdx.x * 128, n0 = blockIdx.y * BN;
    float a_scale = 1.f;                         // scheme 1: the tile lies inside ONE batch row (launch_conv1d_s3: rows_div % 128 == 0)
    if constexpr (SCH == 1) {
        const unsigned bits = p.row_amax[(size_t)((b * L + l0) / p.rows_div) * EGR_ROW_AMAX_STRIDE];
        a_scale = h2_row_scale(bits);
        if (tid < 128) os_tab[tid] = h2_row_inv(bits);
        if (tid == 0) om_tab[0] = 0u;
    }
    const int R = 128 + p.dil * (p.KW - 1), pos0 = l0 - p.pad_l;
    const float* xb = p.x + (size_t)b * L * p.Cin;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // B tiles: 2*NP*BN chunks per slab, up to 3 per thread (as in k_conv_s3)
    constexpr int NBQ = 2 * NP * BN;
    const size_t b_slab = (size_t)p.Cout * 2 * NP;
    const int cpt = p.Cin / 16;                  // slabs per tap in the weight pack
#define C1_BSETUP(I, OFF, SLOT, OK)                                                                               \
    size_t OFF;                                                                                                      \
    int SLOT;                                                                                                        \
    bool OK;                                                                                                         \
    {                                                                                                                \
        const int e = tid + 256 * (I);                                                                               \
        const int plane = e / (2 * BN), rem = e - plane * 2 * BN, nl = rem >> 1, half = rem & 1;                     \
        OK = e < NBQ && n0 + nl < p.Cout;                                                                            \
        SLOT = plane * (BN * 2) + nl * 2 + (half ^ ((nl >> 3) & 1));                                                 \
        OFF = ((size_t)plane * p.Cout + n0 + nl) * 2 + half;                                                         \
    }
    C1_BSETUP(0, boff0, bslot0, bok0)
    C1_BSETUP(1, boff1, bslot1, bok1)
    C1_BSETUP(2, boff2, bslot2, bok2)
#undef C1_BSETUP
    const uint4* zq = (const uint4*)p.zeros;
    // weight tiles are prefetched FOUR slabs ahead through four register sets: at 6..24 MFMAs per slab a one-slab
    // distance is far shorter than an L2 round trip and the thin layers ran at the latency of one load per slab
    struct StageB { uint4 b0, b1, b2; };
    StageB s0, s1, s2, s3;
    s0.b0 = s0.b1 = s0.b2 = s1.b0 = s1.b1 = s1.b2 = s2.b0 = s2.b1 = s2.b2 = s3.b0 = s3.b1 = s3.b2 = make_uint4(0, 0, 0, 0);
    const int spc = p.KW * NSL;                  // slabs per channel chunk
    const int nchunks = p.Cin / CC, stotal = nchunks * spc;
    auto wslab = [&](int sg) {                   // weight-pack slab of global slab sg = (chunk, tap, 16-channel sub-slab)
        const int cc = sg / spc, s = sg - cc * spc, tap = s / NSL, cs = s - tap * NSL;
        return (size_t)(tap * cpt + cc * NSL + cs) * b_slab;
    };
    auto load_b = [&](int sg, StageB& r) {
        const uint4* base = p.w3 + wslab(sg);
        r.b0 = bok0 ? base[boff0] : zq[0];
        if (256 < NBQ) r.b1 = bok1 ? base[boff1] : zq[0];
        if (512 < NBQ) r.b2 = bok2 ? base[boff2] : zq[0];
    };
    auto store_b = [&](int buf, const StageB& r) {
        if (tid < NBQ) Bs[buf][0][bslot0] = r.b0;
        if (tid + 256 < NBQ) Bs[buf][0][bslot1] = r.b1;
        if (tid + 512 < NBQ) Bs[buf][0][bslot2] = r.b2;
    };
    const int li = lane & 31, lk = lane >> 5;
    const int ob_slot = li * 2 + (lk ^ ((li >> 3) & 1));

    // one slab; `nx` is the register set holding tile sg + 1 (tile T lives in set T % 4) and is refilled with tile sg + 5
    auto slab = [&](int sg, StageB& nx) {
        const int cur = sg & 1;
        const int cc = sg / spc, s = sg - cc * spc, tap = s / NSL, cs = s - tap * NSL;
        if (s == 0) {                            // new channel chunk: split its halo tile into LDS (all waves are past the
            const int c0 = cc * CC;              // previous chunk: the barrier that ended its last slab)
            // every load of the tile is requested before the first one is consumed (a run-time trip count would make each
            // round wait for its own HBM round trip: two or three serial latencies at the head of every workgroup)
            constexpr int NIT = (RMAX * NCH + 255) / 256;
            float4 hu[NIT], hv[NIT];
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
                const int e = tid + 256 * i;
                const int r = e / NCH, ch = e - r * NCH, pos = pos0 + r;
                const bool ok = e < R * NCH && (unsigned)pos < (unsigned)L;
                const float* src = ok ? xb + (size_t)pos * p.Cin + c0 + ch * 8 : p.zeros;
                hu[i] = *(const float4*)src;
                hv[i] = *(const float4*)(src + 4);
            }
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
                const int e = tid + 256 * i;
                if (e < R * NCH) {
                    const int r = e / NCH, ch = e - r * NCH;
                    uint4 q[3];
                    split_x8<SCH>(hu[i], hv[i], a_scale, q);
                    const int slot = r * NCH + (ch ^ ((r / (16 / NCH)) & (NCH - 1)));
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl) As[pl][slot] = q[pl];
                }
            }
        }
        __syncthreads();                         // B tile `cur` (stored one iteration ago) and the halo tile are visible
        uint4 bq[TN][3];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < NP; ++q) bq[j][q] = Bs[cur][q][(wn0 + j * 32) * 2 + ob_slot];
        uint4 aq[TM][3];
        const int ch = cs * 2 + lk;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int r = wm0 + i * 32 + li + tap * p.dil;
            const int slot = r * NCH + (ch ^ ((r / (16 / NCH)) & (NCH - 1)));
#pragma unroll
            for (int q = 0; q < NP; ++q) aq[i][q] = As[q][slot];
        }
        if (sg + 1 < stotal) store_b(cur ^ 1, nx);   // buffer cur^1 was last read in iteration sg-1, before this barrier
        if (sg + 5 < stotal) load_b(sg + 5, nx);
        if constexpr (SCH == 0) {
#define C1_MMA(QA, QB)                                                                                                   \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) acc[i][j] =            \
        __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(aq[i][QA]), as_bf(bq[j][QB]), acc[i][j], 0, 0, 0);
            C1_MMA(2, 0)
            C1_MMA(0, 2)
            C1_MMA(1, 1)
            C1_MMA(1, 0)
            C1_MMA(0, 1)
            C1_MMA(0, 0)
#undef C1_MMA
        } else {
            // weights as the first operand: transposed accumulators, 16-byte stores (conv_epilogue_t)
#define C1_MMA(QA, QB)                                                                                                   \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) acc[i][j] =            \
        __builtin_amdgcn_mfma_f32_32x32x16_f16(as_hf(bq[j][QB]), as_hf(aq[i][QA]), acc[i][j], 0, 0, 0);
            C1_MMA(1, 0)
            C1_MMA(0, 1)
            C1_MMA(0, 0)
#undef C1_MMA
        }
        if (s + 1 == spc) __syncthreads();       // last slab of the chunk: everyone is done with the halo tile
    };

    load_b(0, s0);
    store_b(0, s0);
    if (1 < stotal) load_b(1, s1);
    if (2 < stotal) load_b(2, s2);
    if (3 < stotal) load_b(3, s3);
    if (4 < stotal) load_b(4, s0);
    int sg = 0;
    for (; sg + 3 < stotal; sg += 4) {
        slab(sg, s1);
        slab(sg + 1, s2);
        slab(sg + 2, s3);
        slab(sg + 3, s0);
    }
    if (sg < stotal) slab(sg, s1);
    if (sg + 1 < stotal) slab(sg + 1, s2);
    if (sg + 2 < stotal) slab(sg + 2, s3);
    if (SCH) {
        unsigned* const om = p.out_amax ? om_tab : nullptr;          // the tile lies in one batch row: one word
        conv_epilogue_t<TM, TN>(p, acc, b * L + l0, n0, wm0, wn0, os_tab, om);
        if (om) out_amax_commit(p, om_tab, b * L + l0, 128);
    } else {
        conv_epilogue<TM, TN>(p, acc, b * L + l0, n0, wm0, wn0);
    }
}

// true when the input-stationary 1-D kernel applies; launches it
bool launch_conv1d_s3(const ConvP& p, hipStream_t st) {
    static const bool off = getenv("EGR_S3_CONV1D") && atoi(getenv("EGR_S3_CONV1D")) == 0;
    if (off || !p.w3 || p.H != 1 || p.KH != 1 || p.KW < 2 || p.stride != 1 || p.up2 || p.OW != p.W || (p.W % 128) != 0 ||
        p.dil * (p.KW - 1) > 50 || 2 * p.pad_l != p.dil * (p.KW - 1) || (p.Cin % 16) != 0 || p.ksplit > 1 || p.zs_nzb > 0 ||
        p.nz > 1 || p.osy != 1 || p.osx != 1 || p.OHF != p.OH || p.OWF != p.OW || (p.sch && (p.rows_div % 128) != 0))
        return false;
    const int bn = p.Cout > 64 ? 128 : (p.Cout > 32 ? 64 : 32);
    const dim3 grid(p.W / 128, (p.Cout + bn - 1) / bn, p.B);
    if (p.B > 65535) return false;
#define C1_LAUNCH(SCH_)                                                                                   \
    if (p.Cin % 32 == 0) {                                                                                \
        if (bn == 128) hipLaunchKernelGGL((k_conv1d_s3<128, 32, SCH_>), grid, dim3(256), 0, st, p);       \
        else if (bn == 64) hipLaunchKernelGGL((k_conv1d_s3<64, 32, SCH_>), grid, dim3(256), 0, st, p);    \
        else hipLaunchKernelGGL((k_conv1d_s3<32, 32, SCH_>), grid, dim3(256), 0, st, p);                  \
    } else {                                                                                              \
        if (bn == 128) hipLaunchKernelGGL((k_conv1d_s3<128, 16, SCH_>), grid, dim3(256), 0, st, p);       \
        else if (bn == 64) hipLaunchKernelGGL((k_conv1d_s3<64, 16, SCH_>), grid, dim3(256), 0, st, p);    \
        else hipLaunchKernelGGL((k_conv1d_s3<32, 16, SCH_>), grid, dim3(256), 0, st, p);                  \
    }
    if (p.sch) { C1_LAUNCH(1) } else { C1_LAUNCH(0) }
#undef C1_LAUNCH
    return true;
}

}  // namespace egr

// ------------------------------------------------------------------------------------------------------------------
// Strided batched C[b] = alpha * A[b] (M x K) * B[b]^T (B is N x K), both operands fp32 activations (attention: Q K^T, and
// P V with V transposed beforehand): the split-bf16 scheme with BOTH tiles split by the loader.  K % 16 == 0, rows 16-byte
// aligned.  Same LDS layout, swizzle and MFMA schedule as k_conv_s3.
namespace egr {

struct GemmS3P {
    const float* a; const float* b; float* c;
    int M, N, K, lda, ldb, ldc, nb2;
    long long sa1, sa2, sb1, sb2, sc1, sc2;
    float alpha;
    const float* zeros;
    // scheme 1 (two fp16 terms): bits of max |A| and max |B| per OUTER batch index b1 (EGR_ROW_AMAX_STRIDE apart); out_amax optional
    const unsigned* a_amax; const unsigned* b_amax; unsigned* out_amax;
};

// SCH 0: three bf16 terms per operand, six products; SCH 1: two fp16 terms of the operands scaled per outer batch index from their
// own maxima (as k_conv_s3's scheme 1: both operands are activations here), three products
template <int BN, int SCH = 0>
__global__ __launch_bounds__(256, 2) void k_bgemm_s3(GemmS3P p) {
    typedef S3Cfg<128, BN> TC;
    constexpr int TM = TC::TM, TN = TC::TN, NP = SCH ? 2 : 3;
    __shared__ uint4 As[2][NP][128 * 2];
    __shared__ uint4 Bs[2][NP][BN * 2];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm0 = (wave / TC::WN) * (128 / TC::WM), wn0 = (wave % TC::WN) * (BN / TC::WN);
    const int m0 = blockIdx.x * 128, n0 = blockIdx.y * BN;
    const int b1 = blockIdx.z / p.nb2, b2 = blockIdx.z - b1 * p.nb2;
    const float* A = p.a + b1 * p.sa1 + b2 * p.sa2;
    const float* Bm = p.b + b1 * p.sb1 + b2 * p.sb2;
    float* Cm = p.c + b1 * p.sc1 + b2 * p.sc2;
    float sa = 1.f, sb = 1.f, alpha = p.alpha;
    if constexpr (SCH == 1) {
        const unsigned ba = p.a_amax[(size_t)b1 * EGR_ROW_AMAX_STRIDE], bb = p.b_amax[(size_t)b1 * EGR_ROW_AMAX_STRIDE];
        sa = h2_row_scale(ba); sb = h2_row_scale(bb);
        alpha = p.alpha * h2_row_inv(ba) * h2_row_inv(bb);         // (both inverses are powers of two)
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // each thread owns one 8-float chunk of the A tile and (tid < 2 BN) one of the B tile per slab
    const int ar = tid >> 1, ah = tid & 1;
    const bool a_ok = m0 + ar < p.M, b_ok = tid < 2 * BN && n0 + ar < p.N;
    const float* aptr = a_ok ? A + (size_t)(m0 + ar) * p.lda + ah * 8 : p.zeros;
    const float* bptr = b_ok ? Bm + (size_t)(n0 + ar) * p.ldb + ah * 8 : p.zeros;
    const int astep = a_ok ? 16 : 0, bstep = b_ok ? 16 : 0;
    const int slot = ar * 2 + (ah ^ ((ar >> 3) & 1));
    const int li = lane & 31, lk = lane >> 5;
    const int o_slot = li * 2 + (lk ^ ((li >> 3) & 1));
    const int ktiles = p.K / S3_BK;

    float4 ra0, ra1, rb0, rb1;
    auto load = [&]() {
        ra0 = *(const float4*)aptr; ra1 = *(const float4*)(aptr + 4);
        rb0 = *(const float4*)bptr; rb1 = *(const float4*)(bptr + 4);
        aptr += astep; bptr += bstep;
    };
    auto store = [&](int buf) {
        uint4 q0, q1, q2;
        if constexpr (SCH == 0) {
            split3_x8(ra0, ra1, q0, q1, q2);
            As[buf][0][slot] = q0; As[buf][1][slot] = q1; As[buf][2][slot] = q2;
            if (tid < 2 * BN) {
                split3_x8(rb0, rb1, q0, q1, q2);
                Bs[buf][0][slot] = q0; Bs[buf][1][slot] = q1; Bs[buf][2][slot] = q2;
            }
        } else {
            split2h_x8(ra0, ra1, sa, q0, q1);
            As[buf][0][slot] = q0; As[buf][1][slot] = q1;
            if (tid < 2 * BN) {
                split2h_x8(rb0, rb1, sb, q0, q1);
                Bs[buf][0][slot] = q0; Bs[buf][1][slot] = q1;
            }
        }
    };
    load();
    store(0);
    __syncthreads();
    if (1 < ktiles) load();
    for (int kt = 0; kt < ktiles; ++kt) {
        const int cur = kt & 1;
        uint4 a[TM][3], b[TN][3];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int q = 0; q < NP; ++q) a[i][q] = As[cur][q][(wm0 + i * 32) * 2 + o_slot];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < NP; ++q) b[j][q] = Bs[cur][q][(wn0 + j * 32) * 2 + o_slot];
        if (kt + 1 < ktiles) store(cur ^ 1);
        if (kt + 2 < ktiles) load();
        if constexpr (SCH == 0) {
#define G3_MMA(QA, QB)                                                                                                   \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) acc[i][j] =            \
        __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(a[i][QA]), as_bf(b[j][QB]), acc[i][j], 0, 0, 0);
            G3_MMA(2, 0)
            G3_MMA(0, 2)
            G3_MMA(1, 1)
            G3_MMA(1, 0)
            G3_MMA(0, 1)
            G3_MMA(0, 0)
#undef G3_MMA
        } else {
#define G3_MMA(QA, QB)                                                                                                   \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) acc[i][j] =            \
        __builtin_amdgcn_mfma_f32_32x32x16_f16(as_hf(a[i][QA]), as_hf(b[j][QB]), acc[i][j], 0, 0, 0);
            G3_MMA(1, 0)
            G3_MMA(0, 1)
            G3_MMA(0, 0)
#undef G3_MMA
        }
        __syncthreads();
    }
    const int col = lane & 31, rhalf = lane >> 5;
    float vm = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * rhalf;
            if (m < p.M) {
                float* row = Cm + (size_t)m * p.ldc + n0 + wn0 + col;
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    if (n0 + wn0 + j * 32 + col < p.N) { const float v = alpha * acc[i][j][r]; row[j * 32] = v; vm = fmaxf(vm, fabsf(v)); }
            }
        }
    if (SCH == 1 && p.out_amax) {                // max |C| per outer batch index: one checked atomic per wave
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) vm = fmaxf(vm, __shfl_xor(vm, o));
        unsigned* slot_o = p.out_amax + (size_t)b1 * EGR_ROW_AMAX_STRIDE;
        const unsigned bits = __float_as_uint(vm);
        if (lane == 0 && bits > __hip_atomic_load(slot_o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot_o, bits);
    }
}

}  // namespace egr

static int bgemm_nt_launch(const float* a, const float* b, float* c, int nb1, int nb2, int M, int N, int K, int lda, int ldb, int ldc, int64_t sa1,
                           int64_t sa2, int64_t sb1, int64_t sb2, int64_t sc1, int64_t sc2, float alpha, const float* a_amax, const float* b_amax,
                           float* out_amax, void* stream) {
    EGR_CHECK(a && b && c && nb1 >= 1 && nb2 >= 1 && M >= 1 && N >= 1 && K >= 16, EGR_ERR_ARG, "bad gemm argument");
    EGR_CHECK((long long)nb1 * nb2 <= 65535, EGR_ERR_ARG, "too many batches");
    EGR_CHECK(K % 16 == 0 && lda % 4 == 0 && ldb % 4 == 0 && ((((uintptr_t)a) | ((uintptr_t)b)) & 15) == 0 && sa1 % 4 == 0 &&
                  sa2 % 4 == 0 && sb1 % 4 == 0 && sb2 % 4 == 0, EGR_ERR_UNSUPPORTED,
              "split batched GEMM needs K %% 16 == 0 and 16-byte aligned rows");
    const float* zeros = nullptr;
    { const int zrc = zero_page(&zeros); if (zrc) return zrc; }
    GemmS3P p;
    p.a = a; p.b = b; p.c = c; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.nb2 = nb2;
    p.sa1 = sa1; p.sa2 = sa2; p.sb1 = sb1; p.sb2 = sb2; p.sc1 = sc1; p.sc2 = sc2; p.alpha = alpha; p.zeros = zeros;
    p.a_amax = (const unsigned*)a_amax; p.b_amax = (const unsigned*)b_amax; p.out_amax = (unsigned*)out_amax;
    const int bn = N > 64 ? 128 : (N > 32 ? 64 : 32);
    dim3 grid((M + 127) / 128, (N + bn - 1) / bn, nb1 * nb2);
    hipStream_t st = (hipStream_t)stream;
    if (a_amax) {
        if (bn == 128) hipLaunchKernelGGL((k_bgemm_s3<128, 1>), grid, dim3(256), 0, st, p);
        else if (bn == 64) hipLaunchKernelGGL((k_bgemm_s3<64, 1>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((k_bgemm_s3<32, 1>), grid, dim3(256), 0, st, p);
    } else {
        if (bn == 128) hipLaunchKernelGGL((k_bgemm_s3<128>), grid, dim3(256), 0, st, p);
        else if (bn == 64) hipLaunchKernelGGL((k_bgemm_s3<64>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((k_bgemm_s3<32>), grid, dim3(256), 0, st, p);
    }
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}

// The same product on two fp16 terms per operand (both operands are activations: each is scaled per OUTER batch index b1 -- the batch
// row -- by the power of two derived from its own maximum, a_amax[b1] / b_amax[b1] in the row_amax layout); out_amax (optional):
// max |C| per b1.
extern "C" int egr_bgemm_nt_h2(const float* a, const float* b, float* c, int nb1, int nb2, int M, int N, int K, int lda, int ldb, int ldc,
                               int64_t sa1, int64_t sa2, int64_t sb1, int64_t sb2, int64_t sc1, int64_t sc2, float alpha, const float* a_amax,
                               const float* b_amax, float* out_amax, void* stream) {
    EGR_CHECK(a_amax && b_amax, EGR_ERR_ARG, "null operand maxima");
    return bgemm_nt_launch(a, b, c, nb1, nb2, M, N, K, lda, ldb, ldc, sa1, sa2, sb1, sb2, sc1, sc2, alpha, a_amax, b_amax, out_amax, stream);
}

extern "C" int egr_bgemm_nt_s3(const float* a, const float* b, float* c, int nb1, int nb2, int M, int N, int K, int lda, int ldb,
                               int ldc, int64_t sa1, int64_t sa2, int64_t sb1, int64_t sb2, int64_t sc1, int64_t sc2, float alpha,
                               void* stream) {
    return bgemm_nt_launch(a, b, c, nb1, nb2, M, N, K, lda, ldb, ldc, sa1, sa2, sb1, sb2, sc1, sc2, alpha, nullptr, nullptr, nullptr, stream);
}
