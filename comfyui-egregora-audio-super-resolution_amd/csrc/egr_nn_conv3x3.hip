// Input-stationary 3x3 convolutions of the two-term fp16 operand scheme (scheme 1 of egr_nn_gemm_s3.hip) -- a translation unit of
// their own so that the kernels can be rebuilt (and built in variants, tools/build_variant.sh) without the implicit-GEMM family.
#include <stdlib.h>

#include <type_traits>

#include "egr_conv.h"
#include "egr_s3_split.h"

// ------------------------------------------------------------------------------------------------------------------
// Stride-1 pad-1 3x3 convolution, input-stationary in two dimensions (scheme 1 only): the 128-channel level of the VAE
// (512 x 256 images) is HBM-bound as an F(4x4) pipeline -- V and M, 2.25x the tensor each, are written and read back: ~19 GB per
// layer against 3.5 GB of activations -- and as an implicit GEMM every activation is loaded and split nine times.  Here a workgroup
// owns 4 image rows x 32 pixels (the 128 GEMM rows of its tile = four 32-pixel MFMA sub-tiles, one image row each), splits the
// (4 + 2) x (32 + 2) halo patch of a 32-channel chunk ONCE into LDS -- with the producer's GroupNorm (+ SiLU) applied on the way
// (gn_scale / gn_shift per (image, channel); zero padding after it, as the reference pads the normalised tensor) -- and the nine
// taps read their operands from it at row offsets (ky * 34 + kx).  x is read once (1.6x from L2), y written once; weights stream
// through double-buffered LDS tiles exactly as in k_conv1d_s3.  With the GroupNorm fused the operand scale comes from a BOUND of
// the normalised row (row_amax holds max |gn_scale| * max |x| + max |gn_shift|, csrc/egr_nn_ops.hip k_gn_bound): the scale only has
// to put the row's maximum somewhere in [1, 2^15), and the bound is within a few bits of the true maximum.
namespace egr {

// ---- k_conv3x3_isp: the halo patch of channel chunk c + 1 is prepared WHILE chunk c multiplies ----
// Round 4's k_conv3x3_is (removed in round 6; profiles/r05/flashsr_kernel_experiments.log item 2 has its ablation) stopped its matrix pipe
// at every chunk boundary: the global loads of the 6 x 34 patch (an HBM / MALL round trip), the
// fused GroupNorm + SiLU (an IEEE division per element), the operand split and the LDS stores all sat between two barriers -- about
// a third of a workgroup's time, with nothing to cover it but the second workgroup of the CU.  Here the patch is DOUBLE-buffered in
// LDS (2 x 26 KB) and every
// step of the phase is a compile-time position in the slab sequence of the PREVIOUS chunk.  A chunk is 18 slabs (9 taps x two
// 16-k halves: an even number, so the weight register buffer of every slab is a compile-time constant and each slab
// is ONE basic block):
//   slab 0              the patch of chunk c + 1 is requested (8 float4 per thread);
//   slabs 4, 6, 8, 10   one of the thread's four patch items per slab is normalised, activated (x * rcp(1 + exp(-x)): v_rcp_f32,
//                       1 ulp -- the operand keeps 22 bits), split into its two fp16 terms and stored into the OTHER patch buffer:
//                       ~70 VALU instructions and two ds_write_b128 next to the slab's 12 MFMAs, in the matrix pipe's shadow;
//   the barrier that opens chunk c + 1 publishes the patch -- nothing of the phase is left between barriers.
#define C3P_MAX_CIN 512                                      // channels whose GroupNorm coefficients the kernel stages in LDS
// Round 6: the weight fragments go from L2 / L1 STRAIGHT into the MFMA operand registers -- lane (n = li, k-half = lk) of the operand
// layout owns 16 contiguous bytes of the [slab][plane][Cout][16] pack, a wave-load is 1 KB of consecutive addresses -- one slab ahead, in a
// register double buffer of the size the LDS staging registers of round 5 had.  No weight tiles in LDS (57 KB instead of 74 KB), half the
// operand ds_read_b128 per MFMA, and ONE workgroup barrier per channel chunk (the patch hand-over) instead of one per slab: 4 instead of
// 72.  3.52 -> 3.39 ms per 26-row layer stand-alone, same bits (profiles/r06/conv3x3_weights_in_registers.txt); the kernel runs at the
// board's power limit like the GEMM, so what it bought is the energy of the LDS traffic it no longer makes.
template <int BN, int CC, bool GN, bool SILU>
__global__ __launch_bounds__(256, 2) void k_conv3x3_isp(ConvP p) {
    typedef S3Cfg<128, BN> TC;
    constexpr int TM = TC::TM, TN = TC::TN, NCH = CC / 8, NSL = CC / 16, PW = 34, PR = 6, RMAX = PR * PW;
    constexpr int NP = 2, NIT = (RMAX * NCH + 255) / 256, SPC = 9 * NSL;      // SPC: slabs per chunk
    static_assert(CC == 32 && NIT == 4 && 256 % NCH == 0 && SPC % 2 == 0, "four patch items per thread, all of one 8-channel group");
    __shared__ uint4 As[2][NP][RMAX * NCH + 8];           // (+ 8 dump slots: threads without a fourth patch item store there, branch-free)
    __shared__ float os_tab[128];
    __shared__ unsigned om_tab[1];
    // GroupNorm scale / shift of this image's channels, staged once: the conversion slabs read them with ds_read (lgkmcnt) -- a global
    // load there would make the slab wait vmcnt(0), i.e. for the weight tiles it has just requested (one L2 round trip per slab)
    __shared__ float4 gn_tab[GN ? 2 * (C3P_MAX_CIN / 4) : 1];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm0 = (wave / TC::WN) * (128 / TC::WM), wn0 = (wave % TC::WN) * (BN / TC::WN);
    const int tiles_x = p.W / 32;
    const int b = blockIdx.z, ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x, n0 = blockIdx.y * BN;
    const int y0 = ty * 4, x0 = tx * 32;
    const float* xb = p.x + (size_t)b * p.H * p.W * p.Cin;
    const unsigned abits = p.row_amax[(size_t)b * EGR_ROW_AMAX_STRIDE];
    const float a_scale = h2_row_scale(abits);
    if (tid < 128) os_tab[tid] = h2_row_inv(abits);
    if (tid == 0) om_tab[0] = 0u;
    if (GN) {
        for (int i = tid; i < p.Cin / 4; i += 256) {
            gn_tab[i] = ((const float4*)(p.gn_scale + (size_t)b * p.Cin))[i];
            gn_tab[C3P_MAX_CIN / 4 + i] = ((const float4*)(p.gn_shift + (size_t)b * p.Cin))[i];
        }
        __syncthreads();
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const size_t b_slab = (size_t)p.Cout * 2 * NP;     // 16-byte chunks of one 16-k slab of the weight pack
    const int cpt = p.Cin / 16;                  // slabs per tap in the weight pack
    const uint4* zq = (const uint4*)p.zeros;
    const int nchunks = p.Cin / CC;
    const int li = lane & 31, lk = lane >> 5;
    // this lane's fragment of (plane q, column sub-tile j) of a slab sits at base[bw_off + (q Cout + 32 j) 2]; the tile of slab s of chunk cc
    // (s may run past the chunk: the next chunk's first slab; past the end: the last tile again -- no run-time branch inside a slab)
    uint4 bwA[TN][NP], bwB[TN][NP];
    const size_t bw_off = ((size_t)n0 + wn0 + li) * 2 + lk;
    auto load_bw = [&](int cc, int s, uint4 (&dst)[TN][NP]) {
        if (s >= SPC) { s -= SPC; ++cc; }
        if (cc >= nchunks) { cc = nchunks - 1; s = SPC - 1; }
        const int tap = s / NSL, cs = s - tap * NSL;
        const uint4* base = p.w3 + (size_t)(tap * cpt + cc * NSL + cs) * b_slab + bw_off;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < NP; ++q) dst[j][q] = (n0 + wn0 + j * 32 + li < p.Cout) ? base[((size_t)q * p.Cout + j * 32) * 2] : zq[0];
    };

    // ---- the thread's four patch items: the same pixels for every channel chunk ----
    const int hch = tid & (NCH - 1);
    int hoff0, hoff1, hoff2, hoff3, hslot0, hslot1, hslot2, hslot3;
    bool hok0, hok1, hok2, hok3;
#define C3_HSETUP(I, OFF, SLOT, OK)                                                                                  \
    {                                                                                                                \
        const int e = tid + 256 * (I);                                                                               \
        const int r = e / NCH, pr = r / PW, pc = r - pr * PW;                                                        \
        const int iy = y0 - 1 + pr, ix = x0 - 1 + pc;                                                                \
        OK = e < RMAX * NCH && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;                        \
        OFF = OK ? (iy * p.W + ix) * p.Cin + hch * 8 : hch * 8;                                                      \
        SLOT = e < RMAX * NCH ? r * NCH + (hch ^ ((r / (16 / NCH)) & (NCH - 1))) : RMAX * NCH + (tid & 7);                \
    }
    C3_HSETUP(0, hoff0, hslot0, hok0)
    C3_HSETUP(1, hoff1, hslot1, hok1)
    C3_HSETUP(2, hoff2, hslot2, hok2)
    C3_HSETUP(3, hoff3, hslot3, hok3)
#undef C3_HSETUP
    float4 hu0, hv0, hu1, hv1, hu2, hv2, hu3, hv3;          // the floats of the patch items between request and conversion
    auto halo_issue = [&](int cc_in) {
        const int c0 = min(cc_in, nchunks - 1) * CC;
        const float* s0 = xb + hoff0 + c0; const float* s1 = xb + hoff1 + c0; const float* s2 = xb + hoff2 + c0; const float* s3 = xb + hoff3 + c0;
        hu0 = *(const float4*)s0; hv0 = *(const float4*)(s0 + 4);
        hu1 = *(const float4*)s1; hv1 = *(const float4*)(s1 + 4);
        hu2 = *(const float4*)s2; hv2 = *(const float4*)(s2 + 4);
        hu3 = *(const float4*)s3; hv3 = *(const float4*)(s3 + 4);
    };
    // GroupNorm + SiLU + zero padding + split of one item, straight into patch buffer `abuf` (a thread without the item writes a dump slot)
    auto halo_finish = [&](const float4& u, const float4& v, bool ok, int slot, int cc_in, int abuf) {
        float x[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
        if (GN) {
            const int q0 = (min(cc_in, nchunks - 1) * CC + hch * 8) >> 2;
            const float4 gsa = gn_tab[q0], gsb = gn_tab[q0 + 1], gha = gn_tab[C3P_MAX_CIN / 4 + q0], ghb = gn_tab[C3P_MAX_CIN / 4 + q0 + 1];
            const float sc[8] = {gsa.x, gsa.y, gsa.z, gsa.w, gsb.x, gsb.y, gsb.z, gsb.w};
            const float sh[8] = {gha.x, gha.y, gha.z, gha.w, ghb.x, ghb.y, ghb.z, ghb.w};
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                x[q] = fmaf(x[q], sc[q], sh[q]);
                if (SILU) x[q] = x[q] * __builtin_amdgcn_rcpf(1.f + __expf(-x[q]));
            }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) x[q] = ok ? x[q] : 0.f;  // zero padding applies AFTER the normalisation
        uint4 q[3];
        split_x8<1>(make_float4(x[0], x[1], x[2], x[3]), make_float4(x[4], x[5], x[6], x[7]), a_scale, q);
        As[abuf][0][slot] = q[0];
        As[abuf][1][slot] = q[1];
    };

    // slab S (compile-time) of chunk cc: tap S / 2, 16-k half S % 2; multiplies the weight fragments of register buffer S & 1 with patch
    // buffer cc & 1 and requests the fragments of slab S + 1 into the other register buffer
    auto slab = [&](auto s_c, int cc) {
        constexpr int S = decltype(s_c)::value, cur = S & 1, TAP = S / NSL, CS = S % NSL;
        constexpr int ky = TAP / 3, kx = TAP - ky * 3;
        const int ab = cc & 1;
        if (S == 0) __syncthreads();             // the patch of this chunk is visible, every wave is done with the previous chunk's
        uint4 bq[TN][2], aq[TM][2];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < NP; ++q) bq[j][q] = cur ? bwB[j][q] : bwA[j][q];
        const int ch = CS * 2 + lk;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int r = ((wm0 >> 5) + i + ky) * PW + li + kx;       // sub-tile (wm0 / 32 + i) = image row y0 + that
            const int slot = r * NCH + (ch ^ ((r / (16 / NCH)) & (NCH - 1)));
#pragma unroll
            for (int q = 0; q < NP; ++q) aq[i][q] = As[ab][q][slot];
        }
        if (cur) load_bw(cc, S + 1, bwA); else load_bw(cc, S + 1, bwB);          // the fragments of the NEXT slab: a whole slab of MFMAs to land
        if (S == 0) halo_issue(cc + 1);
        if (S == 4) halo_finish(hu0, hv0, hok0, hslot0, cc + 1, ab ^ 1);
        if (S == 6) halo_finish(hu1, hv1, hok1, hslot1, cc + 1, ab ^ 1);
        if (S == 8) halo_finish(hu2, hv2, hok2, hslot2, cc + 1, ab ^ 1);
        if (S == 10) halo_finish(hu3, hv3, hok3, hslot3, cc + 1, ab ^ 1);
        // weights as the first operand: transposed accumulators, 16-byte stores (conv_epilogue_t)
#define C3_MMA(QA, QB)                                                                                                   \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) acc[i][j] =            \
        __builtin_amdgcn_mfma_f32_32x32x16_f16(as_hf(bq[j][QB]), as_hf(aq[i][QA]), acc[i][j], 0, 0, 0);
        C3_MMA(1, 0)
        C3_MMA(0, 1)
        C3_MMA(0, 0)
#undef C3_MMA
        if (S == 4 || S == 6 || S == 8 || S == 10) {
            // the item's ~70 VALU / transcendental instructions go BETWEEN the slab's MFMAs, six per MFMA (the machine scheduler
            // otherwise issues them as one block in front: ~350 cycles in which this wave feeds nothing to the matrix pipe)
#pragma unroll
            for (int k = 0; k < TM * TN * 3; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x402, 6, 0);
            }
        }
    };
    auto chunk = [&](int cc) {
#define C3_S(N) slab(std::integral_constant<int, N>(), cc);
        C3_S(0) C3_S(1) C3_S(2) C3_S(3) C3_S(4) C3_S(5) C3_S(6) C3_S(7) C3_S(8)
        C3_S(9) C3_S(10) C3_S(11) C3_S(12) C3_S(13) C3_S(14) C3_S(15) C3_S(16) C3_S(17)
#undef C3_S
    };
    static_assert(SPC == 18, "chunk() lists the 18 slabs of a chunk");

    // prologue: the patch of chunk 0 and the first weight tiles
    halo_issue(0);
    load_bw(0, 0, bwA);
    halo_finish(hu0, hv0, hok0, hslot0, 0, 0);
    halo_finish(hu1, hv1, hok1, hslot1, 0, 0);
    halo_finish(hu2, hv2, hok2, hslot2, 0, 0);
    halo_finish(hu3, hv3, hok3, hslot3, 0, 0);
    for (int cc = 0; cc < nchunks; ++cc) chunk(cc);
    unsigned* const om = p.out_amax ? om_tab : nullptr;
    conv_epilogue_t<TM, TN>(p, acc, (b * p.H + y0) * p.W + x0, n0, wm0, wn0, os_tab, om, p.W, p.gn_part);
    if (om) out_amax_commit(p, om_tab, (b * p.H + y0) * p.W + x0, 1);
}

// the instantiation launch_conv3x3_is picks for (Cout, fused GroupNorm, SiLU): what the per-kernel profile and profiles/traffic.json key on
const char* conv3x3_is_name(int cout, bool gn, bool silu) {
    const bool wide = cout > 64;
    if (gn && silu) return wide ? "k_conv3x3_isp<128, 32, true, true>" : "k_conv3x3_isp<64, 32, true, true>";
    if (gn) return wide ? "k_conv3x3_isp<128, 32, true, false>" : "k_conv3x3_isp<64, 32, true, false>";
    return wide ? "k_conv3x3_isp<128, 32, false, false>" : "k_conv3x3_isp<64, 32, false, false>";
}

// true when the input-stationary 3x3 kernel applies (scheme 1, big images, Cin <= 512, one image < 2^31 elements); launches it
bool launch_conv3x3_is(const ConvP& p, hipStream_t st) {
    static const bool off = getenv("EGR_S3_CONV3X3") && atoi(getenv("EGR_S3_CONV3X3")) == 0;
    if (off || !p.sch || !p.w3 || p.KH != 3 || p.KW != 3 || p.stride != 1 || p.dil != 1 || p.pad_t != 1 || p.pad_l != 1 || p.up2 ||
        p.OH != p.H || p.OW != p.W || (p.W % 32) != 0 || (p.H % 4) != 0 || (p.Cin % 32) != 0 || (p.Cout % 4) != 0 || p.ksplit > 1 ||
        p.zs_nzb > 0 || p.nz > 1 || p.osy != 1 || p.osx != 1 || p.OHF != p.OH || p.OWF != p.OW || p.bias_b || p.B > 65535 ||
        p.rows_div != p.H * p.W || (long long)(p.H / 4) * (p.W / 32) * p.B < 512 ||
        (size_t)p.H * p.W * p.Cin >= ((size_t)1 << 31) || p.Cin > C3P_MAX_CIN)
        return false;
    const int bn = p.Cout > 64 ? 128 : 64;
    const dim3 grid((p.H / 4) * (p.W / 32), (p.Cout + bn - 1) / bn, p.B);
    if (p.gn_scale && p.gn_silu) {
        if (bn == 128) hipLaunchKernelGGL((k_conv3x3_isp<128, 32, true, true>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((k_conv3x3_isp<64, 32, true, true>), grid, dim3(256), 0, st, p);
    } else if (p.gn_scale) {
        if (bn == 128) hipLaunchKernelGGL((k_conv3x3_isp<128, 32, true, false>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((k_conv3x3_isp<64, 32, true, false>), grid, dim3(256), 0, st, p);
    } else {
        if (bn == 128) hipLaunchKernelGGL((k_conv3x3_isp<128, 32, false, false>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((k_conv3x3_isp<64, 32, false, false>), grid, dim3(256), 0, st, p);
    }
    return true;
}

}  // namespace egr


