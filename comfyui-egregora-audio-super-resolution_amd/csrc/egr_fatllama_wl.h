// The iteration loop of upstream's feed.upscale as the reference calls it (egregora_fat_llama_gpu.py:213-224,
// egregora_fat_llama_cpu.py:126-134; algorithm: SPEC.md section 3, oracle/fatllama.py) --
// loop kernels of the hot Fat-Llama plan (rows of 2304 points, columns of 625 points) with TWO workgroup barriers each instead
// of one or two per radix stage (k_row<false, 1>: 15, k_col<1, 2>: 10 -- 54-67 % of their wave cycles were spent parked at them,
// profiles/r02/chain60_counters.txt).  Included by egr_fatllama.hip; same state layout, same arithmetic per element (tables rounded
// once from long double, pair hook and 1/M in double), different factorisation order, so results agree with the stage-by-stage
// kernels to float32 round-off, not bit for bit.
//
// k_row_wl<N1, Q> -- a row of L = N1 Q^2 points (14 lengths 384 ... 4608; below: N1 = 16, Q = 12, L = 2304) as a four-step transform
// INSIDE the workgroup:
//   n = 144 n1 + n2, k = k1 + 16 k2:  X[k1 + 16 k2] = sum_n2 W_144^(n2 k2) [ W_L^(n2 k1) sum_n1 x[144 n1 + n2] W_16^(n1 k1) ]
//   "cross" step  (one thread per n2): 16 coalesced global loads -> radix-16 butterfly in registers -> x W_L^(n2 k1) -> LDS block k1
//                 (the state never sits in LDS in natural order: the load IS the first stage, the store IS the last one)
//   "local" step  (12 lanes of ONE wave per block): 144 = 12 x 12 -- radix-12 over a (n2 = 12 a + b), x W_144^(b c), a 12 x 12
//                 transpose through the block's own LDS area (wave-local: program order of one wave's DS instructions, no
//                 s_barrier), radix-12 over b: lane c ends with X[k1 + 16 (c + 12 d)], d = 0..11, in registers.
//   The real-split partner of element k of row a is element L-1-k of row b = (block 15-k1, lane 11-c, register 11-d): the 12
//   lanes that transform block k1 of row a also transform block 15-k1 of row b with the lane order reversed, so BOTH members of
//   every (k, M-k) pair are registers of one thread and the hook needs no exchange at all.  The inverse runs the same steps
//   backwards (conjugate twiddles): local, barrier, cross + coalesced global store.
//   Barriers: after the cross-forward LDS writes, before the cross-inverse LDS reads.  LDS passes per element: 8 (was 16).
//   A self-paired row (o = 0; o = R/2 for even R) takes its pairs from LDS with the generic loop of k_row (two more barriers).
//
// k_col_wl<R, TC> -- a tile of TC adjacent columns of L = R x R points (below: 625 = 25 x 25, TC = 8; also 441 = 21 x 21, TC = 12), one thread per (b, column):
//   i = 25 a + b, n = c + 25 d:  thread (b, col) loads its 25 elements i = 25 a + b straight from global memory (64-byte row
//   segments per 8 lanes, the same segments the staged tile load touched), x conj W_M^(col i) (a geometric run in double: two
//   table products per thread, one double multiply per element), inverse radix-25 over a, LDS transpose (barrier), x conj
//   W_625^(b c), inverse radix-25 over b: thread c holds the time-domain points n = c + 25 d -- which are exactly the inputs
//   n = 25 a' + b' (b' = c) its forward radix-25 over a' wants: no exchange between the inverse and the forward transform.
//   x W_625^(b' c''), written TRANSPOSED into the row of the matrix the thread itself has just read (no write-after-read
//   hazard against other threads), barrier, forward radix-25 over b', x W_M^(col i), store to the addresses it loaded from.
//   Barriers: 2 (+1 for the LDS copy of the 25 x 25 stage table, shared with the first).  LDS passes per element: 4 (was 10).
#pragma once

namespace egr {

#ifndef EGR_WL_COL_WAVES
#define EGR_WL_COL_WAVES 1
#endif
// Precision switches of the two-barrier kernels (bit mask; round 6, profiles/r06/c3_error_attribution.txt):
//   1  hook: a pair whose two bins both survive the threshold leaves as Z / M -- the real split followed by its inverse is the identity
//   2  k_col_wl: the four-step twiddles W_M^(col i) as two floats      4  W_(R^2)^(b c) as two floats
//   8  k_row_wl: W_L^(n2 k1) as two floats                             16 W_(Q^2)^(b c) as two floats      32 the pair twiddle W_N^k as two floats
//   64 k_row_wl: W_L^(n2 k1) as a geometric run in double from one table entry, products in double (instead of 8)
//   128 k_row_wl: W_(Q^2)^(b c) likewise (instead of 16)               256 k_col_wl: the four-step twiddle products in double (instead of 2)
// Shipped: 2 + 64 + 128 (and EGR_BFLY_HILO = 15 in this translation unit).  N = 2 880 000, 800 iterations, against the float64 loop, in units
// of the float32 pocketfft oracle's own error (profiles/r06/c3_error_attribution_pass{1,2,3}.txt): none 2.27x rms / 1.51x max / plain LSD
// 2.2e-3 dB at 28.1 ms per stereo stage; shipped 0.86x / 0.59x / 7.4e-4 dB at 29.8 ms.  Alone: bit 2 1.87x (+0.6 ms), 64 2.00x (+0.1; as two
// floats from a table, bit 8, +5 ms), 128 2.06x (+0.9), butterfly constants 1.71x (+2), 4 / 32 / 1 within 3 % of none; everything 0.74x at 42 ms.
#ifndef EGR_WL_HILO
#define EGR_WL_HILO 194
#endif
// Geometry of k_col_wl<R, TC>: columns of R x R points (625 = 25^2: lengths with the factor 5^4 -- every multiple of 5 s at 48 kHz;
// 441 = 21^2: lengths with the factor 3^2 7^2 -- whole seconds at 44.1 kHz), TC adjacent columns per workgroup, R TC threads on
// four waves.  For R = 25, TC = 8 (200 of 256 lanes) is what ships.  Measured against it on one box (C3 stage,
// profiles/r04/fatllama_c3_experiments.log): TC = 12 (300 threads on FIVE waves: two waves of one workgroup share a SIMD) +1.1 ms;
// TC = 10 (250 of 256 lanes, ragged last tile, 80-byte row segments) +0.6 ms -- the loop is bound by the latency chain of a
// workgroup and by how the two channel pipelines interleave, not by lane fill.  R = 21: TC = 12 (252 threads).
template <int R, int TC> struct WlCol {
    static constexpr int THREADS = ((R * TC + 63) / 64) * 64;
    static constexpr int LDS = R * R * TC * 8;
};
static inline int wl_col_radix(int L) { return L == 625 ? 25 : (L == 441 ? 21 : 0); }      // column lengths with a k_col_wl instantiation
// The factors 1/2 of the real split are folded away: the hook works on 2 X (exact: a power of two commutes with every rounding),
// compares against 4 thr^2 / 2 t, and the 1/4 this leaves joins 1/M.  The scaling by 1/M is a product with a two-term float value
// (sc_hi + sc_lo = 1/(4M) to 2^-48: one rounding per result, as the product in double it replaces) -- a float 1/M alone carries a
// relative error of up to 6e-8 that every iteration would apply again, in the same direction.
struct WlScale { float hi, lo; };
__device__ __forceinline__ WlScale wl_scale(const double scd) {
    WlScale s;
    const double q = 0.25 * scd;
    s.hi = (float)q;
    s.lo = (float)(q - (double)s.hi);
    return s;
}
__device__ __forceinline__ float wl_scaled(const float v, const WlScale s) { return fmaf(v, s.hi, v * s.lo); }
template <int HOOK>
__device__ __forceinline__ float wl_pair_hook(const cplx Za, const cplx Zb, const dcplx Wkd, const float thr2x4, const float tlevx2, const int soft,
                                              const WlScale sc, cplx& na, cplx& nb) {
    const cplx Wk = make_float2((float)Wkd.x, (float)Wkd.y);
    const cplx Wl = make_float2((float)(Wkd.x - (double)Wk.x), (float)(Wkd.y - (double)Wk.y));
    const cplx E = make_float2(Za.x + Zb.x, Za.y - Zb.y);                   // 2 E
    const cplx O = make_float2(Za.y + Zb.y, Zb.x - Za.x);                   // 2 O
    const cplx WO = (EGR_WL_HILO & 32) ? cmul2(O, Wk, Wl) : cmul(Wk, O);
    cplx Xk = cadd(E, WO), Xm = csub(E, WO);                                // 2 X[k], 2 X[M - k]
    float mxn = 0.f;
    if (HOOK == 2) return 0.25f * fmaxf(Xk.x * Xk.x + Xk.y * Xk.y, Xm.x * Xm.x + Xm.y * Xm.y);
    if (HOOK == 1) {
        // the comparison on squares as in the default hook; the soft gain 1 - t / |X| through v_rsq_f32 (1 ulp: 6e-8 of a gain <= 1) --
        // round 5's two correctly rounded sqrtf and two divisions per pair were a tenth of the kernel
        // (hook 1 costs the C3 stage 2.6 ms over hook 0 with the two channel pipelines overlapped and 1.1 ms with one pipeline, whatever the
        // threshold gates and whether soft / hard is a run-time branch or an instantiation of its own (tried: hook 3): not this arithmetic --
        // tools/r06_hook_times.py, profiles/r06/relative_threshold_timing.txt)
        const float mk2 = Xk.x * Xk.x + Xk.y * Xk.y, mm2 = Xm.x * Xm.x + Xm.y * Xm.y;
        const bool kk = mk2 > thr2x4, km = mm2 > thr2x4;
        if (!soft) {
            if (!kk) Xk = make_float2(0.f, 0.f);
            if (!km) Xm = make_float2(0.f, 0.f);
            mxn = 0.25f * fmaxf(kk ? mk2 : 0.f, km ? mm2 : 0.f);                          // max |S(X)|^2: the next iteration's spectrum
        } else {
            const float gk = kk ? 1.f - tlevx2 * __builtin_amdgcn_rsqf(mk2) : 0.f, gm = km ? 1.f - tlevx2 * __builtin_amdgcn_rsqf(mm2) : 0.f;
            Xk.x *= gk; Xk.y *= gk; Xm.x *= gm; Xm.y *= gm;
            mxn = 0.25f * fmaxf(gk * gk * mk2, gm * gm * mm2);                            // |g X|^2 = g^2 |X|^2
        }
    } else {
        if (!(Xk.x * Xk.x + Xk.y * Xk.y > thr2x4)) Xk = make_float2(0.f, 0.f);
        if (!(Xm.x * Xm.x + Xm.y * Xm.y > thr2x4)) Xm = make_float2(0.f, 0.f);
    }
    const cplx E2 = cadd(Xk, Xm);                                           // 4 E'
    const cplx H = csub(Xk, Xm);
    const cplx O2 = (EGR_WL_HILO & 32) ? cmulc2(H, Wk, Wl) : cmulc(H, Wk);   // 4 O'
    na = make_float2(wl_scaled(E2.x - O2.y, sc), wl_scaled(E2.y + O2.x, sc));
    nb = make_float2(wl_scaled(E2.x + O2.y, sc), wl_scaled(O2.x - E2.y, sc));
    if ((EGR_WL_HILO & 1) && HOOK == 0) {
        // both bins kept: split . un-split is the identity (E2 + i O2 = 4 Za exactly in exact arithmetic), so the pair leaves as Z / M
        // with the scale's one rounding instead of the eight of the detour
        const bool both = (Xk.x != 0.f || Xk.y != 0.f) && (Xm.x != 0.f || Xm.y != 0.f);
        if (both) {
            na = make_float2(wl_scaled(4.f * Za.x, sc), wl_scaled(4.f * Za.y, sc));
            nb = make_float2(wl_scaled(4.f * Zb.x, sc), wl_scaled(4.f * Zb.y, sc));
        }
    }
    return mxn;
}

// Geometry of k_row_wl<N1, Q>: rows of L = N1 Q^2 points (N1: cross radix, even or odd; Q x Q blocks on Q lanes of one wave).
template <int N1, int Q> struct WlRow {
    static constexpr int QQ = Q * Q, L = N1 * QQ;
    static constexpr int TS = Q + 2;                                   // row stride of the Q x Q transpose (even: 16-byte accesses)
    // LDS block stride in elements (>= QQ and >= Q TS).  Q = 10: six 80-byte lane groups per wave access -- a stride of 138 elements
    // (1104 bytes = 80 mod 256) lays them end to end over the banks (two-way, the minimum) where 124 stacked three of them:
    // 60 s at 44.1 kHz 29.9 -> 28.7 ms (profiles/r04/fatllama_c3_experiments.log)
    static constexpr int S = Q == 10 ? 138 : Q * TS + 4;
    static constexpr int RS = N1 * S;
    static constexpr int LWAVES = (N1 + 64 / Q - 1) / (64 / Q);       // waves of the local step
    static constexpr int UPW = (N1 + LWAVES - 1) / LWAVES;             // units (block pairs) per wave, balanced
    static constexpr int THREADS = 64 * (LWAVES < 2 ? 2 : LWAVES);
    static constexpr int XIT = (2 * QQ + THREADS - 1) / THREADS;       // rounds of the 2 Q^2 cross butterflies
    static constexpr int LDS = (2 * RS + (N1 % 2 ? S : 0)) * 8;         // odd N1: a spare block (the middle unit of a self-paired row)
};
// the N1 twiddles W_L^(n2 k1) of one cross butterfly (row n2 of the [Q^2][N1] table): 16-byte loads where the row is aligned
template <int N1>
__device__ __forceinline__ void wl_load_tw(const cplx* __restrict__ row, cplx (&w)[N1]) {
    if constexpr (N1 % 2 == 0) {
        const float4* tp = (const float4*)row;
#pragma unroll
        for (int q = 0; q < N1 / 2; ++q) {
            const float4 t = tp[q];
            w[2 * q] = make_float2(t.x, t.y);
            w[2 * q + 1] = make_float2(t.z, t.w);
        }
    } else {
#pragma unroll
        for (int k = 0; k < N1; ++k) w[k] = row[k];
    }
}

template <int N1, int Q, int HOOK>
__global__ __launch_bounds__((WlRow<N1, Q>::THREADS)) void k_row_wl(RowP p, WlRowT tb, long long M, cplx* __restrict__ work) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    EGR_LDS_CANARY_ARM(smem);
    using G = WlRow<N1, Q>;
    constexpr int L = G::L, QQ = G::QQ, S = G::S, TS = G::TS, RS = G::RS, THREADS = G::THREADS, XIT = G::XIT, UPW = G::UPW;
    static_assert(S >= QQ && S % 2 == 0, "k_row_wl geometry");
    const int R = p.R;
    const int oa = blockIdx.x;
    const int ob = (R - oa) % R;
    const bool self = (oa == ob);
    const int ch = blockIdx.y;
    cplx* lds = (cplx*)EGR_LDS_BASE(smem);
    cplx* W = work + (size_t)ch * M;
    const int ra = (oa % p.Ma) * p.Mb + oa / p.Ma;
    const int rbw = (ob % p.Ma) * p.Mb + ob / p.Ma;
    cplx* ga = W + (size_t)ra * L;
    cplx* gb = W + (size_t)rbw * L;
    const int tid = threadIdx.x;
    // ---- roles
    const int xlim = self ? QQ : 2 * QQ;
    const int wv = tid >> 6, lane = tid & 63, un = lane / Q, l = lane - Q * un;
    const int k1 = UPW * wv + un;                                          // local step: block k1 of row a, block N1 - 1 - k1 of row b
    const bool lact = un < UPW && k1 < N1 && !(self && k1 >= (N1 + 1) / 2);      // a self-paired row: blocks k1 and N1 - 1 - k1 of the SAME row
    cplx* ba = lds + (k1 < N1 ? k1 : 0) * S;
    cplx* bb = lds + (self ? 0 : RS) + (k1 < N1 ? N1 - 1 - k1 : 0) * S;
    // odd N1, self-paired row: the middle block is its own partner -- the unit transforms it once (as A); its B half works on a spare block
    if ((N1 & 1) && self && 2 * k1 == N1 - 1) bb = lds + 2 * RS;
    // the pair twiddles W_N^(o + R k), k = k1 + N1 l + N1 Q d: a geometric run in double over d (ratio W_(2L)^(N1 Q) = W_(2Q))
    dcplx wrun = make_double2(1.0, 0.0);
    if (lact) wrun = dcmul(tw2d(p.wo, (unsigned)oa), p.wk[k1 + N1 * l]);
    // the carried spectrum maximum is requested HERE: the slots were written by atomics of the previous iteration's launch (memory side), so
    // the read is a miss of this XCD's L2 -- a microsecond that the forward transform below covers; behind the first barrier it was on every
    // workgroup's critical path
    float max2v = 0.f;
    if (HOOK == 1 && p.max2) max2v = fl_max2_read(p.max2, ch);
    EGR_STAMP(p, 0);

    // ---- cross step, forward: global -> radix N1 -> twiddle -> LDS blocks
#pragma unroll
    for (int xi = 0; xi < XIT; ++xi) {          // cross step: row xr, n2 = xn2 (2 Q^2 butterflies over the workgroup's threads)
        const int xt = tid + xi * THREADS;
        if (xt >= xlim) break;
        const int xr = xt >= QQ ? 1 : 0, xn2 = xt - QQ * xr;
        const cplx* g = xr ? gb : ga;
        cplx v[N1];
#pragma unroll
        for (int n1 = 0; n1 < N1; ++n1) v[n1] = g[n1 * QQ + xn2];
        cplx w[N1];
        if (!(EGR_WL_HILO & 64)) wl_load_tw<N1>(tb.t1 + xn2 * N1, w);
        Bfly<N1>::run(v);
        cplx* d = lds + xr * RS + xn2;
        d[0] = v[0];
        if (EGR_WL_HILO & 64) {                 // W_L^(n2 k1) as a geometric run in double from ONE table entry, the product in double
            const dcplx st = p.twd[xn2];
            dcplx cur = st;
#pragma unroll
            for (int k = 1; k < N1; ++k) {
                const double vx = (double)v[k].x, vy = (double)v[k].y;
                d[k * S] = make_float2((float)(vx * cur.x - vy * cur.y), (float)(vx * cur.y + vy * cur.x));
                cur = dcmul(cur, st);
            }
        } else if (EGR_WL_HILO & 8) {
            cplx wl[N1];
            wl_load_tw<N1>(tb.t1l + xn2 * N1, wl);
#pragma unroll
            for (int k = 1; k < N1; ++k) d[k * S] = cmul2(v[k], w[k], wl[k]);
        } else
#pragma unroll
        for (int k = 1; k < N1; ++k) d[k * S] = cmul(v[k], w[k]);
    }
    __syncthreads();
    EGR_STAMP(p, 1);

    cplx A[Q], B[Q];
    if (lact) {
        // ---- local step, forward: Q^2 = Q x Q per block, wave-local transposes
        const cplx* t2 = tb.t2 + l * Q;
        const cplx* t2l = tb.t2l + l * Q;
#pragma unroll
        for (int a = 0; a < Q; ++a) { A[a] = ba[Q * a + l]; B[a] = bb[Q * a + l]; }
        wl_wave_sync();
        Bfly<Q>::run(A);
        Bfly<Q>::run(B);
        dcplx t2cur = make_double2(1.0, 0.0);
        const dcplx t2st = (EGR_WL_HILO & 128) ? tb.t2d[l] : t2cur;        // W_(Q^2)^l: the run over c
#pragma unroll
        for (int c = 0; c < Q; ++c) {
            if (EGR_WL_HILO & 128) {
                if (c) {
                    const double ax = (double)A[c].x, ay = (double)A[c].y, bx = (double)B[c].x, by = (double)B[c].y;
                    ba[c * TS + l] = make_float2((float)(ax * t2cur.x - ay * t2cur.y), (float)(ax * t2cur.y + ay * t2cur.x));
                    bb[(Q - 1 - c) * TS + l] = make_float2((float)(bx * t2cur.x - by * t2cur.y), (float)(bx * t2cur.y + by * t2cur.x));
                } else { ba[l] = A[0]; bb[(Q - 1) * TS + l] = B[0]; }
                t2cur = dcmul(t2cur, t2st);
                continue;
            }
            const cplx w = t2[c];
            if (EGR_WL_HILO & 16) {
                const cplx wl = t2l[c];
                ba[c * TS + l] = c ? cmul2(A[c], w, wl) : A[c];
                bb[(Q - 1 - c) * TS + l] = c ? cmul2(B[c], w, wl) : B[c];
                continue;
            }
            ba[c * TS + l] = c ? cmul(A[c], w) : A[c];
            bb[(Q - 1 - c) * TS + l] = c ? cmul(B[c], w) : B[c];          // row b: lane l will hold c' = Q - 1 - l
        }
        wl_wave_sync();
#pragma unroll
        for (int b2 = 0; b2 < Q; b2 += 2) {
            const float4 x = *(const float4*)(ba + l * TS + b2), y = *(const float4*)(bb + l * TS + b2);
            A[b2] = make_float2(x.x, x.y); A[b2 + 1] = make_float2(x.z, x.w);
            B[b2] = make_float2(y.x, y.y); B[b2 + 1] = make_float2(y.z, y.w);
        }
        wl_wave_sync();
        Bfly<Q>::run(A);           // A[d] = Xa[k1 + N1 (l + Q d)]
        Bfly<Q>::run(B);           // B[d] = Xb[(N1 - 1 - k1) + N1 ((Q - 1 - l) + Q d)]
    }
    EGR_STAMP(p, 2);
    float thr2 = p.thr2, tlev = p.thr;
    if (HOOK == 1 && p.max2) {                   // level relative to this iteration's spectrum maximum: carried from the previous
        tlev = p.thr * sqrtf(max2v);                         // iteration's hook, or written by k_row_wl<.., 2> (iteration 0)
        thr2 = tlev * tlev;
    }
    if (HOOK == 1 && p.max2_zero && blockIdx.x == 0 && tid < EGR_FL_MAX_SUB) fl_max2_clear(p.max2_zero, ch, tid);
    thr2 *= 4.f; tlev *= 2.f;                    // the hook judges 2 X (wl_pair_hook)
    const int soft = p.soft;
    const WlScale scd = wl_scale(p.inv_M_d);
    float mx2 = 0.f;
    if (!self) {
        if (lact) {
            const dcplx st = tb.hook_step;                                  // W_(2Q)
#pragma unroll
            for (int d = 0; d < Q; ++d) {
                cplx na, nb;
                mx2 = fmaxf(mx2, wl_pair_hook<HOOK>(A[d], B[Q - 1 - d], wrun, thr2, tlev, soft, scd, na, nb));
                if (HOOK != 2) { A[d] = na; B[Q - 1 - d] = nb; }
                wrun = dcmul(wrun, st);
            }
        }
    } else {
        // a self-paired row: spectrum to LDS in (block, k2) order, the generic pair loop of k_row, back to registers
        if (lact) {
#pragma unroll
            for (int d = 0; d < Q; ++d) { ba[l + Q * d] = A[d]; bb[(Q - 1 - l) + Q * d] = B[d]; }
        }
        __syncthreads();
        const dcplx wa = tw2d(p.wo, (unsigned)oa);
        int cnt, boff;
        if (oa == 0) { cnt = L / 2 + 1; boff = L; } else { cnt = (L + 1) / 2; boff = L - 1; }
        for (int k2 = tid; k2 < cnt; k2 += THREADS) {
            int pb = boff - k2;
            if (pb >= L) pb -= L;
            cplx* ea = lds + (k2 % N1) * S + (k2 / N1);
            cplx* eb = lds + (pb % N1) * S + (pb / N1);
            cplx na, nb;
            mx2 = fmaxf(mx2, wl_pair_hook<HOOK>(*ea, *eb, dcmul(wa, p.wk[k2]), thr2, tlev, soft, scd, na, nb));
            if (HOOK != 2) {
                *ea = na;
                if (pb != k2) *eb = nb;
            }
        }
        __syncthreads();
        if (HOOK != 2 && lact) {
#pragma unroll
            for (int d = 0; d < Q; ++d) { A[d] = ba[l + Q * d]; B[d] = bb[(Q - 1 - l) + Q * d]; }
            wl_wave_sync();
        }
    }
    EGR_STAMP(p, 3);
    __shared__ float red[16];
    if (HOOK == 2) {                             // the reduction a threshold RELATIVE to the spectrum's maximum needs: nothing is written back
        mx2 = block_max(mx2, red);
        if (tid == 0) fl_max2_commit(p.max2_out, ch, mx2);
        return;
    }
    if (HOOK == 1 && p.max2_next) {              // carried maximum: per-wave value now, ONE commit behind the barrier below
        mx2 = wave_max(mx2);
        if (lane == 0) red[wv] = mx2;
    }
    if (lact) {
        // ---- local step, inverse
        const cplx* t2 = tb.t2 + l * Q;                  // row a: c = l; W_(Q^2)^(b c) is symmetric in (b, c)
        const cplx* t2b = tb.t2 + (Q - 1 - l) * Q;       // row b: c' = Q - 1 - l
        const cplx* t2l = tb.t2l + l * Q;
        const cplx* t2bl = tb.t2l + (Q - 1 - l) * Q;
        dcplx ia_cur = make_double2(1.0, 0.0), ib_cur = ia_cur, ia_st = ia_cur, ib_st = ia_cur;
        if (EGR_WL_HILO & 128) { ia_st = tb.t2d[l]; ib_st = tb.t2d[Q - 1 - l]; }
        wl_bfly_inv<Q>(A);
        wl_bfly_inv<Q>(B);
#pragma unroll
        for (int b2 = 0; b2 < Q; b2 += 2) {
            cplx a0 = b2 ? cmulc(A[b2], t2[b2]) : A[b2], a1 = cmulc(A[b2 + 1], t2[b2 + 1]);
            cplx c0 = b2 ? cmulc(B[b2], t2b[b2]) : B[b2], c1 = cmulc(B[b2 + 1], t2b[b2 + 1]);
            if (EGR_WL_HILO & 16) {
                a0 = b2 ? cmulc2(A[b2], t2[b2], t2l[b2]) : A[b2]; a1 = cmulc2(A[b2 + 1], t2[b2 + 1], t2l[b2 + 1]);
                c0 = b2 ? cmulc2(B[b2], t2b[b2], t2bl[b2]) : B[b2]; c1 = cmulc2(B[b2 + 1], t2b[b2 + 1], t2bl[b2 + 1]);
            }
            if (EGR_WL_HILO & 128) {            // row a: W^(l b), row b: W^((Q - 1 - l) b) -- two runs in double
                auto mulc = [](const cplx z, const dcplx w) {
                    const double x = (double)z.x, y = (double)z.y;
                    return make_float2((float)(x * w.x + y * w.y), (float)(y * w.x - x * w.y));
                };
                a0 = b2 ? mulc(A[b2], ia_cur) : A[b2];
                c0 = b2 ? mulc(B[b2], ib_cur) : B[b2];
                ia_cur = dcmul(ia_cur, ia_st); ib_cur = dcmul(ib_cur, ib_st);
                a1 = mulc(A[b2 + 1], ia_cur);
                c1 = mulc(B[b2 + 1], ib_cur);
                ia_cur = dcmul(ia_cur, ia_st); ib_cur = dcmul(ib_cur, ib_st);
            }
            *(float4*)(ba + l * TS + b2) = make_float4(a0.x, a0.y, a1.x, a1.y);
            *(float4*)(bb + l * TS + b2) = make_float4(c0.x, c0.y, c1.x, c1.y);
        }
        wl_wave_sync();
#pragma unroll
        for (int c = 0; c < Q; ++c) { A[c] = ba[c * TS + l]; B[c] = bb[(Q - 1 - c) * TS + l]; }
        wl_wave_sync();
        wl_bfly_inv<Q>(A);
        wl_bfly_inv<Q>(B);
#pragma unroll
        for (int a = 0; a < Q; ++a) { ba[Q * a + l] = A[a]; bb[Q * a + l] = B[a]; }
    }
    __syncthreads();
    if (HOOK == 1 && p.max2_next && tid == 0) {
        float r = red[0];
#pragma unroll
        for (int i = 1; i < THREADS / 64; ++i) r = fmaxf(r, red[i]);
        fl_max2_commit(p.max2_next, ch, r);
    }
    EGR_STAMP(p, 4);
    // ---- cross step, inverse: LDS blocks -> twiddle^-1 -> inverse radix N1 -> global
#pragma unroll
    for (int xi = 0; xi < XIT; ++xi) {
        const int xt = tid + xi * THREADS;
        if (xt >= xlim) break;
        const int xr = xt >= QQ ? 1 : 0, xn2 = xt - QQ * xr;
        cplx* g = xr ? gb : ga;
        const cplx* s = lds + xr * RS + xn2;
        cplx v[N1], w[N1];
        if (!(EGR_WL_HILO & 64)) wl_load_tw<N1>(tb.t1 + xn2 * N1, w);
        v[0] = s[0];
        if (EGR_WL_HILO & 64) {
            const dcplx st = p.twd[xn2];
            dcplx cur = st;
#pragma unroll
            for (int k = 1; k < N1; ++k) {
                const cplx z = s[k * S];
                const double vx = (double)z.x, vy = (double)z.y;
                v[k] = make_float2((float)(vx * cur.x + vy * cur.y), (float)(vy * cur.x - vx * cur.y));
                cur = dcmul(cur, st);
            }
        } else if (EGR_WL_HILO & 8) {
            cplx wl[N1];
            wl_load_tw<N1>(tb.t1l + xn2 * N1, wl);
#pragma unroll
            for (int k = 1; k < N1; ++k) v[k] = cmulc2(s[k * S], w[k], wl[k]);
        } else
#pragma unroll
        for (int k = 1; k < N1; ++k) v[k] = cmulc(s[k * S], w[k]);
        wl_bfly_inv<N1>(v);
#pragma unroll
        for (int n1 = 0; n1 < N1; ++n1) g[n1 * QQ + xn2] = v[n1];
    }
    EGR_STAMP(p, 5);
}

// row lengths with a k_row_wl instantiation: EGR_WL_ROW_LIST(X), X(L, N1, Q)
#include "egr_wl_rows.h"
typedef void (*WlRowFn)(RowP, WlRowT, long long, cplx*);
struct WlRowEntry { int L, n1, q, threads, lds; WlRowFn fn, fn_variant, fn_max; };      // hooks 0 / 1 / 2
#define EGR_WL_ROW_ENTRY(LL, A, B) {LL, A, B, WlRow<A, B>::THREADS, WlRow<A, B>::LDS, k_row_wl<A, B, 0>, k_row_wl<A, B, 1>, k_row_wl<A, B, 2>},
static const WlRowEntry kWlRows[] = {EGR_WL_ROW_LIST(EGR_WL_ROW_ENTRY)};

// MODE 1 only (the middle pass of the loop): state -> twiddle^-1 -> IFFT_(R^2) -> FFT_(R^2) -> twiddle -> state, tiles of TC columns.
template <int R, int TC>
__global__ __launch_bounds__((WlCol<R, TC>::THREADS), EGR_WL_COL_WAVES) void k_col_wl(ColP p, WlColT tb, long long M, cplx* __restrict__ work) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    EGR_LDS_CANARY_ARM(smem);
    constexpr int RR = R * R, THREADS = WlCol<R, TC>::THREADS;
    __shared__ cplx t3s[RR];
    __shared__ cplx t3ls[(EGR_WL_HILO & 4) ? RR : 1];
    const int tile = (blockIdx.x & 7) * p.tiles_per_xcd + (blockIdx.x >> 3);
    if (tile >= p.ntiles) return;
    const int ch = blockIdx.y / p.nplanes, plane = blockIdx.y - ch * p.nplanes;
    const int nc = p.ncols;
    cplx* lds = (cplx*)EGR_LDS_BASE(smem);
    const int tid = threadIdx.x;
    const int b = tid / TC, cl = tid - b * TC, col = tile * TC + cl;
    const bool act = tid < R * TC && col < nc;
    cplx* W = work + (size_t)ch * M + (size_t)plane * RR * nc + col;
    for (int e = tid; e < RR; e += THREADS) t3s[e] = tb.t3[e];
    if (EGR_WL_HILO & 4) for (int e = tid; e < RR; e += THREADS) t3ls[e] = tb.t3l[e];
    EGR_STAMP(p, 0);
    cplx v[R], tw[R];
    cplx twl[(EGR_WL_HILO & 2) ? R : 1];
    // four-step twiddles W_M^(col (R a + b)) of this thread's rows: first value and ratio from the hi/lo tables, in double
    if (act) {
#pragma unroll
        for (int a = 0; a < R; ++a) v[a] = W[(size_t)(R * a + b) * nc];
        const dcplx w0 = tw2d(p.big, (unsigned)col * (unsigned)b), wst = tw2d(p.big, (unsigned)col * (unsigned)R);
        dcplx cur = w0;
#pragma unroll
        for (int a = 0; a < R; ++a) {
            tw[a] = make_float2((float)cur.x, (float)cur.y);             // kept for the way out: the run is formed once
            if (EGR_WL_HILO & 256) {                                     // the product itself in double; the run is formed again on the way out
                const double vx = (double)v[a].x, vy = (double)v[a].y;
                v[a] = make_float2((float)(vx * cur.x + vy * cur.y), (float)(vy * cur.x - vx * cur.y));
            } else if (EGR_WL_HILO & 2) {
                twl[a] = make_float2((float)(cur.x - (double)tw[a].x), (float)(cur.y - (double)tw[a].y));
                v[a] = cmulc2(v[a], tw[a], twl[a]);
            } else
            v[a] = cmulc(v[a], tw[a]);
            cur = dcmul(cur, wst);
        }
        wl_bfly_inv<R>(v);                               // v[c] = Z[c][b]
#pragma unroll
        for (int c = 0; c < R; ++c) lds[(c * R + b) * TC + cl] = v[c];
    }
    __syncthreads();
    EGR_STAMP(p, 1);
    if (act) {
        const cplx* tr = t3s + b * R;                    // this thread is now c = b of the first step; row c of the (symmetric) table
        const cplx* trl = t3ls + ((EGR_WL_HILO & 4) ? b * R : 0);
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const cplx z = lds[(b * R + j) * TC + cl];
            if (EGR_WL_HILO & 4) { v[j] = j ? cmulc2(z, tr[j], trl[j]) : z; continue; }
            v[j] = j ? cmulc(z, tr[j]) : z;
        }
        wl_bfly_inv<R>(v);                               // v[d] = t[c + R d]: the time-domain column (nothing happens to it in the loop)
        EGR_STAMP(p, 2);
        Bfly<R>::run(v);                                 // forward over a' = d: v[c''] = Z2[c''][b' = c]
#pragma unroll
        for (int j = 0; j < R; ++j) {                     // transposed: into the row it has just read
            if (EGR_WL_HILO & 4) { lds[(b * R + j) * TC + cl] = j ? cmul2(v[j], tr[j], trl[j]) : v[j]; continue; }
            lds[(b * R + j) * TC + cl] = j ? cmul(v[j], tr[j]) : v[j];
        }
    }
    __syncthreads();
    if (act) {
#pragma unroll
        for (int j = 0; j < R; ++j) v[j] = lds[(j * R + b) * TC + cl];           // Z2[c'' = b][b' = j]
        Bfly<R>::run(v);                                 // v[d''] = X[c'' + R d'']
        EGR_STAMP(p, 3);
        if (EGR_WL_HILO & 256) {
            dcplx cur = tw2d(p.big, (unsigned)col * (unsigned)b);
            const dcplx wst = tw2d(p.big, (unsigned)col * (unsigned)R);
#pragma unroll
            for (int a = 0; a < R; ++a) {
                const double vx = (double)v[a].x, vy = (double)v[a].y;
                W[(size_t)(R * a + b) * nc] = make_float2((float)(vx * cur.x - vy * cur.y), (float)(vx * cur.y + vy * cur.x));
                cur = dcmul(cur, wst);
            }
        } else
#pragma unroll
        for (int a = 0; a < R; ++a) W[(size_t)(R * a + b) * nc] = (EGR_WL_HILO & 2) ? cmul2(v[a], tw[a], twl[(EGR_WL_HILO & 2) ? a : 0]) : cmul(v[a], tw[a]);
    }
    EGR_STAMP(p, 4);
}
// A: the plan's outer ColP (columns of 625 or 441 points); ny = grid.y (channels x planes)
template <int R, int TC>
static inline void wl_launch_col_t(ColP A, const WlColT& tb, long long M, cplx* work, int ny, hipStream_t st) {
    A.TC = TC; A.TClog2 = 0; A.ntiles = (A.ncols + TC - 1) / TC; A.tiles_per_xcd = (A.ntiles + 7) / 8;      // (a ragged last tile is fine)
    hipLaunchKernelGGL((k_col_wl<R, TC>), dim3(8 * A.tiles_per_xcd, ny), dim3(WlCol<R, TC>::THREADS), EGR_LDS((WlCol<R, TC>::LDS)), st, A, tb, M, work);
}
static inline void wl_launch_col(const ColP& A, const WlColT& tb, long long M, cplx* work, int ny, hipStream_t st) {
    if (A.L == 441) wl_launch_col_t<21, 12>(A, tb, M, work, ny, st);
    else wl_launch_col_t<25, 8>(A, tb, M, work, ny, st);
}

// k_colb_wl -- the INNER column pass of a three-level plan (length L = LA x LB <= 144 over n2 inside each k1 plane, stride M3), one
// barrier: thread (b, col) loads n2 = LB a + b, radix-LA in registers, x W_L^(b c), transpose through LDS, thread (c, col) does the
// radix-LB over b and stores k2 = c + LA d.  FWD: FFT then x W_(L nc)^(col k2) (k_col<4>); else x conj W^(col n2) then IFFT
// (k_col<3>).  LB = 1: the whole transform is one register butterfly, no LDS.  The four-step twiddles are geometric runs in
// double (two table products per thread).  Small tiles (L x TCW x 8 bytes <= 18 KB), so 8-10 workgroups stream per CU.
template <int LA, int LB> struct WlInner {
    static constexpr int T = LA > LB ? LA : LB;                                   // threads per column
    static constexpr int TCW = LB == 1 ? 256 : (T >= 8 ? 16 : 32);               // columns per workgroup
    static constexpr int THREADS = LB == 1 ? 256 : ((T * TCW + 63) / 64) * 64;
    static constexpr int LDS = LB == 1 ? 0 : LA * LB * TCW * 8;
};
template <int R, bool INV> __device__ __forceinline__ void wl_bfly(cplx (&v)[R]) {
    if (INV) wl_bfly_inv<R>(v); else Bfly<R>::run(v);
}
template <> __device__ __forceinline__ void wl_bfly<1, false>(cplx (&)[1]) {}
template <> __device__ __forceinline__ void wl_bfly<1, true>(cplx (&)[1]) {}

template <int LA, int LB, bool FWD>
__global__ __launch_bounds__((WlInner<LA, LB>::THREADS)) void k_colb_wl(ColP p, const cplx* __restrict__ tab, long long M, cplx* __restrict__ work) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    EGR_LDS_CANARY_ARM(smem);
    using G = WlInner<LA, LB>;
    constexpr int TCW = G::TCW, L = LA * LB;
    const int tile = (blockIdx.x & 7) * p.tiles_per_xcd + (blockIdx.x >> 3);
    if (tile >= p.ntiles) return;
    const int ch = blockIdx.y / p.nplanes, plane = blockIdx.y - ch * p.nplanes;
    const int nc = p.ncols;
    cplx* lds = (cplx*)EGR_LDS_BASE(smem);
    const int tid = threadIdx.x;
    const int b = tid / TCW, cl = tid - b * TCW, col = tile * TCW + cl;
    const bool incol = col < nc;
    cplx* W = work + (size_t)ch * M + (size_t)plane * L * nc + col;
    cplx v[LA];
    if (incol && b < LB) {
#pragma unroll
        for (int a = 0; a < LA; ++a) v[a] = W[(size_t)(LB * a + b) * nc];
        if (!FWD) {
            dcplx cur = tw2d(p.big, (unsigned)col * (unsigned)b);
            const dcplx st = tw2d(p.big, (unsigned)col * (unsigned)LB);
#pragma unroll
            for (int a = 0; a < LA; ++a) {
                v[a] = cmulc(v[a], make_float2((float)cur.x, (float)cur.y));
                cur = dcmul(cur, st);
            }
        }
        wl_bfly<LA, !FWD>(v);
        if (LB == 1) {
            dcplx cur = make_double2(1.0, 0.0);
            const dcplx st = tw2d(p.big, (unsigned)col);
#pragma unroll
            for (int c = 0; c < LA; ++c) {
                W[(size_t)c * nc] = FWD ? cmul(v[c], make_float2((float)cur.x, (float)cur.y)) : v[c];
                if (FWD) cur = dcmul(cur, st);
            }
        } else {
            const cplx* t = tab + b * LA;
#pragma unroll
            for (int c = 0; c < LA; ++c) {
                const cplx z = c ? (FWD ? cmul(v[c], t[c]) : cmulc(v[c], t[c])) : v[c];
                lds[(c * LB + b) * TCW + cl] = z;
            }
        }
    }
    if (LB == 1) return;
    __syncthreads();
    if (incol && b < LA) {
        const int c = b;
        cplx u[LB];
#pragma unroll
        for (int j = 0; j < LB; ++j) u[j] = lds[(c * LB + j) * TCW + cl];
        wl_bfly<LB, !FWD>(u);
        dcplx cur = make_double2(1.0, 0.0), st = cur;
        if (FWD) {
            cur = tw2d(p.big, (unsigned)col * (unsigned)c);
            st = tw2d(p.big, (unsigned)col * (unsigned)LA);
        }
#pragma unroll
        for (int d = 0; d < LB; ++d) {
            W[(size_t)(c + LA * d) * nc] = FWD ? cmul(u[d], make_float2((float)cur.x, (float)cur.y)) : u[d];
            if (FWD) cur = dcmul(cur, st);
        }
    }
}

// inner lengths with a k_colb_wl instantiation: X(L, LA, LB) -- every 13-smooth length up to 64 that is one register butterfly
// or a product of two, and a few larger ones
#define EGR_WL_INNER_LIST(X) \
    X(2, 2, 1) X(3, 3, 1) X(4, 4, 1) X(5, 5, 1) X(6, 6, 1) X(7, 7, 1) X(8, 8, 1) X(9, 9, 1) \
    X(10, 10, 1) X(11, 11, 1) X(12, 12, 1) X(13, 13, 1) X(14, 2, 7) X(15, 3, 5) X(16, 16, 1) X(18, 3, 6) \
    X(20, 4, 5) X(21, 3, 7) X(22, 2, 11) X(24, 4, 6) X(25, 5, 5) X(26, 2, 13) X(27, 3, 9) X(28, 4, 7) \
    X(30, 5, 6) X(32, 4, 8) X(33, 3, 11) X(35, 5, 7) X(36, 6, 6) X(39, 3, 13) X(40, 5, 8) X(42, 6, 7) \
    X(44, 4, 11) X(45, 5, 9) X(48, 6, 8) X(49, 7, 7) X(50, 5, 10) X(52, 4, 13) X(54, 6, 9) X(55, 5, 11) \
    X(56, 7, 8) X(60, 6, 10) X(63, 7, 9) X(64, 8, 8) X(72, 8, 9) X(80, 8, 10) X(90, 9, 10) X(96, 8, 12) \
    X(100, 10, 10) X(120, 10, 12) X(144, 12, 12)

static bool wl_inner_supported(int L) {
    switch (L) {
#define X(LL, LA, LB) case LL:
        EGR_WL_INNER_LIST(X)
#undef X
        return true;
        default: return false;
    }
}
static void wl_inner_geometry(int L, int* la, int* lb, int* tcw) {
    switch (L) {
#define X(LL, LA, LB) case LL: *la = LA; *lb = LB; *tcw = WlInner<LA, LB>::TCW; return;
        EGR_WL_INNER_LIST(X)
#undef X
        default: *la = *lb = *tcw = 0;
    }
}
// B: the plan's inner ColP with TC / ntiles / tiles_per_xcd set for WlInner<LA, LB>::TCW
template <int LA, int LB> static void wl_launch_inner_t(const ColP& B, const cplx* tab, bool forward, long long M, cplx* work, int nstates, hipStream_t st) {
    using G = WlInner<LA, LB>;
    const dim3 grid(8 * B.tiles_per_xcd, nstates * B.nplanes), blk(G::THREADS);
    const size_t lds = EGR_LDS(G::LDS);
    if (forward) hipLaunchKernelGGL((k_colb_wl<LA, LB, true>), grid, blk, lds, st, B, tab, M, work);
    else hipLaunchKernelGGL((k_colb_wl<LA, LB, false>), grid, blk, lds, st, B, tab, M, work);
}
static void wl_launch_inner(const ColP& B, const cplx* tab, bool forward, long long M, cplx* work, int nstates, hipStream_t st) {
    switch (B.L) {
#define X(LL, LA, LB) case LL: wl_launch_inner_t<LA, LB>(B, tab, forward, M, work, nstates, st); break;
        EGR_WL_INNER_LIST(X)
#undef X
        default: break;
    }
}

}  // namespace egr
