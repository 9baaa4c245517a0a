// Host-side planning: radix schedules, twiddle tables, four-step split selection.  Pure host code
// (no HIP runtime calls) so it can be exercised on machines without a GPU.
#include "egr_plan.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <vector>
#include <string>

#include "egr_common.h"
#include "egr_wl_rows.h"

namespace egr {

static thread_local std::string g_err;

void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}

const char* last_error_cstr() { return g_err.c_str(); }

bool make_schedule(int L, FftDesc* d, int max_prime) {
    memset(d, 0, sizeof(*d));
    d->L = L;
    int n = L, st = 0, ns = 1;
    auto push = [&](int r) -> bool {
        if (st >= EGR_MAX_STAGES) return false;
        d->radix[st] = r;
        d->ns[st] = ns;
        d->inv_ns[st] = 1.0f / (float)ns;
        ns *= r;
        ++st;
        return true;
    };
    if (L < 1) return false;
    // Radix 8 and 9 (nested 4x2 / 3x3 register butterflies, EGR_RADIX_8_9, default) cut the stage count at 105 VGPRs;
    // the larger composites 16 / 25 (EGR_COMPOSITE_RADIX) push the all-radix kernel to 190 VGPRs and measured slower.
#ifdef EGR_COMPOSITE_RADIX
    static const bool small_only = [] { const char* e = getenv("EGR_FFT_RADIX"); return e && e[0] == 's'; }();
#else
    const bool small_only = true;
#endif
    if (!small_only) {
        while (n % 16 == 0) { if (!push(16)) return false; n /= 16; }
        if (n % 8 == 0) { if (!push(8)) return false; n /= 8; }
    }
#ifdef EGR_RADIX_8_9
    while (n % 8 == 0) { if (!push(8)) return false; n /= 8; }
#endif
    while (n % 4 == 0) {
        if (!push(4)) return false;
        n /= 4;
    }
    if (n % 2 == 0) {
        if (!push(2)) return false;
        n /= 2;
    }
    if (!small_only) {
        while (n % 25 == 0) { if (!push(25)) return false; n /= 25; }
        while (n % 9 == 0) { if (!push(9)) return false; n /= 9; }
    }
#ifdef EGR_RADIX_8_9
    while (n % 9 == 0) { if (!push(9)) return false; n /= 9; }
#endif
    const int odd[] = {3, 5, 7, 11, 13};
    for (int p : odd) {
        while (n % p == 0) {
            if (!push(p)) return false;
            n /= p;
        }
    }
    // larger primes (STFT frame lengths of the metrics widgets, e.g. n_fft = 2176 = 2^7 * 17): one generic stage each, evaluated
    // as a direct r-point DFT from LDS (fft_stage_generic)
    for (int p = 17; p <= max_prime && n > 1; p += 2) {
        while (n % p == 0) {
            if (!push(p)) return false;
            n /= p;
        }
    }
    d->nst = st;
    return n == 1;
}

void make_twiddles(std::vector<float2>& out, int64_t count, int64_t num, int64_t den) {
    out.resize((size_t)count);
    const long double two_pi = 6.283185307179586476925286766559L;
    for (int64_t j = 0; j < count; ++j) {
        // reduce j*num mod den in integers so the angle stays small and exact before the multiply
        __int128 r = ((__int128)j * (__int128)num) % (__int128)den;
        long double ang = -two_pi * (long double)(int64_t)r / (long double)den;
        out[(size_t)j] = make_float2((float)cosl(ang), (float)sinl(ang));
    }
}

void make_twiddles_d(std::vector<double2>& out, int64_t count, int64_t num, int64_t den) {
    out.resize((size_t)count);
    const long double two_pi = 6.283185307179586476925286766559L;
    for (int64_t j = 0; j < count; ++j) {
        __int128 r = ((__int128)j * (__int128)num) % (__int128)den;
        long double ang = -two_pi * (long double)(int64_t)r / (long double)den;
        out[(size_t)j] = make_double2((double)cosl(ang), (double)sinl(ang));
    }
}

static int ilog2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

struct FlSplit;
static void finish_split_impl(FlSplit& best, int tc_hint);
static inline void finish_split(FlSplit& best, int tc_hint) { finish_split_impl(best, tc_hint); }

static int pick_tc(int L, int ncols, int hint) {
    int tc = hint > 0 ? hint : (L > 512 ? 4 : 16);
    if (L > 1024 && tc > 4) tc = 4;                   // 2 L TC 8 bytes of LDS: columns of up to 2048 points take 4-column tiles
    if (L == 1 && hint <= 0) tc = 256;               // degenerate column pass: plain streaming
    int l2 = ilog2(tc);
    tc = 1 << l2;
    while (tc > 1 && tc / 2 >= ncols) tc /= 2;
    return tc;
}

FlSplit plan_split(int64_t N, int m1_hint, int tc_hint, int max_col) {
    FlSplit best;
    memset(&best, 0, sizeof(best));
    best.ok = false;
    best.N = N;
    best.M3 = 1;
    if (N < 2 || (N & 1)) return best;
    const int64_t M = N / 2;
    best.M = M;
    // LDS-capacity limits of k_col / k_row: outer columns up to 2048 points (4-column tiles above 1024), inner columns
    // of a three-level plan up to 1024, rows up to 4096.  Two levels reach N = 16.8 M (5.8 minutes at 48 kHz) that way; a third
    // level costs two more passes over the state per iteration.
    const int64_t MAX_COL = max_col > 1024 ? 2048 : 1024, MAX_COL_INNER = 1024, MAX_ROW = 4096;
    double best_score = 1e300;
    // ---- two levels: M = M1 * M2 ----
    for (int64_t m1 = 1; m1 <= MAX_COL && m1 <= M; ++m1) {
        if (M % m1) continue;
        const int64_t m2 = M / m1;
        if (m2 > MAX_ROW && !(m1 == 625 && m2 == 4608)) continue;      // 625 x 4608 (120 s at 48 kHz): rows on k_row_wl<32, 12>, 88 KB of LDS
        if (m1_hint > 0 && m1 != m1_hint) continue;
        FftDesc f1, f2;
        if (!make_schedule((int)m1, &f1) || !make_schedule((int)m2, &f2)) continue;
        // fewer LDS stages first; then keep the column tile small enough for >= 2 workgroups per CU;
        // then prefer a balanced split.
        double score = 1000.0 * (f1.nst + f2.nst);
        if (m1 > 640) score += 300.0;
        if (m2 > 2560) score += 300.0;
        score += 10.0 * fabs(log((double)m1 * 2.0 / (double)m2));
        // columns of 625 points run on k_col_wl (egr_fatllama_wl.h: two barriers, 8-column tiles) whatever the row length: measured
        // against the planner's other choice at 10 / 20 / 50 / 75 / 90 s of 48 kHz audio (stereo, 800 iterations): 29.0 -> 23,
        // 38.8 -> 27, 50.9 -> 46, 72.6 -> 62, 85.9 -> 77 ms
        if (m1 == 625 && m2 % 8 == 0) score -= 2500.0;
        // columns of 441 = 21 x 21 points likewise (k_col_wl<21, 12>): the 44.1 kHz family, N / 2 = 441 x 50 T for T seconds; the
        // rows of 50 T points run on k_row_wl<T / 2, 10> where instantiated (19 lengths, T = 8 ... 64 s), else stage by stage
        if (m1 == 441 && m2 % 4 == 0) score -= 2500.0;
        if (score < best_score) {
            best_score = score;
            best.ok = true; best.levels = 2;
            best.M1 = (int)m1; best.M2 = (int)m2; best.M3 = 1;
            best.f1 = f1; best.f2 = f2;
        }
    }
    // ---- M = 625 x k x 2304 (whole minutes at 48 kHz, k >= 3): the outer column pass and the row pass run on the two-barrier
    // kernels of egr_fatllama_wl.h (k_col_wl, k_row_wl) and the inner pass of length k on k_colb_wl where k has an instantiation.
    // Measured against the planner's other choices (tools/probe_c5_plans.py, stereo, ms per iteration): k = 60 (N = 172.8 M,
    // BASELINE configs[4]) 2.86 vs 6.03 (640 x 72 x 1875); k = 10: 0.46 vs 0.54; k = 5: 0.27 vs 0.47 for the TWO-level plan
    // 1800 x 4000; k = 2 ties with 1125 x 2560 and keeps two levels.
    if (m1_hint <= 0 && tc_hint <= 0 && M % (625LL * 2304LL) == 0) {
        const int64_t k = M / (625LL * 2304LL);
        FftDesc fk;
        if (k >= 3 && k <= MAX_COL_INNER && make_schedule((int)k, &fk)) {
            FlSplit sp = plan_split_explicit(N, 625, (int)k, 2304, 0);
            if (sp.ok) return sp;
        }
    }
    // ---- the same shape for every row length with a two-barrier kernel: M = C x k x L, C = 625 or 441 columns on k_col_wl, rows of L
    // points on k_row_wl<N1, Q> (egr_wl_rows.h), the inner pass of length k stage by stage or on k_colb_wl -- taken when no two-level
    // plan with such columns exists (files beyond ~100 s: 150 s at 48 kHz = 625 x 2 x 2880, 120 s at 44.1 kHz = 441 x 2 x 3000).  The
    // longest such row wins (the inner pass is the cheapest of the four; measured in profiles/r04/fatllama_lengths_end_of_round.log).
    if (m1_hint <= 0 && tc_hint <= 0 && !(best.ok && (best.M1 == 625 || best.M1 == 441))) {
        static const int wl_rows[] = {
#define X(LL, A, B) LL,
            EGR_WL_ROW_LIST(X)
#undef X
        };
        // A third level costs two more passes over the state per iteration, so it only replaces a two-level plan the planner itself
        // prices as slow (columns beyond 640 points: 4-column tiles; rows beyond 2560) -- 150 s at 48 kHz, 1800 x 4000 -> 625 x 2 x 2880.
        const bool two_level_is_fine = best.ok && best.M1 <= 640 && best.M2 <= 2560;
        struct Cand { int64_t L, cols, k; };
        std::vector<Cand> cands;
        for (int64_t cols : {625LL, 441LL})
            for (int L : wl_rows) {
                if (M % (cols * L)) continue;
                const int64_t k = M / (cols * L);
                FftDesc fk;
                if (k < 2 || k > MAX_COL_INNER || !make_schedule((int)k, &fk)) continue;
                cands.push_back({(int64_t)L, cols, k});
            }
        std::sort(cands.begin(), cands.end(), [](const Cand& a, const Cand& b) { return a.L != b.L ? a.L > b.L : a.cols > b.cols; });
        if (!two_level_is_fine)
            for (const Cand& c : cands) {                  // longest row first; a length the explicit plan refuses falls through to the next
                FlSplit sp = plan_split_explicit(N, (int)c.cols, (int)c.k, (int)c.L, 0, 4608);
                if (sp.ok) return sp;
            }
    }
    // ---- three levels: M = M1 * M2 * M3 (only when two do not fit) ----
    if (!best.ok) {
        for (int64_t m3 = 2; m3 <= MAX_ROW && m3 <= M; ++m3) {
            if (M % m3) continue;
            FftDesc f3;
            if (!make_schedule((int)m3, &f3)) continue;
            const int64_t R = M / m3;
            for (int64_t m1 = 2; m1 <= MAX_COL && m1 <= R; ++m1) {
                if (R % m1) continue;
                const int64_t m2 = R / m1;
                if (m2 > MAX_COL_INNER || m2 < 2) continue;
                if (m1_hint > 0 && m1 != m1_hint) continue;
                FftDesc f1, f2;
                if (!make_schedule((int)m1, &f1) || !make_schedule((int)m2, &f2)) continue;
                double score = 1000.0 * (f1.nst + 2 * f2.nst + f3.nst);   // pass B runs twice per iteration
                if (m1 > 640) score += 300.0;
                if (m2 > 640) score += 300.0;
                if (m3 > 2560) score += 300.0;
                score += 10.0 * (fabs(log((double)m1 / (double)m2)) + fabs(log((double)m1 * 2.0 / (double)m3)));
                if (score < best_score) {
                    best_score = score;
                    best.ok = true; best.levels = 3;
                    best.M1 = (int)m1; best.M2 = (int)m2; best.M3 = (int)m3;
                    best.f1 = f1; best.f2 = f2; best.f3 = f3;
                }
            }
        }
    }
    if (!best.ok) return best;
    finish_split(best, tc_hint);
    return best;
}

FlSplit plan_split_explicit(int64_t N, int m1, int m2, int m3, int tc_hint, int max_row) {
    FlSplit sp;
    memset(&sp, 0, sizeof(sp));
    sp.N = N; sp.M = N / 2; sp.M3 = 1;
    if (N < 2 || (N & 1) || m1 < 1 || m2 < 1 || m3 < 1) return sp;
    if ((int64_t)m1 * m2 * m3 != sp.M) return sp;
    sp.levels = m3 > 1 ? 3 : 2;
    sp.M1 = m1; sp.M2 = m2; sp.M3 = m3;
    if (m1 > 2048 || (sp.levels == 3 ? (m2 > 1024 || m3 > max_row) : m2 > max_row)) return sp;
    if (!make_schedule(m1, &sp.f1) || !make_schedule(m2, &sp.f2)) return sp;
    if (sp.levels == 3 && !make_schedule(m3, &sp.f3)) return sp;
    sp.ok = true;
    finish_split(sp, tc_hint);
    return sp;
}

static void finish_split_impl(FlSplit& best, int tc_hint) {
    if (best.levels == 2) {
        best.TC = pick_tc(best.M1, best.M2, tc_hint);
        best.TClog2 = ilog2(best.TC);
        best.TCb = 1; best.TCblog2 = 0;
        best.lds_col = (size_t)2 * best.M1 * best.TC * sizeof(float2);
        best.lds_colb = 0;
        best.lds_row = (size_t)2 * 2 * best.M2 * sizeof(float2);
    } else {
        best.TC = pick_tc(best.M1, best.M2 * best.M3, tc_hint);
        best.TClog2 = ilog2(best.TC);
        best.TCb = pick_tc(best.M2, best.M3, tc_hint);
        best.TCblog2 = ilog2(best.TCb);
        best.lds_col = (size_t)2 * best.M1 * best.TC * sizeof(float2);
        best.lds_colb = (size_t)2 * best.M2 * best.TCb * sizeof(float2);
        best.lds_row = (size_t)2 * 2 * best.M3 * sizeof(float2);
    }
}

}  // namespace egr

extern "C" const char* egr_last_error(void) { return egr::last_error_cstr(); }
extern "C" int egr_abi_version(void) { return EGR_ABI_VERSION; }
