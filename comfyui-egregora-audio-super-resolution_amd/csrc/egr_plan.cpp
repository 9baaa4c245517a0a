// Host-side planning: radix schedules, twiddle tables, four-step split selection.  Pure host code
// (no HIP runtime calls) so it can be exercised on machines without a GPU.
#include "egr_plan.h"

#include <math.h>
#include <stdio.h>
#include <string.h>

#include <mutex>
#include <string>

#include "egr_common.h"

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

bool make_schedule(int L, FftDesc* d) {
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
    while (n % 4 == 0) {
        if (!push(4)) return false;
        n /= 4;
    }
    if (n % 2 == 0) {
        if (!push(2)) return false;
        n /= 2;
    }
    const int odd[] = {3, 5, 7, 11, 13};
    for (int p : odd) {
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

static int ilog2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

FlSplit plan_split(int64_t N, int m1_hint, int tc_hint) {
    FlSplit best;
    memset(&best, 0, sizeof(best));
    best.ok = false;
    best.N = N;
    if (N < 2 || (N & 1)) return best;
    const int64_t M = N / 2;
    best.M = M;
    const int64_t MAX_M1 = 1024, MAX_M2 = 4096;   // LDS-capacity limits of k_col / k_row
    double best_score = 1e300;
    for (int64_t m1 = 1; m1 <= MAX_M1 && m1 <= M; ++m1) {
        if (M % m1) continue;
        const int64_t m2 = M / m1;
        if (m2 > MAX_M2) continue;
        if (m1_hint > 0 && m1 != m1_hint) continue;
        FftDesc f1, f2;
        if (!make_schedule((int)m1, &f1) || !make_schedule((int)m2, &f2)) continue;
        // fewer LDS stages first; then keep the column tile small enough for >= 2 workgroups per CU;
        // then prefer a balanced split.
        double score = 1000.0 * (f1.nst + f2.nst);
        if (m1 > 640) score += 300.0;
        if (m2 > 2560) score += 300.0;
        score += 10.0 * fabs(log((double)m1 * 2.0 / (double)m2));
        if (score < best_score) {
            best_score = score;
            best.ok = true;
            best.M1 = (int)m1;
            best.M2 = (int)m2;
            best.f1 = f1;
            best.f2 = f2;
        }
    }
    if (!best.ok) return best;
    int tc = tc_hint > 0 ? tc_hint : (best.M1 > 512 ? 8 : 16);
    if (best.M1 == 1) tc = tc_hint > 0 ? tc_hint : 256;   // degenerate column pass: plain streaming
    // power of two, and never wider than the row
    int l2 = ilog2(tc);
    tc = 1 << l2;
    while (tc > 1 && tc / 2 >= best.M2) { tc /= 2; --l2; }
    best.TC = tc;
    best.TClog2 = l2;
    best.lds_col = (size_t)2 * best.M1 * tc * sizeof(float2);
    best.lds_row = (size_t)2 * 2 * best.M2 * sizeof(float2);
    return best;
}

}  // namespace egr

extern "C" const char* egr_last_error(void) { return egr::last_error_cstr(); }
extern "C" int egr_abi_version(void) { return EGR_ABI_VERSION; }
