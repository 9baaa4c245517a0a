// FlashSR model handle behind the C ABI:
//   egr_flashsr_create / egr_flashsr_infer / egr_flashsr_destroy stand in for the upstream calls
//   `FlashSR(student_ldm.pth, sr_vocoder.pth, vae.pth)` + `.eval().to(dev)` and `model(x[C,245760], lowpass_input)` that the
//   reference makes at egregora_audio_super_resolution.py:346-359 and :361-369.
// The whole graph walk lives here: layer table (egr_flashsr_config = flashsr_arch.FlashSRConfig) -> weight repacking -> kernel
// launches of the operators in egr_nn_*.hip, with a stream-ordered scratch arena instead of one allocation per operator.
// The host (Python or C) only supplies named fp32 tensors in torch layouts and calls infer on device pointers.
#include <math.h>
#include <cmath>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "egr_common.h"

extern "C" {
int egr_pack_weight(const float* src, float* dst, int layout, int K, int N, int Ci, int Co, int KH, int KW, void* stream);
int egr_phase_weights(const float* w_oihw, float* dst4, int Co, int Ci, void* stream);
int egr_winograd_pack_u(const float* w_oihw, float* dst, const double* G_dev, int np, int Co, int Ci, void* stream);
}

namespace {

using egr::set_error;

enum { ACT_NONE = 0, ACT_SILU = 1, ACT_TANH = 2, ACT_LEAKY = 3, ACT_LOGCLAMP = 4 };
enum { EW_ADD = 0, EW_AXPBY = 1, EW_SILU = 2, EW_SCALE = 3, EW_COPY = 4, EW_ADD_SCALE = 5 };

#define OKR(call)                 \
    do {                          \
        int rc__ = (call);        \
        if (rc__ != EGR_OK) return rc__; \
    } while (0)

// ------------------------------------------------------------------------------------------------ scratch arena
// Best-fit allocator with splitting and coalescing over a few large hipMalloc'ed slabs.  Everything runs on ONE stream, so a
// block may be handed out again as soon as the host has enqueued its last consumer (stream order does the rest).  After the first
// forward of a given row count no allocation reaches the runtime, and the footprint stays near the peak of simultaneously live
// activations (exact-size free lists, the first version, held 47 GB for a 26-row pass; this holds the live peak + slab slack).
// host time spent in the arenas' hipMalloc calls, in microseconds (EGR_FSR_TRACE=1 prints it per call); forwards on different devices run
// under different per-device locks, so the counter is atomic
static std::atomic<long long> g_arena_malloc_us{0};
static inline double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Arena {
    struct Slab {
        char* base = nullptr;
        size_t size = 0;
        std::map<char*, size_t> by_addr;                 // free blocks
        std::multimap<size_t, char*> by_size;
        void insert(char* p, size_t n) { by_addr[p] = n; by_size.emplace(n, p); }
        void erase(char* p, size_t n) {
            by_addr.erase(p);
            auto r = by_size.equal_range(n);
            for (auto it = r.first; it != r.second; ++it) if (it->second == p) { by_size.erase(it); break; }
        }
    };
    std::vector<Slab> slabs;
    size_t total = 0;
    static size_t round_up(size_t b) { b = (b + 255) & ~(size_t)255; return b ? b : 256; }
    void* get(size_t bytes) {
        bytes = round_up(bytes);
        Slab* best = nullptr;
        std::multimap<size_t, char*>::iterator bi;
        for (auto& s : slabs) {
            auto it = s.by_size.lower_bound(bytes);
            if (it != s.by_size.end() && (!best || it->first < bi->first)) { best = &s; bi = it; }
        }
        if (!best) {
            Slab s;
            s.size = std::max(bytes, (size_t)1 << 31);                      // 2 GiB slabs, or one request if larger
            const double t0 = now_ms();
            struct T { double t0; ~T() { g_arena_malloc_us += (long long)((now_ms() - t0) * 1e3); } } timer{t0};
            if (hipMalloc((void**)&s.base, s.size) != hipSuccess) {
                s.size = bytes;                                             // memory is tight: exactly what is needed
                if (hipMalloc((void**)&s.base, s.size) != hipSuccess) {
                    set_error("FlashSR scratch arena: hipMalloc(%zu) failed (%zu bytes held)", bytes, total);
                    return nullptr;
                }
            }
            total += s.size;
            s.insert(s.base, s.size);
            slabs.push_back(std::move(s));
            best = &slabs.back();
            bi = best->by_size.lower_bound(bytes);
        }
        char* p = bi->second;
        const size_t have = bi->first;
        best->erase(p, have);
        if (have > bytes) best->insert(p + bytes, have - bytes);
        return p;
    }
    void put(void* ptr, size_t bytes) {
        bytes = round_up(bytes);
        char* p = (char*)ptr;
        for (auto& s : slabs) {
            if (p < s.base || p >= s.base + s.size) continue;
            auto nx = s.by_addr.lower_bound(p);
            if (nx != s.by_addr.end() && nx->first == p + bytes) {          // merge with the block behind
                const size_t n = nx->second;
                s.erase(nx->first, n);
                bytes += n;
            }
            auto pv = s.by_addr.lower_bound(p);
            if (pv != s.by_addr.begin()) {
                --pv;
                if (pv->first + pv->second == p) {                          // merge with the block in front
                    char* q = pv->first;
                    const size_t n = pv->second;
                    s.erase(q, n);
                    p = q;
                    bytes += n;
                }
            }
            s.insert(p, bytes);
            return;
        }
    }
    ~Arena() { for (auto& s : slabs) hipFree(s.base); }
};

struct Ten {                       // an fp32 activation owned by the arena (move-only)
    Arena* a = nullptr;
    float* p = nullptr;
    size_t bytes = 0;
    int nd = 0;
    int64_t d[4] = {0, 0, 0, 0};
    std::shared_ptr<Ten> part;     // GroupNorm partial sums left by the F(4x4) output transform (egr_winograd4_output_stats)
    int part_tiles = 0;
    // per-batch-row max |x| (bits; a slice of the context's pool, valid until the end of the forward): written by the producer of
    // the tensor or by the first split contraction that reads it (row_amax_of), reused by every later one
    mutable unsigned* rs = nullptr;
    Ten() {}
    Ten(const Ten&) = delete;
    Ten& operator=(const Ten&) = delete;
    Ten(Ten&& o) noexcept { *this = std::move(o); }
    Ten& operator=(Ten&& o) noexcept {
        if (this != &o) {
            release();
            a = o.a; p = o.p; bytes = o.bytes; nd = o.nd; memcpy(d, o.d, sizeof(d)); part = std::move(o.part); part_tiles = o.part_tiles;
            rs = o.rs; o.rs = nullptr;
            o.a = nullptr; o.p = nullptr; o.bytes = 0;
        }
        return *this;
    }
    ~Ten() { release(); }
    void release() { if (a && p) a->put(p, bytes); p = nullptr; a = nullptr; part.reset(); rs = nullptr; }
    int64_t numel() const { int64_t n = 1; for (int i = 0; i < nd; ++i) n *= d[i]; return n; }
    // non-owning view of the same storage (the owner must outlive it); keeps the GroupNorm partials the owner carries
    Ten alias() const { Ten v; v.nd = nd; memcpy(v.d, d, sizeof(d)); v.p = p; v.part = part; v.part_tiles = part_tiles; v.rs = rs; return v; }
    void view(std::initializer_list<int64_t> s) { nd = 0; for (int64_t v : s) d[nd++] = v; }
};

struct Wt {                        // one entry of the weight store
    float* w = nullptr;            // raw tensor (norms, biases, snake parameters) or fp32 slab-major pack
    void* w3 = nullptr;            // three-way bf16 split of the pack (egr_split3_pack)
    void* w2 = nullptr;            // two fp16 terms of w * w_scale (egr_split2h_pack)
    float w_scale = 1.f;
    int KH = 0, KW = 0, Cin = 0, Cout = 0;   // logical shape of a packed contraction (Cout = GEMM N)
    int64_t zfloats = 0;           // floats per component of a z-stacked Winograd pack
    int64_t numel = 0;
};

struct ProfRec { std::string kind; double flops; hipEvent_t a, b; std::string detail; };

}  // namespace

// Everything one in-flight forward owns: its stream, scratch arena and workspaces.  Context 0 runs on the caller's stream; the
// others on the handle's side streams when a pass is split into concurrent row groups (egr_flashsr_infer).
struct FsrCtx {
    hipStream_t st = nullptr;
    Arena arena;
    void* gn_ws = nullptr;
    size_t gn_ws_bytes = 0;
    std::map<int, egr_fatllama_plan*> lp_plans;       // input low-pass: spectral-gain plans per row count
    hipEvent_t done = nullptr;
    // per-row operand maxima of the forward in flight (fp16 operand scheme): one R-entry slice per measured tensor, zeroed as a
    // whole when a forward starts
    unsigned* rs_pool = nullptr;
    size_t rs_cap = 0, rs_used = 0;
    std::vector<unsigned*> rs_retired;                // pools outgrown in mid-forward: still read by enqueued kernels, freed when the next forward starts
    unsigned* rs_ones = nullptr;                      // "maximum 1.0" for every row (operands known to lie in [0, 1]: soft-max outputs); rs_ones_rows entries
    int rs_ones_rows = 0;
};

struct egr_flashsr {
    egr_flashsr_config cfg;
    unsigned flags = 0;
    int device = 0;
    std::vector<std::unique_ptr<FsrCtx>> ctxs;        // [0] = the caller's stream
    FsrCtx* cx = nullptr;                             // context of the forward being enqueued
    int max_groups = 2;                               // concurrent row groups per pass (EGREGORA_FLASHSR_STREAMS)
    int min_group_rows = 6;
    hipEvent_t ev_fork = nullptr;
    std::vector<hipStream_t> verified_for;            // caller streams the side streams were checked against
    std::vector<void*> owned;                         // weight allocations
    std::unordered_map<std::string, Wt> W;
    std::vector<std::string> blk_name;                // UNet block table (flashsr_arch.unet_blocks)
    std::vector<int> blk_cin, blk_cout, blk_attn;
    float* window = nullptr; float* filt = nullptr;
    int ldm = 0, lat_h = 0, lat_w = 0;
    float alpha = 0.f, sigma = 0.f;
    float* d_wmax = nullptr;                          // one float: weight maxima at pack time
    int rows_per_pass = 32;
    // scratch budget (EGREGORA_FLASHSR_ARENA_GB / egr_flashsr_set_arena_cap; 0 = none): rows per pass are lowered until the arenas
    // of a pass are expected to fit (measured bytes per row once a pass has run, an estimate from the layer table before)
    double arena_cap = 0.0, arena_row_bytes = 0.0;
    int wino_min_ch = 128;
    double flops = 0.0; bool count_flops = false;
    bool profiling = false;
    std::vector<ProfRec> prof;
    hipStream_t st = nullptr;                         // stream of the forward being enqueued (= cx->st)
    void use(FsrCtx* c) { cx = c; st = c->st; }
    // Two-term fp16 operand scheme of egr_flashsr_infer (csrc/egr_nn_gemm_s3.hip, scheme 1).  Every batch row of every split
    // contraction's input is scaled by its own power of two, which the kernel derives from that row's max |x|; the maxima are
    // written on the device by the tensor's producer (Winograd input transforms) or by one k_absmax_rows pass the first time a
    // tensor feeds a split contraction (row_amax_of).  Nothing about the scales passes through the host or survives the call:
    // the result of a row is a function of (weights, that row's input, seed, row id) and of the row count of its forward only
    // through tile choices; no value can leave fp16's range (the scale comes from the row's own maximum), so there is nothing to
    // verify or re-run, and the call is as asynchronous as the bf16 one.
    bool h2 = true;                                   // scheme available (off: EGR_FSR_SPLIT_BF16X3, EGREGORA_FLASHSR_SPLIT=bf16x3, f32 MFMA)
    int h2_mode = -1;                                 // of the forward being enqueued: -1 bf16 terms, 1 fp16 terms
    bool h2_fwd = false;                              // egr_flashsr_forward too runs the fp16 terms (set_split 2: stage taps for tests)
    int h2_nweights = 0;                              // contraction weights that hold fp16 terms
    int R = 1;                                        // batch rows of the forward being enqueued
    bool out_amax_on = true;                          // contraction epilogues leave the row maxima of their outputs (EGREGORA_FLASHSR_OUT_AMAX=0: off)
    int direct3x3_max_cout = 128;                     // EGREGORA_FLASHSR_DIRECT3X3_MAX_COUT (0: off): see gn_conv3
    bool fused_amp = true;                            // EGREGORA_FLASHSR_FUSED_AMP=0: the thin AMP units as four launches each
    bool next_out_ra = false;                         // set by a call site whose output feeds another split contraction directly; consumed by the next conv()
    int64_t h2_calls = 0;

    bool f32_mfma() const { return (flags & EGR_FSR_F32_MFMA) != 0; }
    bool has(const std::string& k) const { return W.find(k) != W.end(); }
    const Wt* get(const std::string& k) const { auto it = W.find(k); return it == W.end() ? nullptr : &it->second; }
    float* ptr(const std::string& k) const { auto it = W.find(k); return it == W.end() ? nullptr : it->second.w; }
};

namespace {

typedef egr_flashsr M;

int dev_alloc(M* m, size_t bytes, void** out) {
    void* p = nullptr;
    if (bytes == 0) bytes = 16;
    if (hipMalloc(&p, bytes) != hipSuccess) { set_error("FlashSR weights: hipMalloc(%zu) failed", bytes); return EGR_ERR_ALLOC; }
    m->owned.push_back(p);
    *out = p;
    return EGR_OK;
}

int new_ten(M* m, Ten& t, std::initializer_list<int64_t> shape) {
    t.release();
    t.view(shape);
    t.bytes = (size_t)t.numel() * sizeof(float);
    t.p = (float*)m->cx->arena.get(t.bytes);
    if (!t.p) return EGR_ERR_ALLOC;
    t.a = &m->cx->arena;
    return EGR_OK;
}

// ------------------------------------------------------------------------------------------------ layer table
void build_blocks(M* m) {
    const egr_flashsr_config& c = m->cfg;
    auto is_attn = [&](int ds) { for (int i = 0; i < c.unet_n_attn; ++i) if (c.unet_attn_ds[i] == ds) return 1; return 0; };
    auto add = [&](const std::string& n, int ci, int co, int at) { m->blk_name.push_back(n); m->blk_cin.push_back(ci); m->blk_cout.push_back(co); m->blk_attn.push_back(at); };
    const int mc = c.unet_ch;
    add("in.0.conv_in", 2 * c.z_ch, mc, 0);
    std::vector<int> skip{mc};
    int ch = mc, ds = 1, idx = 1;
    for (int lv = 0; lv < c.unet_levels; ++lv) {
        for (int r = 0; r < c.unet_res; ++r) {
            add("in." + std::to_string(idx) + ".block", ch, mc * c.unet_mult[lv], is_attn(ds));
            ch = mc * c.unet_mult[lv];
            skip.push_back(ch);
            ++idx;
        }
        if (lv != c.unet_levels - 1) {
            add("in." + std::to_string(idx) + ".down", ch, ch, 0);
            skip.push_back(ch);
            ds *= 2;
            ++idx;
        }
    }
    add("mid.0.block", ch, ch, 1);
    add("mid.1.block", ch, ch, 0);
    idx = 0;
    for (int lv = c.unet_levels - 1; lv >= 0; --lv) {
        for (int i = 0; i <= c.unet_res; ++i) {
            const int sc = skip.back(); skip.pop_back();
            add("out." + std::to_string(idx) + ".block", ch + sc, mc * c.unet_mult[lv], is_attn(ds));
            ch = mc * c.unet_mult[lv];
            if (lv != 0 && i == c.unet_res) { add("out." + std::to_string(idx) + ".up", ch, ch, 0); ds /= 2; }
            ++idx;
        }
    }
}

int up_kernel(int r) { return 2 * r + (r % 2); }

// ------------------------------------------------------------------------------------------------ weight packing
// power of two that brings a tensor whose largest magnitude is amax to (2^(e-1), 2^e]; 1 for an empty / non-finite measurement
float h2_scale_for(float amax, int e) {
    if (!(amax > 0.f) || !std::isfinite(amax)) return 1.f;
    int ex = 0;
    const float fr = frexpf(amax, &ex);              // amax = fr 2^ex, fr in [0.5, 1)
    if (fr == 0.5f) --ex;                            // exact power of two: 2^(ex-1)
    int k = e - ex;
    k = std::max(-60, std::min(60, k));
    return ldexpf(1.f, k);
}

int split3(M* m, Wt& w) {
    if (m->f32_mfma() || !w.w) return EGR_OK;
    const int64_t ns = w.numel / ((int64_t)w.Cout * 16);
    void* p3 = nullptr;
    OKR(dev_alloc(m, (size_t)ns * 3 * w.Cout * 16 * 2, &p3));
    OKR(egr_split3_pack(w.w, p3, ns, w.Cout, m->st));
    w.w3 = p3;
    if (m->h2) {                                      // fp16 terms of w * 2^k, k from the pack's largest magnitude
        if (!m->d_wmax && hipMalloc((void**)&m->d_wmax, sizeof(float)) != hipSuccess) { set_error("hipMalloc(weight maximum) failed"); return EGR_ERR_ALLOC; }
        float wmax = 0.f;
        EGR_HIP(hipMemsetAsync(m->d_wmax, 0, sizeof(float), m->st));
        OKR(egr_absmax(w.w, w.numel, m->d_wmax, m->st));
        EGR_HIP(hipMemcpyAsync(&wmax, m->d_wmax, sizeof(float), hipMemcpyDeviceToHost, m->st));
        EGR_HIP(hipStreamSynchronize(m->st));
        w.w_scale = h2_scale_for(wmax, 13);
        void* p2 = nullptr;
        OKR(dev_alloc(m, (size_t)ns * 2 * w.Cout * 16 * 2, &p2));
        OKR(egr_split2h_pack(w.w, p2, ns, w.Cout, w.w_scale, m->st));
        w.w2 = p2;
        ++m->h2_nweights;
    }
    return EGR_OK;
}

// src in torch layout on the device; layout as in egr_pack_weight; registers key with logical (KH, KW, Cin, N)
int add_packed(M* m, const std::string& key, const float* src, int layout, int K, int N, int Ci, int Co, int KH, int KW, int logical_kh,
               int logical_kw, int logical_cin) {
    Wt w;
    const int64_t slabs = (K + 15) / 16;
    w.numel = slabs * N * 16;
    void* p = nullptr;
    OKR(dev_alloc(m, (size_t)w.numel * sizeof(float), &p));
    w.w = (float*)p;
    OKR(egr_pack_weight(src, w.w, layout, K, N, Ci, Co, KH, KW, m->st));
    w.KH = logical_kh; w.KW = logical_kw; w.Cin = logical_cin; w.Cout = N;
    if (logical_cin % 16 == 0) {
        const bool keep = m->h2;
        if (key == "mel_fb") m->h2 = false;          // spectra span > 90 dB and a log follows: the mel projection keeps three bf16 terms
        const int rc = split3(m, w);
        m->h2 = keep;
        OKR(rc);
    }
    m->W[key] = w;
    return EGR_OK;
}

int add_weight(M* m, const std::string& key, const egr_tensor_desc& t) {
    const int64_t* s = t.shape;
    if (t.ndim == 4) return add_packed(m, key, t.data, 0, (int)(s[2] * s[3] * s[1]), (int)s[0], (int)s[1], (int)s[0], (int)s[2], (int)s[3], (int)s[2], (int)s[3], (int)s[1]);
    if (key.rfind("voc.ups.", 0) == 0 && t.ndim == 3)       // ConvTranspose1d [Ci][Co][k] -> GEMM [Ci][k*Co]
        return add_packed(m, key, t.data, 1, (int)s[0], (int)(s[2] * s[1]), (int)s[0], (int)s[1], 1, (int)s[2], 1, 1, (int)s[0]);
    if (t.ndim == 3) return add_packed(m, key, t.data, 0, (int)(s[2] * s[1]), (int)s[0], (int)s[1], (int)s[0], 1, (int)s[2], 1, (int)s[2], (int)s[1]);
    return add_packed(m, key, t.data, 0, (int)s[1], (int)s[0], (int)s[1], (int)s[0], 1, 1, 1, 1, (int)s[1]);
}

int add_phases(M* m, const std::string& key, const egr_tensor_desc& t) {
    const int Co = (int)t.shape[0], Ci = (int)t.shape[1];
    float* tmp = nullptr;
    const size_t n = (size_t)4 * Co * Ci * 4;
    if (hipMalloc((void**)&tmp, n * sizeof(float)) != hipSuccess) { set_error("hipMalloc(phase weights) failed"); return EGR_ERR_ALLOC; }
    int rc = egr_phase_weights(t.data, tmp, Co, Ci, m->st);
    for (int a = 0; a < 2 && rc == EGR_OK; ++a)
        for (int b = 0; b < 2 && rc == EGR_OK; ++b)
            rc = add_packed(m, key + ".ph" + std::to_string(a) + std::to_string(b), tmp + (size_t)(2 * a + b) * Co * Ci * 4, 0, 4 * Ci, Co, Ci, Co, 2, 2, 2, 2, Ci);
    hipStreamSynchronize(m->st);
    hipFree(tmp);
    return rc;
}

const double kG2[12] = {1.0, 0.0, 0.0, 0.5, 0.5, 0.5, 0.5, -0.5, 0.5, 0.0, 0.0, 1.0};

int add_winograd(M* m, const std::string& key, const egr_tensor_desc& t, const double* G2_dev, const double* G4_dev) {
    const int Co = (int)t.shape[0], Ci = (int)t.shape[1];
    const int64_t zf = (int64_t)((Ci + 15) / 16) * Co * 16;
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1 && (m->flags & EGR_FSR_NO_WINO_F4)) break;
        const int np = pass == 0 ? 4 : 6;
        Wt w;
        w.numel = (int64_t)np * np * zf;
        w.zfloats = zf;
        w.KH = w.KW = 1; w.Cin = Ci; w.Cout = Co;
        float* pk = nullptr;
        if (hipMalloc((void**)&pk, (size_t)w.numel * sizeof(float)) != hipSuccess) { set_error("hipMalloc(Winograd U) failed"); return EGR_ERR_ALLOC; }
        int rc = egr_winograd_pack_u(t.data, pk, pass == 0 ? G2_dev : G4_dev, np, Co, Ci, m->st);
        w.w = pk;
        if (rc == EGR_OK && Ci % 16 == 0) rc = split3(m, w);
        if (rc == EGR_OK && w.w3) {                         // the fp32 pack is not needed once split
            hipStreamSynchronize(m->st);
            hipFree(pk);
            w.w = nullptr;
        } else if (rc == EGR_OK) {
            m->owned.push_back(pk);
        } else {
            hipFree(pk);
            return rc;
        }
        m->W[key + (pass == 0 ? ".wino" : ".wino4")] = w;
    }
    return EGR_OK;
}

bool contains(const std::string& s, const char* sub) { return s.find(sub) != std::string::npos; }
bool ends_with(const std::string& s, const char* suf) { const size_t n = strlen(suf); return s.size() >= n && s.compare(s.size() - n, n, suf) == 0; }

int pack_all(M* m, const egr_tensor_desc* ts, int n) {
    double g4[18];
    OKR(egr_winograd4_g(g4));
    double* Gd = nullptr;
    OKR(dev_alloc(m, 30 * sizeof(double), (void**)&Gd));
    EGR_HIP(hipMemcpyAsync(Gd, kG2, 12 * sizeof(double), hipMemcpyHostToDevice, m->st));
    EGR_HIP(hipMemcpyAsync(Gd + 12, g4, 18 * sizeof(double), hipMemcpyHostToDevice, m->st));
    const bool thin = !(m->flags & EGR_FSR_NO_THIN_ENDS);
    for (int i = 0; i < n; ++i) {
        const egr_tensor_desc& t = ts[i];
        EGR_CHECK(t.name && t.data && t.ndim >= 0 && t.ndim <= 4, EGR_ERR_ARG, "tensor %d: bad descriptor", i);
        const std::string k = t.name;
        if (k.rfind("const.", 0) == 0) continue;
        if (ends_with(k, ".weight") && t.ndim >= 2) {
            OKR(add_weight(m, k, t));
            const int64_t* s = t.shape;
            if (contains(k, ".upsample.conv.") || (k.rfind("unet.", 0) == 0 && contains(k, ".up.conv."))) OKR(add_phases(m, k, t));
            if (thin && t.ndim == 4 && s[2] * s[3] * s[0] <= 32 && s[1] % 16 == 0)       // few outputs: 1x1 contraction onto per-tap products
                OKR(add_packed(m, k + ".taps", t.data, 2, (int)s[1], (int)(s[2] * s[3] * s[0]), (int)s[1], (int)s[0], (int)s[2], (int)s[3], 1, 1, (int)s[1]));
            if (!(m->flags & EGR_FSR_NO_WINOGRAD) && t.ndim == 4 && s[2] == 3 && s[3] == 3 && std::min(s[0], s[1]) >= m->wino_min_ch &&
                !contains(k, "downsample") && !contains(k, ".down.conv") && !contains(k, "upsample") && !contains(k, ".up.conv"))
                OKR(add_winograd(m, k, t, Gd, Gd + 12));
        } else {
            Wt w;
            int64_t ne = 1;
            for (int d = 0; d < t.ndim; ++d) ne *= t.shape[d];
            w.numel = ne;
            void* p = nullptr;
            OKR(dev_alloc(m, (size_t)ne * sizeof(float), &p));
            EGR_HIP(hipMemcpyAsync(p, t.data, (size_t)ne * sizeof(float), hipMemcpyDeviceToDevice, m->st));
            w.w = (float*)p;
            m->W[k] = w;
        }
    }
    return EGR_OK;
}

// ------------------------------------------------------------------------------------------------ profiling helpers
std::string kind_of(long long Mrows, int Cin, int Cout, bool s3, bool vec = true, long long K = -1) {
    int bn = Cout > 64 ? 128 : (Cout > 32 ? 64 : 32);
    char buf[96];
    if (s3) {
        if (Cout >= 256 && Cout % 256 == 0) bn = 256;
        if (K >= 0 && (K + 15) / 16 < 32)
            while (bn > 64 && ((Mrows + 127) / 128) * ((Cout + bn - 1) / bn) < 256) bn >>= 1;
        const int bm = (bn == 128 && ((Mrows + 255) / 256) * ((Cout + 127) / 128) >= 1024) ? 256 : 128;
        snprintf(buf, sizeof(buf), "k_conv_s3<%d, %d, 1, false>", bm, bn);
    } else {
        snprintf(buf, sizeof(buf), "k_conv_igemm<%d, %s>", bn, vec ? "true" : "false");
    }
    return buf;
}

struct ProfScope {
    M* m; bool on; hipEvent_t a = nullptr, b = nullptr;
    explicit ProfScope(M* mm) : m(mm), on(mm->profiling) {
        if (on) { hipEventCreate(&a); hipEventCreate(&b); hipEventRecord(a, m->st); }
    }
    void end(const std::string& kind, double flops, const std::string& detail = std::string()) {
        if (!on) return;
        hipEventRecord(b, m->st);
        m->prof.push_back(ProfRec{kind, flops, a, b, detail});
        on = false;
    }
    ~ProfScope() { if (on) { hipEventDestroy(a); hipEventDestroy(b); } }
};

// ------------------------------------------------------------------------------------------------ operators
const void* s3_of(const M* m, const Wt* w, int Cin, const float* x) {
    if (m->f32_mfma() || !w || Cin % 16 != 0 || (((uintptr_t)x) & 15) != 0) return nullptr;
    return w->w3;
}

// One split contraction (egr_conv_s3's argument list; zfloats: floats per z problem of a stacked pack, 0 otherwise) in the operand
// scheme of the forward being enqueued.  *h2 tells the profiler which kernel family ran.
// R-entry slice of the context's pool of per-row maxima (zero at the start of the forward)
unsigned* rs_take(M* m) {
    FsrCtx* c = m->cx;
    if (c->rs_used + (size_t)m->R * EGR_ROW_AMAX_STRIDE > c->rs_cap) {
        // the start-of-forward size is a heuristic over the layer table; a graph that measures more tensors than it allowed for gets a
        // second, larger pool here (zeroed on the forward's stream) instead of a failed call -- the next forward starts at that size
        const size_t grown = std::max(2 * c->rs_cap, (size_t)64 * m->R * EGR_ROW_AMAX_STRIDE);
        unsigned* np = nullptr;
        if (hipMalloc((void**)&np, grown * sizeof(unsigned)) != hipSuccess) { set_error("FlashSR: hipMalloc(row maxima pool, %zu entries) failed", grown); return nullptr; }
        if (hipMemsetAsync(np, 0, grown * sizeof(unsigned), m->st) != hipSuccess) { hipFree(np); set_error("FlashSR: hipMemsetAsync(row maxima pool) failed"); return nullptr; }
        if (c->rs_pool) c->rs_retired.push_back(c->rs_pool);
        c->rs_pool = np; c->rs_cap = grown; c->rs_used = 0;
    }
    unsigned* p = c->rs_pool + c->rs_used;
    c->rs_used += (size_t)m->R * EGR_ROW_AMAX_STRIDE;
    return p;
}

// slice for the row maxima of a tensor an element-wise producer is about to write (null outside the fp16 scheme)
int rs_for_output(M* m, Ten& y, float** ra) {
    *ra = nullptr;
    if (!(m->h2 && m->h2_mode == 1 && m->out_amax_on) || y.numel() % m->R != 0) return EGR_OK;
    y.rs = rs_take(m);
    if (!y.rs) return EGR_ERR_ALLOC;
    *ra = (float*)y.rs;
    return EGR_OK;
}

// per-batch-row max |x| of a tensor whose leading dimension runs over the R rows of the forward (nz > 1: nz blocks zx floats
// apart, the Winograd V layout); measured once per tensor, reused by every later contraction that reads it
int row_amax_of(M* m, const float* x, int64_t numel, int nz, int64_t zx, unsigned** slot) {
    if (*slot) return EGR_OK;
    EGR_CHECK(numel % m->R == 0, EGR_ERR_ARG, "FlashSR: a tensor of %lld elements does not split into %d batch rows", (long long)numel, m->R);
    unsigned* p = rs_take(m);
    if (!p) return EGR_ERR_ALLOC;
    OKR(egr_absmax_rows(x, m->R, numel / m->R, nz, zx, (float*)p, m->st));
    *slot = p;
    return EGR_OK;
}

// y_rs (optional): the launch leaves the per-row maxima of y there (fp16 scheme, nz == 1; a fresh slice is taken when *y_rs is null
// -- the four phase launches of an up-sampling convolution share one)
int s3_launch(M* m, const Wt* w, const float* x, int64_t x_numel, unsigned** x_rs, const float* bias, const float* res, float* y, int B, int H, int W, int Cin,
              int OH, int OW, int Cout, int KH, int KW, int stride, int dil, int pad_t, int pad_l, int up2, int act, float act_param, int osy, int osx,
              int ooy, int oox, int OHF, int OWF, int nz, int64_t zx, int64_t zfloats, int64_t zy, bool* h2 = nullptr, unsigned** y_rs = nullptr) {
    const bool use = m->h2 && w->w2 && m->h2_mode == 1 && ((int64_t)B * OH * OW) % m->R == 0;
    if (h2) *h2 = use;
    if (use) {
        unsigned* local = nullptr;
        if (!x_rs) x_rs = &local;
        OKR(row_amax_of(m, x, nz > 1 ? x_numel / nz : x_numel, nz, zx, x_rs));
        float* out_ra = nullptr;
        if (y_rs && nz == 1 && m->out_amax_on) {
            if (!*y_rs) *y_rs = rs_take(m);
            if (!*y_rs) return EGR_ERR_ALLOC;
            out_ra = (float*)*y_rs;
        }
        return egr_conv_h2(x, w->w2, bias, nullptr, res, y, B, H, W, Cin, OH, OW, Cout, KH, KW, stride, dil, pad_t, pad_l, up2, act, act_param, osy,
                           osx, ooy, oox, OHF, OWF, nz, zx, zfloats * 2 / 8, zy, w->w_scale, (const float*)*x_rs, m->R, out_ra, m->st);
    }
    return egr_conv_s3(x, w->w3, bias, nullptr, res, y, B, H, W, Cin, OH, OW, Cout, KH, KW, stride, dil, pad_t, pad_l, up2, act, act_param, osy, osx,
                       ooy, oox, OHF, OWF, nz, zx, zfloats * 3 / 8, zy, m->st);
}

std::string h2_kind(std::string kind, bool h2) {      // "k_conv_s3<128, 256, 1, false>" -> "k_conv_s3<128, 256, 1, false, 1>"
    if (h2 && kind.rfind("k_conv_s3<256, 128", 0) == 0) kind.replace(0, 18, "k_conv_s3<128, 128");      // the fp16-term path has no 256 x 128 tile
    if (h2 && !kind.empty() && kind.back() == '>' && kind.rfind("k_conv", 0) == 0) { kind.pop_back(); kind += ", 1>"; }
    return kind;
}

// general convolution: weights by key (key + ".weight", bias key + ".bias") or explicit entry `wk`
int conv(M* m, Ten& y, const Ten& x, const std::string& wkey, int B, int H, int W, int Cin, int OH, int OW, int Cout, int KH, int KW,
         int stride = 1, int dil = 1, int pad_t = 0, int pad_l = 0, int up2 = 0, int act = ACT_NONE, bool bias = true,
         const float* bias_t = nullptr, const float* res = nullptr, float act_param = 0.f, const Wt* wk = nullptr) {
    OKR(new_ten(m, y, {B, OH, OW, Cout}));
    const bool want_out_amax = m->next_out_ra;
    m->next_out_ra = false;
    const Wt* w = wk ? wk : m->get(wkey + ".weight");
    EGR_CHECK(w != nullptr, EGR_ERR_ARG, "FlashSR: weight %s.weight missing", wkey.c_str());
    const float* bt = bias_t ? bias_t : (bias && !wk ? m->ptr(wkey + ".bias") : nullptr);
    const double fl = 2.0 * B * OH * OW * Cout * KH * KW * Cin;
    ProfScope ps(m);
    const void* w3 = s3_of(m, w, Cin, x.p);
    bool h2 = false;
    if (w3) {
        OKR(s3_launch(m, w, x.p, (int64_t)B * H * W * Cin, &x.rs, bt, res, y.p, B, H, W, Cin, OH, OW, Cout, KH, KW, stride, dil, pad_t, pad_l, up2, act,
                      act_param, 1, 1, 0, 0, OH, OW, 1, 0, 0, 0, &h2, want_out_amax ? &y.rs : nullptr));
    } else {
        EGR_CHECK(w->w != nullptr, EGR_ERR_ARG, "FlashSR: no fp32 pack for %s", wkey.c_str());
        OKR(egr_conv_nhwc(x.p, w->w, bt, nullptr, res, y.p, B, H, W, Cin, OH, OW, Cout, KH, KW, stride, dil, pad_t, pad_l, up2, act, act_param,
                          m->st));
    }
    if (ps.on) {
        const bool vec = Cin % 16 == 0 && (((uintptr_t)x.p) & 15) == 0;
        std::string kind = kind_of((long long)B * OH * OW, Cin, Cout, w3 != nullptr, vec, (long long)KH * KW * Cin);
        if (w3 && H == 1 && KH == 1 && KW >= 2 && stride == 1 && !up2 && OW == W && W % 128 == 0 && dil * (KW - 1) <= 50 &&
            2 * pad_l == dil * (KW - 1)) {                        // launch_conv1d_s3's conditions
            char buf[64];
            snprintf(buf, sizeof(buf), "k_conv1d_s3<%d, %d>", Cout > 64 ? 128 : (Cout > 32 ? 64 : 32), Cin % 32 == 0 ? 32 : 16);
            kind = buf;
        }
        char det[160];
        snprintf(det, sizeof(det), "%s B%d %dx%d Cin%d -> %dx%d Cout%d k%dx%d s%d d%d up%d", wkey.c_str(), B, H, W, Cin, OH, OW, Cout, KH, KW, stride, dil, up2);
        ps.end(h2_kind(kind, h2), fl, det);
    }
    if (m->count_flops) m->flops += fl;
    return EGR_OK;
}

int gn_scratch(M* m, size_t need) {
    FsrCtx* c = m->cx;
    if (c->gn_ws_bytes < need) {
        if (c->gn_ws) { hipStreamSynchronize(m->st); hipFree(c->gn_ws); c->gn_ws = nullptr; c->gn_ws_bytes = 0; }
        if (hipMalloc(&c->gn_ws, need + 1024) != hipSuccess) { set_error("hipMalloc(GroupNorm workspace) failed"); return EGR_ERR_ALLOC; }
        c->gn_ws_bytes = need + 1024;
    }
    return EGR_OK;
}

int groupnorm(M* m, Ten& y, const Ten& x, const std::string& key, float eps, bool silu) {
    const int B = (int)x.d[0], Cc = (int)x.d[x.nd - 1];
    const int HW = (int)(x.numel() / ((int64_t)B * Cc));
    const int G = m->cfg.gn_groups;
    OKR(gn_scratch(m, egr_groupnorm_workspace_bytes(B, Cc, G)));
    y.release();
    y.nd = x.nd; memcpy(y.d, x.d, sizeof(y.d));
    y.bytes = (size_t)y.numel() * 4;
    y.p = (float*)m->cx->arena.get(y.bytes);
    if (!y.p) return EGR_ERR_ALLOC;
    y.a = &m->cx->arena;
    float* ra = nullptr;
    if (B == m->R) OKR(rs_for_output(m, y, &ra));
    return egr_groupnorm_nhwc_ra(x.p, m->ptr(key + ".weight"), m->ptr(key + ".bias"), y.p, B, HW, Cc, G, eps, silu ? 1 : 0, m->cx->gn_ws, ra, m->st);
}

int gn_coeff(M* m, Ten& sc, Ten& sh, const Ten& x, const std::string& key, float eps) {
    const int B = (int)x.d[0], Cc = (int)x.d[x.nd - 1];
    const int HW = (int)(x.numel() / ((int64_t)B * Cc));
    const int G = m->cfg.gn_groups;
    OKR(gn_scratch(m, egr_groupnorm_workspace_bytes(B, Cc, G)));
    OKR(new_ten(m, sc, {B, Cc}));
    OKR(new_ten(m, sh, {B, Cc}));
    if (x.part) {        // x came out of egr_winograd4_output_stats: reduce its partials instead of re-reading x
        Ten stats;
        OKR(new_ten(m, stats, {B, G, 4}));            // [B][G][2] doubles
        OKR(egr_groupnorm_stats_from_partials(x.part->p, B, x.part_tiles, Cc, G, (double*)stats.p, m->st));
        return egr_groupnorm_coeff_from_stats((const double*)stats.p, m->ptr(key + ".weight"), m->ptr(key + ".bias"), B, HW, Cc, G, eps, sc.p, sh.p, m->st);
    }
    return egr_groupnorm_coeff(x.p, m->ptr(key + ".weight"), m->ptr(key + ".bias"), B, HW, Cc, G, eps, m->cx->gn_ws, sc.p, sh.p, m->st);
}

int conv_winograd(M* m, Ten& y, const Ten& x, const std::string& key, int act, const float* res, const float* bias_t, const float* gsc = nullptr,
                  const float* gsh = nullptr, int gsilu = 0) {
    const int B = (int)x.d[0], H = (int)x.d[1], W = (int)x.d[2], Cin = (int)x.d[3];
    const Wt* wb = m->get(key + ".weight");
    const int Cout = wb->Cout;
    const bool f4 = m->has(key + ".weight.wino4") && H % 4 == 0 && W % 4 == 0;
    const Wt* wz = m->get(key + (f4 ? ".weight.wino4" : ".weight.wino"));
    const int nz = f4 ? 36 : 16, ts = f4 ? 4 : 2;
    const int TH = (H + ts - 1) / ts, TW = (W + ts - 1) / ts;
    const int64_t P = (int64_t)B * TH * TW;
    Ten V, Mx;
    OKR(new_ten(m, V, {nz, P, Cin}));
    // fp16 operand scheme: the F(4x4) input transform leaves the per-row maxima of V as it writes it
    const bool v_h2 = m->h2 && m->h2_mode == 1 && wz && wz->w2 && Cin % 16 == 0 && B == m->R;
    if (f4 && v_h2) {
        V.rs = rs_take(m);
        if (!V.rs) return EGR_ERR_ALLOC;
        OKR(egr_winograd4_input_ra(x.p, gsc, gsh, gsilu, B, H, W, Cin, V.p, (float*)V.rs, m->st));
    } else if (f4) OKR(egr_winograd4_input(x.p, gsc, gsh, gsilu, B, H, W, Cin, V.p, m->st));
    else OKR(egr_winograd_input(x.p, gsc, gsh, gsilu, B, H, W, Cin, V.p, m->st));
    OKR(new_ten(m, Mx, {nz, P, Cout}));
    const double fl = nz * 2.0 * (double)P * Cin * Cout;
    {
        ProfScope ps(m);
        const void* w3 = s3_of(m, wz, Cin, V.p);
        bool h2 = false;
        if (w3) {
            OKR(s3_launch(m, wz, V.p, (int64_t)nz * P * Cin, &V.rs, nullptr, nullptr, Mx.p, (int)P, 1, 1, Cin, 1, 1, Cout, 1, 1, 1, 1, 0, 0, 0, 0, 0.0f, 1, 1,
                          0, 0, 1, 1, nz, P * Cin, wz->zfloats, P * Cout, &h2));
        } else {
            EGR_CHECK(wz->w != nullptr, EGR_ERR_ARG, "FlashSR: no fp32 Winograd pack for %s", key.c_str());
            OKR(egr_gemm_zbatched(V.p, wz->w, Mx.p, nz, (int)P, Cin, Cout, P * Cin, wz->zfloats, P * Cout, m->st));
        }
        if (ps.on) {
            std::string kind = kind_of(P, Cin, Cout, w3 != nullptr);
            if (w3 && Cout > 64) {          // s3_zs_nzb (csrc/egr_nn_gemm_s3.hip): z-streamed when >= 2 z per workgroup
                const int bn = (Cout >= 256 && Cout % 256 == 0) ? 256 : 128;
                const long long tiles = ((P + 127) / 128) * ((Cout + bn - 1) / bn);
                const long long groups = std::min<long long>(std::max<long long>((2048 + tiles - 1) / tiles, 1), nz);
                if ((nz + groups - 1) / groups >= 2 && !(bn == 256 && Cin > 256)) {
                    char buf[64];
                    snprintf(buf, sizeof(buf), "k_conv_s3<128, %d, 1, true>", bn);
                    kind = buf;
                }
            }
            ps.end(h2_kind(kind, h2), fl);
        }
    }
    if (m->count_flops) m->flops += fl;
    V.release();
    OKR(new_ten(m, y, {B, H, W, Cout}));
    const float* bt = bias_t ? bias_t : m->ptr(key + ".bias");
    const int G = m->cfg.gn_groups;
    const int silu = act == ACT_SILU ? 1 : 0;
    // fp16 operand scheme: the output transform leaves the per-row maxima of y (it may feed a 1x1 / strided / phase convolution)
    float* y_ra = nullptr;
    if (f4 && m->h2 && m->h2_mode == 1 && m->out_amax_on && B == m->R) {
        y.rs = rs_take(m);
        if (!y.rs) return EGR_ERR_ALLOC;
        y_ra = (float*)y.rs;
    }
    if (f4 && !(m->flags & EGR_FSR_NO_GN_PARTIALS) && Cout % G == 0 && (Cout / G) % 4 == 0) {
        auto part = std::make_shared<Ten>();
        OKR(new_ten(m, *part, {P, Cout / 4, 2}));
        OKR(egr_winograd4_output_ra(Mx.p, bt, res, y.p, B, H, W, Cout, silu, part->p, y_ra, m->st));
        y.part = part;
        y.part_tiles = TH * TW;
    } else if (f4) {
        OKR(egr_winograd4_output_ra(Mx.p, bt, res, y.p, B, H, W, Cout, silu, nullptr, y_ra, m->st));
    } else {
        OKR(egr_winograd_output(Mx.p, bt, res, y.p, B, H, W, Cout, silu, m->st));
    }
    return EGR_OK;
}

int conv_up2_phases(M* m, Ten& y, const Ten& x, const std::string& key, int act) {
    const int B = (int)x.d[0], H = (int)x.d[1], W = (int)x.d[2], Cin = (int)x.d[3];
    const int Cout = m->get(key + ".weight")->Cout;
    OKR(new_ten(m, y, {B, 2 * H, 2 * W, Cout}));
    const float* bt = m->ptr(key + ".bias");
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b) {
            const Wt* w = m->get(key + ".weight.ph" + std::to_string(a) + std::to_string(b));
            const double fl = 2.0 * B * H * W * Cout * 4 * Cin;
            ProfScope ps(m);
            const void* w3 = s3_of(m, w, Cin, x.p);
            bool h2 = false;
            if (w3)
                OKR(s3_launch(m, w, x.p, (int64_t)B * H * W * Cin, &x.rs, bt, nullptr, y.p, B, H, W, Cin, H, W, Cout, 2, 2, 1, 1, 1 - a, 1 - b, 0, act, 0.0f, 2, 2,
                              a, b, 2 * H, 2 * W, 1, 0, 0, 0, &h2, &y.rs));
            else
                OKR(egr_conv_nhwc_placed(x.p, w->w, bt, nullptr, nullptr, y.p, B, H, W, Cin, H, W, Cout, 2, 2, 1, 1, 1 - a, 1 - b, 0, act, 0.0f, 2, 2,
                                         a, b, 2 * H, 2 * W, m->st));
            if (ps.on) ps.end(h2_kind(kind_of((long long)B * H * W, Cin, Cout, w3 != nullptr, Cin % 16 == 0, 4LL * Cin), h2), fl);
            if (m->count_flops) m->flops += fl;
        }
    return EGR_OK;
}

int conv3(M* m, Ten& y, const Ten& x, const std::string& key, int stride = 1, int up2 = 0, int act = ACT_NONE, const float* res = nullptr,
          int pad = 1, const float* bias_t = nullptr) {
    const int B = (int)x.d[0], H = (int)x.d[1], W = (int)x.d[2], Cin = (int)x.d[3];
    const Wt* wb = m->get(key + ".weight");
    EGR_CHECK(wb != nullptr, EGR_ERR_ARG, "FlashSR: weight %s.weight missing", key.c_str());
    const int Cout = wb->Cout;
    const bool want_ra = m->next_out_ra;            // only the generic path below hands it to conv(); the other paths track on their own
    m->next_out_ra = false;
    if (up2 && m->has(key + ".weight.ph00") && stride == 1 && pad == 1 && !res && !bias_t) return conv_up2_phases(m, y, x, key, act);
    if (m->has(key + ".weight.wino") && !up2 && stride == 1 && pad == 1 && (act == ACT_NONE || act == ACT_SILU) && H % 2 == 0 && W % 2 == 0 &&
        Cin % 16 == 0)
        return conv_winograd(m, y, x, key, act, res, bias_t);
    const bool plain = !up2 && stride == 1 && pad == 1 && act == ACT_NONE && !res && !bias_t;
    const bool thin = !(m->flags & EGR_FSR_NO_THIN_ENDS);
    const Wt* taps = m->get(key + ".weight.taps");
    if (plain && taps && s3_of(m, taps, Cin, x.p)) {
        Ten P;
        OKR(conv(m, P, x, key, B, H, W, Cin, H, W, 9 * Cout, 1, 1, 1, 1, 0, 0, 0, ACT_NONE, false, nullptr, nullptr, 0.f, taps));
        OKR(new_ten(m, y, {B, H, W, Cout}));
        return egr_tap_gather(P.p, m->ptr(key + ".bias"), y.p, B, H, W, 3, 3, Cout, 1, 1, m->st);
    }
    if (plain && thin && Cin == 1 && Cout % 4 == 0 && 256 % (Cout / 4) == 0) {
        OKR(new_ten(m, y, {B, H, W, Cout}));
        OKR(egr_conv_cin1(x.p, wb->w, m->ptr(key + ".bias"), y.p, B, H, W, Cout, 3, 3, 1, 1, m->st));
        if (m->count_flops) m->flops += 2.0 * B * H * W * Cout * 9;
        return EGR_OK;
    }
    const int LH = up2 ? 2 * H : H, LW = up2 ? 2 * W : W;
    m->next_out_ra = want_ra;
    return conv(m, y, x, key, B, H, W, Cin, LH / stride, LW / stride, Cout, 3, 3, stride, 1, pad, pad, up2, act, true, bias_t, res);
}

int conv1x1(M* m, Ten& y, const Ten& x, const std::string& key, const float* res = nullptr, int act = ACT_NONE) {
    const int B = (int)x.d[0], H = (int)x.d[1], W = (int)x.d[2], Cin = (int)x.d[3];
    return conv(m, y, x, key, B, H, W, Cin, H, W, m->get(key + ".weight")->Cout, 1, 1, 1, 1, 0, 0, 0, act, true, nullptr, res);
}

int linear(M* m, Ten& y, const Ten& x2, const std::string& key, const float* res = nullptr, int act = ACT_NONE, bool bias = true) {
    const int rows = (int)x2.d[0], Cin = (int)x2.d[1];
    const Wt* w = m->get(key + ".weight");
    EGR_CHECK(w != nullptr, EGR_ERR_ARG, "FlashSR: weight %s.weight missing", key.c_str());
    OKR(conv(m, y, x2, key, rows, 1, 1, Cin, 1, 1, w->Cout, 1, 1, 1, 1, 0, 0, 0, act, bias, nullptr, res));
    y.view({rows, w->Cout});
    return EGR_OK;
}

int conv1d(M* m, Ten& y, const Ten& x, const std::string& key, int k, int stride = 1, int dil = 1, int pad = 0, int act = ACT_NONE,
           const float* res = nullptr) {
    const int B = (int)x.d[0], L = (int)x.d[1], Cin = (int)x.d[2];
    const int Cout = m->get(key + ".weight")->Cout;
    const int OL = (L + 2 * pad - dil * (k - 1) - 1) / stride + 1;
    OKR(conv(m, y, x, key, B, 1, L, Cin, 1, OL, Cout, 1, k, stride, dil, 0, pad, 0, act, true, nullptr, res));
    y.view({B, OL, Cout});
    return EGR_OK;
}

// conv3x3(silu(groupnorm(x))) with the normalisation fused into the Winograd input transform when the layer takes that path
int gn_conv3(M* m, Ten& y, const Ten& x, const std::string& norm_key, float eps, const std::string& conv_key, const float* res = nullptr,
             const float* bias_t = nullptr) {
    const int H = (int)x.d[1], W = (int)x.d[2], Cin = (int)x.d[3];
    // The big, thin layers (128 output channels on 512 x 256 images) as DIRECT convolutions on the input-stationary kernel with the
    // GroupNorm + SiLU in its loader (fp16 operand scheme only): as an F(4x4) pipeline they are bound by V and M round trips
    // (~19 GB per layer against 3.5 GB of activations), here x is read once and y written once.
    {
        const Wt* wd = m->get(conv_key + ".weight");
        const int B = (int)x.d[0];
        if (wd && wd->w2 && m->h2 && m->h2_mode == 1 && m->direct3x3_max_cout > 0 && wd->Cout <= m->direct3x3_max_cout && wd->KH == 3 &&
            wd->KW == 3 && B == m->R && H % 4 == 0 && W % 32 == 0 && Cin % 32 == 0 && (int64_t)(H / 4) * (W / 32) * B >= 512 &&
            Cin % m->cfg.gn_groups == 0 && (((uintptr_t)x.p) & 15) == 0) {
            Ten sc, sh;
            OKR(gn_coeff(m, sc, sh, x, norm_key, eps));
            OKR(row_amax_of(m, x.p, x.numel(), 1, 0, &x.rs));
            unsigned* bound = rs_take(m);
            if (!bound) return EGR_ERR_ALLOC;
            OKR(egr_gn_operand_bound(sc.p, sh.p, B, Cin, (const float*)x.rs, (float*)bound, m->st));
            OKR(new_ten(m, y, {B, H, W, wd->Cout}));
            float* out_ra = nullptr;
            if (m->out_amax_on) {
                y.rs = rs_take(m);
                if (!y.rs) return EGR_ERR_ALLOC;
                out_ra = (float*)y.rs;
            }
            const float* bt = bias_t ? bias_t : m->ptr(conv_key + ".bias");
            const double fl = 2.0 * B * H * W * (double)wd->Cout * 9 * Cin;
            // GroupNorm partial statistics of y from the epilogue (the next norm then reads 1/32 of y's bytes instead of y)
            void* gpart = nullptr;
            const int G = m->cfg.gn_groups;
            if (!(m->flags & EGR_FSR_NO_GN_PARTIALS) && wd->Cout % G == 0 && (wd->Cout / G) % 4 == 0) {
                auto part = std::make_shared<Ten>();
                OKR(new_ten(m, *part, {(int64_t)B * H * W / 32, wd->Cout / 4, 2}));
                gpart = part->p;
                y.part = part;
                y.part_tiles = H * W / 32;
            }
            ProfScope ps(m);
            OKR(egr_conv_h2_gn(x.p, sc.p, sh.p, 1, wd->w2, bt, res, y.p, B, H, W, Cin, wd->Cout, ACT_NONE, wd->w_scale, (const float*)bound, out_ra, gpart, m->st));
            if (ps.on) {
                ps.end(egr::conv3x3_is_name(wd->Cout, true, true), fl, conv_key);       // (the launcher's own choice: egr_nn_conv3x3.hip)
            }
            if (m->count_flops) m->flops += fl;
            return EGR_OK;
        }
    }
    const bool wino = m->has(conv_key + ".weight.wino") && H % 2 == 0 && W % 2 == 0;
    const bool fused_ok = !(m->flags & EGR_FSR_NO_FUSE_GN) && Cin % 16 == 0 && Cin % m->cfg.gn_groups == 0 && wino;
    if (!fused_ok) {
        Ten h;
        OKR(groupnorm(m, h, x, norm_key, eps, true));
        return conv3(m, y, h, conv_key, 1, 0, ACT_NONE, res, 1, bias_t);
    }
    Ten sc, sh;
    OKR(gn_coeff(m, sc, sh, x, norm_key, eps));
    return conv_winograd(m, y, x, conv_key, ACT_NONE, res, bias_t, sc.p, sh.p, 1);
}

int layernorm(M* m, Ten& y, const Ten& x2, const std::string& key) {
    OKR(new_ten(m, y, {x2.d[0], x2.d[1]}));
    float* ra = nullptr;
    if (x2.d[0] % m->R == 0) OKR(rs_for_output(m, y, &ra));
    if (!ra) return egr_layernorm_rows(x2.p, m->ptr(key + ".weight"), m->ptr(key + ".bias"), y.p, x2.d[0], (int)x2.d[1], 1e-5f, m->st);
    return egr_layernorm_rows_ra(x2.p, m->ptr(key + ".weight"), m->ptr(key + ".bias"), y.p, x2.d[0], (int)x2.d[1], 1e-5f, m->R, ra, m->st);
}

int eltwise(M* m, Ten& y, const Ten& a, const float* b, int op, float s0 = 0.f, float s1 = 0.f, bool want_ra = false) {
    y.release();
    y.nd = a.nd; memcpy(y.d, a.d, sizeof(y.d));
    y.bytes = (size_t)y.numel() * 4;
    y.p = (float*)m->cx->arena.get(y.bytes);
    if (!y.p) return EGR_ERR_ALLOC;
    y.a = &m->cx->arena;
    float* ra = nullptr;
    if (want_ra) OKR(rs_for_output(m, y, &ra));
    if (!ra) return egr_eltwise(a.p, b, y.p, a.numel(), op, s0, s1, m->st);
    return egr_eltwise_ra(a.p, b, y.p, a.numel(), op, s0, s1, m->R, ra, m->st);
}

// q, k, v [B*T][C] -> [B*T][C]: softmax(q k^T / sqrt(d)) v per head
int attention(M* m, Ten& o, const Ten& q, const Ten& k, const Ten& v, int B, int T, int Cc, int heads) {
    const int d = Cc / heads;
    Ten S;
    OKR(new_ten(m, S, {B, heads, T, T}));
    const bool s3 = !m->f32_mfma() && d % 16 == 0 && T % 16 == 0 && Cc % 4 == 0;
    const float scale = 1.0f / sqrtf((float)d);
    const float sc = (float)pow((double)d, -0.5);
    (void)scale;
    // fp16 operand scheme: both products on two fp16 terms, every operand scaled per batch row from its own maximum (q / k / v carry
    // theirs from the epilogues of their projections; the soft-max output lies in [0, 1]: a constant)
    const bool h2 = s3 && m->h2 && m->h2_mode == 1 && B == m->R;
    if (h2) {
        OKR(row_amax_of(m, q.p, q.numel(), 1, 0, &q.rs));
        OKR(row_amax_of(m, k.p, k.numel(), 1, 0, &k.rs));
        OKR(row_amax_of(m, v.p, v.numel(), 1, 0, &v.rs));
        OKR(egr_bgemm_nt_h2(q.p, k.p, S.p, B, heads, T, T, d, Cc, Cc, T, (int64_t)T * Cc, d, (int64_t)T * Cc, d, (int64_t)heads * T * T,
                            (int64_t)T * T, sc, (const float*)q.rs, (const float*)k.rs, nullptr, m->st));
    } else if (s3)
        OKR(egr_bgemm_nt_s3(q.p, k.p, S.p, B, heads, T, T, d, Cc, Cc, T, (int64_t)T * Cc, d, (int64_t)T * Cc, d, (int64_t)heads * T * T,
                            (int64_t)T * T, sc, m->st));
    else
        OKR(egr_bgemm(q.p, k.p, S.p, B, heads, T, T, d, Cc, Cc, T, (int64_t)T * Cc, d, (int64_t)T * Cc, d, (int64_t)heads * T * T, (int64_t)T * T, 1,
                      sc, m->st));
    OKR(egr_softmax_rows(S.p, (int64_t)B * heads * T, T, m->st));
    OKR(new_ten(m, o, {(int64_t)B * T, Cc}));
    if (s3) {
        Ten vt;
        OKR(new_ten(m, vt, {B, Cc, T}));
        OKR(egr_transpose_batched(v.p, vt.p, B, T, Cc, m->st));
        if (h2) {
            float* o_ra = nullptr;                 // o feeds the output projection: its row maxima ride along
            OKR(rs_for_output(m, o, &o_ra));
            OKR(egr_bgemm_nt_h2(S.p, vt.p, o.p, B, heads, T, d, T, T, T, Cc, (int64_t)heads * T * T, (int64_t)T * T, (int64_t)Cc * T, (int64_t)d * T,
                                (int64_t)T * Cc, d, 1.0f, (const float*)m->cx->rs_ones, (const float*)v.rs, o_ra, m->st));
        } else
        OKR(egr_bgemm_nt_s3(S.p, vt.p, o.p, B, heads, T, d, T, T, T, Cc, (int64_t)heads * T * T, (int64_t)T * T, (int64_t)Cc * T, (int64_t)d * T,
                            (int64_t)T * Cc, d, 1.0f, m->st));
    } else {
        OKR(egr_bgemm(S.p, v.p, o.p, B, heads, T, d, T, T, Cc, Cc, (int64_t)heads * T * T, (int64_t)T * T, (int64_t)T * Cc, d, (int64_t)T * Cc, d, 0,
                      1.0f, m->st));
    }
    if (m->count_flops) m->flops += 4.0 * B * heads * (double)T * T * d;
    return EGR_OK;
}

int snake(M* m, Ten& y, const Ten& x, const std::string& akey, const std::string& bkey) {
    OKR(new_ten(m, y, {x.d[0], x.d[1], x.d[2]}));
    float* ra = nullptr;                           // fp16 operand scheme: y feeds a 1-D convolution; its row maxima ride along
    if (m->h2 && m->h2_mode == 1 && (int)x.d[0] == m->R) {
        y.rs = rs_take(m);
        if (!y.rs) return EGR_ERR_ALLOC;
        ra = (float*)y.rs;
    }
    return egr_snake_aa_ra(x.p, m->ptr(akey), m->ptr(bkey), m->filt, y.p, (int)x.d[0], (int)x.d[1], (int)x.d[2], m->cfg.aa_taps, ra, m->st);
}

int concat(M* m, Ten& y, const Ten& a, const Ten& b) {
    const int64_t rows = a.d[0] * a.d[1] * a.d[2];
    OKR(new_ten(m, y, {a.d[0], a.d[1], a.d[2], a.d[3] + b.d[3]}));
    float* ra = nullptr;
    if ((int)a.d[0] == m->R) OKR(rs_for_output(m, y, &ra));
    if (!ra) return egr_concat_channels(a.p, b.p, y.p, rows, (int)a.d[3], (int)b.d[3], m->st);
    return egr_concat_channels_ra(a.p, b.p, y.p, rows, (int)a.d[3], (int)b.d[3], m->R, ra, m->st);
}

// ------------------------------------------------------------------------------------------------ constant sub-graph
// time embedding at t = T-1: silu(MLP(emb)) is shared by all res-blocks; each block's projection is folded into its conv bias
int fold_time_embedding(M* m, const float* emb_dev) {
    const egr_flashsr_config& c = m->cfg;
    Ten emb, t1, t2;
    OKR(new_ten(m, emb, {1, c.unet_ch}));
    EGR_HIP(hipMemcpyAsync(emb.p, emb_dev, (size_t)c.unet_ch * 4, hipMemcpyDeviceToDevice, m->st));
    OKR(linear(m, t1, emb, "unet.time_embed.0", nullptr, ACT_SILU));
    OKR(linear(m, t2, t1, "unet.time_embed.2", nullptr, ACT_SILU));
    for (size_t i = 0; i < m->blk_name.size(); ++i) {
        const std::string& name = m->blk_name[i];
        if (!ends_with(name, ".block")) continue;
        const std::string base = "unet." + name;
        Ten e;
        OKR(linear(m, e, t2, base + ".res.emb"));
        const Wt* b = m->get(base + ".res.in_conv.bias");
        EGR_CHECK(b != nullptr, EGR_ERR_ARG, "FlashSR: %s.res.in_conv.bias missing", base.c_str());
        Wt bt;
        bt.numel = b->numel;
        void* p = nullptr;
        OKR(dev_alloc(m, (size_t)b->numel * 4, &p));
        bt.w = (float*)p;
        OKR(egr_eltwise(b->w, e.p, bt.w, b->numel, EW_ADD, 0.f, 0.f, m->st));
        m->W[base + ".res.in_conv.bias_t"] = bt;
    }
    EGR_HIP(hipStreamSynchronize(m->st));
    return EGR_OK;
}

// ------------------------------------------------------------------------------------------------ stages
int log_mel(M* m, Ten& mel, const float* x, int B, int L) {
    const egr_flashsr_config& c = m->cfg;
    const int rpad = (c.n_fft - c.hop) / 2;
    const int t_valid = std::min(c.n_frames, (L + 2 * rpad - c.n_fft) / c.hop + 1);
    Ten mag;
    OKR(new_ten(m, mag, {B, c.n_frames, m->ldm}));
    OKR(egr_stft_frames(x, B, L, c.n_fft, c.hop, rpad, c.n_frames, t_valid, m->ldm, m->window, mag.p, m->st));
    mag.view({(int64_t)B * c.n_frames, 1, 1, m->ldm});
    OKR(conv(m, mel, mag, "mel_fb", B * c.n_frames, 1, 1, m->ldm, 1, 1, c.n_mels, 1, 1, 1, 1, 0, 0, 0, ACT_LOGCLAMP, false, nullptr, nullptr,
             c.log_floor, m->get("mel_fb")));
    mel.view({B, c.n_frames, c.n_mels, 1});
    return EGR_OK;
}

// lowpass_input=True: cutoff from the STFT energy, zero-phase 8th-order Chebyshev-I gain applied on the Fat-Llama transform passes
int lowpass(M* m, Ten& y, const float* x, int B, int L) {
    const egr_flashsr_config& c = m->cfg;
    const int rpad = (c.n_fft - c.hop) / 2;
    const int T = (L + 2 * rpad - c.n_fft) / c.hop + 1;
    const int nb = c.n_fft / 2 + 1;
    Ten mag, cut, gain;
    OKR(new_ten(m, mag, {B, T, m->ldm}));
    OKR(egr_stft_frames(x, B, L, c.n_fft, c.hop, rpad, T, T, m->ldm, m->window, mag.p, m->st));
    OKR(new_ten(m, cut, {B}));
    const int64_t nbins = L / 2 + 1;
    OKR(new_ten(m, gain, {B, nbins}));
    OKR(egr_lowpass_gain(mag.p, B, T, m->ldm, nb, 0.985f, (float)c.sr, 8, 0.05f, nbins, (int*)cut.p, gain.p, m->st));
    egr_fatllama_plan*& plan = m->cx->lp_plans[B];
    if (!plan) OKR(egr_fatllama_plan_create(&plan, L, B, 1, 0, 0));
    OKR(new_ten(m, y, {B, L}));
    return egr_spectral_gain(plan, x, gain.p, y.p, m->st);
}

int vae_res(M* m, Ten& y, const Ten& x, const std::string& name) {
    Ten h, sc;
    OKR(gn_conv3(m, h, x, name + ".norm1", 1e-6f, name + ".conv1"));
    const float* scp = x.p;
    if (m->has(name + ".nin_shortcut.weight")) { OKR(conv1x1(m, sc, x, name + ".nin_shortcut")); scp = sc.p; }
    return gn_conv3(m, y, h, name + ".norm2", 1e-6f, name + ".conv2", scp);
}

int vae_attn(M* m, Ten& y, const Ten& x, const std::string& name) {
    const int B = (int)x.d[0], H = (int)x.d[1], W = (int)x.d[2], Cc = (int)x.d[3];
    Ten h, q, k, v, o;
    OKR(groupnorm(m, h, x, name + ".norm", 1e-6f, false));
    m->next_out_ra = true;
    OKR(conv1x1(m, q, h, name + ".q"));
    m->next_out_ra = true;
    OKR(conv1x1(m, k, h, name + ".k"));
    m->next_out_ra = true;
    OKR(conv1x1(m, v, h, name + ".v"));
    OKR(attention(m, o, q, k, v, B, H * W, Cc, 1));
    o.view({B, H, W, Cc});
    m->next_out_ra = true;
    return conv1x1(m, y, o, name + ".proj_out", x.p);
}

int vae_encode(M* m, Ten& z, const Ten& mel) {
    const egr_flashsr_config& c = m->cfg;
    Ten h, t;
    OKR(conv3(m, h, mel, "vae.encoder.conv_in"));
    for (int lv = 0; lv < c.vae_levels; ++lv) {
        for (int b = 0; b < c.vae_res; ++b) {
            OKR(vae_res(m, t, h, "vae.encoder.down." + std::to_string(lv) + ".block." + std::to_string(b)));
            h = std::move(t);
        }
        if (lv != c.vae_levels - 1) {
            m->next_out_ra = true;
            OKR(conv3(m, t, h, "vae.encoder.down." + std::to_string(lv) + ".downsample.conv", 2, 0, ACT_NONE, nullptr, 0));
            h = std::move(t);
        }
    }
    OKR(vae_res(m, t, h, "vae.encoder.mid.block_1")); h = std::move(t);
    OKR(vae_attn(m, t, h, "vae.encoder.mid.attn_1")); h = std::move(t);
    OKR(vae_res(m, t, h, "vae.encoder.mid.block_2")); h = std::move(t);
    OKR(groupnorm(m, t, h, "vae.encoder.norm_out", 1e-6f, true));
    OKR(conv3(m, h, t, "vae.encoder.conv_out"));
    Ten mom;
    OKR(conv1x1(m, mom, h, "vae.quant_conv"));
    // the mean half of the moments: channels [0, z_ch) of 2 z_ch
    const int64_t rows = mom.d[0] * mom.d[1] * mom.d[2];
    OKR(new_ten(m, z, {mom.d[0], mom.d[1], mom.d[2], c.z_ch}));
    EGR_HIP(hipMemcpy2DAsync(z.p, (size_t)c.z_ch * 4, mom.p, (size_t)2 * c.z_ch * 4, (size_t)c.z_ch * 4, (size_t)rows, hipMemcpyDeviceToDevice, m->st));
    return EGR_OK;
}

int vae_decode(M* m, Ten& y, const Ten& z) {
    const egr_flashsr_config& c = m->cfg;
    Ten h, t;
    m->next_out_ra = true;
    OKR(conv1x1(m, t, z, "vae.post_quant_conv"));
    OKR(conv3(m, h, t, "vae.decoder.conv_in"));
    OKR(vae_res(m, t, h, "vae.decoder.mid.block_1")); h = std::move(t);
    OKR(vae_attn(m, t, h, "vae.decoder.mid.attn_1")); h = std::move(t);
    OKR(vae_res(m, t, h, "vae.decoder.mid.block_2")); h = std::move(t);
    for (int lv = c.vae_levels - 1; lv >= 0; --lv) {
        for (int b = 0; b <= c.vae_res; ++b) {
            OKR(vae_res(m, t, h, "vae.decoder.up." + std::to_string(lv) + ".block." + std::to_string(b)));
            h = std::move(t);
        }
        if (lv != 0) {
            OKR(conv3(m, t, h, "vae.decoder.up." + std::to_string(lv) + ".upsample.conv", 1, 1));
            h = std::move(t);
        }
    }
    OKR(groupnorm(m, t, h, "vae.decoder.norm_out", 1e-6f, true));
    return conv3(m, y, t, "vae.decoder.conv_out");
}

int unet_block(M* m, Ten& y, const Ten& x, const std::string& base, bool has_attn) {
    const egr_flashsr_config& c = m->cfg;
    Ten h, sc, r;
    OKR(gn_conv3(m, h, x, base + ".res.in_norm", 1e-5f, base + ".res.in_conv", nullptr, m->ptr(base + ".res.in_conv.bias_t")));
    const float* scp = x.p;
    if (m->has(base + ".res.skip.weight")) { OKR(conv1x1(m, sc, x, base + ".res.skip")); scp = sc.p; }
    OKR(gn_conv3(m, r, h, base + ".res.out_norm", 1e-5f, base + ".res.out_conv", scp));
    h.release(); sc.release();
    if (!has_attn) { y = std::move(r); return EGR_OK; }
    const int B = (int)r.d[0], H = (int)r.d[1], W = (int)r.d[2], Cc = (int)r.d[3];
    const int T = H * W, heads = Cc / c.head_dim;
    Ten g0, t;
    OKR(groupnorm(m, g0, r, base + ".st.norm", 1e-6f, false));
    OKR(conv1x1(m, t, g0, base + ".st.proj_in"));
    g0.release();
    t.view({(int64_t)B * T, Cc});
    for (int a = 1; a <= 2; ++a) {
        const std::string an = base + ".st.attn" + std::to_string(a);
        Ten n_, q, k, v, o, t2;
        OKR(layernorm(m, n_, t, an + "_ln"));
        m->next_out_ra = true;                         // q, k, v feed the attention products: their row maxima ride along
        OKR(linear(m, q, n_, an + ".to_q", nullptr, ACT_NONE, false));
        m->next_out_ra = true;
        OKR(linear(m, k, n_, an + ".to_k", nullptr, ACT_NONE, false));
        m->next_out_ra = true;
        OKR(linear(m, v, n_, an + ".to_v", nullptr, ACT_NONE, false));
        OKR(attention(m, o, q, k, v, B, T, Cc, heads));
        OKR(linear(m, t2, o, an + ".to_out", t.p));
        t = std::move(t2);
    }
    Ten ln, u, g, t3;
    OKR(layernorm(m, ln, t, base + ".st.ff_ln"));
    OKR(linear(m, u, ln, base + ".st.ff.geglu"));
    OKR(new_ten(m, g, {(int64_t)B * T, 4 * Cc}));
    {
        float* ra = nullptr;
        if (B == m->R) OKR(rs_for_output(m, g, &ra));
        if (ra) OKR(egr_geglu_ra(u.p, g.p, (int64_t)B * T, 4 * Cc, m->R, ra, m->st));
        else OKR(egr_geglu(u.p, g.p, (int64_t)B * T, 4 * Cc, m->st));
    }
    m->next_out_ra = true;
    OKR(linear(m, t3, g, base + ".st.ff.out", t.p));
    t3.view({B, H, W, Cc});
    m->next_out_ra = true;
    return conv1x1(m, y, t3, base + ".st.proj_out", r.p);
}

int unet(M* m, Ten& out, Ten&& x0) {
    std::vector<Ten> skips;
    Ten h = std::move(x0), t;
    for (size_t i = 0; i < m->blk_name.size(); ++i) {
        const std::string& name = m->blk_name[i];
        const std::string base = "unet." + name;
        const std::string part = name.substr(0, name.find('.'));
        const std::string kind = name.substr(name.rfind('.') + 1);
        if (kind == "conv_in") {
            OKR(conv3(m, t, h, base));
            Ten cp = t.alias();                        // the skip stack owns the tensor, h views it
            skips.push_back(std::move(t));
            h = std::move(cp);
        } else if (kind == "down") {
            m->next_out_ra = true;
            OKR(conv3(m, t, h, base + ".conv", 2, 0, ACT_NONE, nullptr, 1));
            Ten cp = t.alias();
            skips.push_back(std::move(t));
            h = std::move(cp);
        } else if (kind == "up") {
            OKR(conv3(m, t, h, base + ".conv", 1, 1));
            h = std::move(t);
        } else {
            if (part == "out") {
                Ten cat;
                OKR(concat(m, cat, h, skips.back()));
                skips.pop_back();
                h = std::move(cat);
            }
            OKR(unet_block(m, t, h, base, m->blk_attn[i] != 0));
            if (part == "in") {
                Ten cp = t.alias();
                skips.push_back(std::move(t));
                h = std::move(cp);
            } else {
                h = std::move(t);
            }
        }
    }
    OKR(groupnorm(m, t, h, "unet.out_norm", 1e-5f, true));
    return conv3(m, out, t, "unet.out_conv");
}

int amp(M* m, Ten& y, Ten&& h, int j) {
    const egr_flashsr_config& c = m->cfg;
    Ten acc;
    for (int ki = 0; ki < c.voc_n_kernels; ++ki) {
        const int k = c.voc_kernels[ki];
        Ten x;                                   // aliases h on the first dilation
        const Ten* cur = &h;
        for (int di = 0; di < c.voc_n_dils; ++di) {
            const int d = c.voc_dils[di];
            const std::string b = "voc.amp." + std::to_string(j) + "." + std::to_string(ki) + "." + std::to_string(di);
            Ten xt, xc, xs, xn;
            // the 16- and 32-channel stages (245 760 / 122 880 samples per row): the unit's four launches are pure streaming there -- one fused kernel
            // keeps the L-tile on the CU (csrc/egr_nn_amp.hip); fp16 operand scheme only
            {
                const Wt* w1 = m->get(b + ".conv1.weight");
                const Wt* w2 = m->get(b + ".conv2.weight");
                const int Cc = (int)cur->d[2];
                if (m->fused_amp && m->h2 && m->h2_mode == 1 && w1 && w2 && w1->w2 && w2->w2 && Cc == 16 && w1->Cout == Cc && w2->Cout == Cc && (k & 1) &&
                    k <= 11 && d * (k - 1) / 2 <= 25 && c.aa_taps == 12 && cur->d[1] >= 16) {
                    OKR(new_ten(m, xn, {cur->d[0], cur->d[1], cur->d[2]}));
                    ProfScope ps(m);
                    OKR(egr_amp_unit_h2(cur->p, xn.p, (int)cur->d[0], (int)cur->d[1], Cc, k, d, m->ptr(b + ".alpha1"), m->ptr(b + ".beta1"), w1->w2, w1->w_scale,
                                        m->ptr(b + ".conv1.bias"), m->ptr(b + ".alpha2"), m->ptr(b + ".beta2"), w2->w2, w2->w_scale, m->ptr(b + ".conv2.bias"),
                                        m->filt, c.aa_taps, m->st));
                    const double fl = 2.0 * 2.0 * (double)cur->d[0] * cur->d[1] * Cc * Cc * k;
                    if (ps.on) ps.end(Cc == 16 ? "k_amp_unit<16>" : "k_amp_unit<32>", fl, b);
                    if (m->count_flops) m->flops += fl;
                    x = std::move(xn);
                    cur = &x;
                    continue;
                }
            }
            OKR(snake(m, xt, *cur, b + ".alpha1", b + ".beta1"));
            OKR(conv1d(m, xc, xt, b + ".conv1", k, 1, d, d * (k - 1) / 2));
            xt.release();
            OKR(snake(m, xs, xc, b + ".alpha2", b + ".beta2"));
            xc.release();
            OKR(conv1d(m, xn, xs, b + ".conv2", k, 1, 1, (k - 1) / 2, ACT_NONE, cur->p));
            x = std::move(xn);
            cur = &x;
        }
        if (ki == 0) {
            acc = std::move(x);
        } else if (ki + 1 < c.voc_n_kernels) {
            Ten s;
            OKR(eltwise(m, s, acc, x.p, EW_ADD));
            acc = std::move(s);
        } else {                                   // last branch: the mean's scale rides on the last add
            return eltwise(m, y, acc, x.p, EW_ADD_SCALE, 1.0f / c.voc_n_kernels, 0.f, true);   // feeds the next stage's up-sampling GEMM
        }
    }
    return eltwise(m, y, acc, nullptr, EW_SCALE, 1.0f / c.voc_n_kernels, 0.f, true);
}

int vocoder(M* m, Ten& y, const Ten& mel_hat, const float* wave, int B) {
    const egr_flashsr_config& c = m->cfg;
    const int T = (int)mel_hat.d[1], Fm = (int)mel_hat.d[2];
    const int n = c.voc_n_rates;
    std::vector<Ten> feats(n);
    {
        Ten e0;                                     // non-owning view of the input rows
        e0.view({B, c.chunk, 1});
        e0.p = const_cast<float*>(wave);
        const Ten* e = &e0;
        for (int i = 0; i < n; ++i) {
            const int r = c.voc_rates[n - 1 - i];
            m->next_out_ra = i + 1 < n;                 // feeds the next strided convolution of the encoder
            OKR(conv1d(m, feats[i], *e, "voc.wave_enc." + std::to_string(i), 2 * r + 1, r, 1, r, ACT_LEAKY));
            e = &feats[i];
        }
    }
    Ten mh;                                        // [B][T][Fm] view of mel_hat
    mh.view({B, T, Fm});
    mh.p = mel_hat.p;
    Ten h;
    m->next_out_ra = true;
    OKR(conv1d(m, h, mh, "voc.conv_pre", 7, 1, 1, 3, ACT_NONE, feats[n - 1].p));
    feats[n - 1].release();
    for (int j = 0; j < n; ++j) {
        const int r = c.voc_rates[j];
        const int kt = up_kernel(r);
        const int Bc = (int)h.d[0], Lin = (int)h.d[1], Ci = (int)h.d[2];
        const Wt* wt = m->get("voc.ups." + std::to_string(j) + ".weight");
        const int Co = wt->Cout / kt;
        Ten Y, out, hx;
        hx.view({(int64_t)Bc * Lin, 1, 1, Ci});
        hx.p = h.p;
        hx.rs = h.rs;                                  // (a view: the row maxima of h are its own)
        OKR(conv(m, Y, hx, "voc.ups." + std::to_string(j), Bc * Lin, 1, 1, Ci, 1, 1, kt * Co, 1, 1, 1, 1, 0, 0, 0, ACT_NONE, false, nullptr, nullptr, 0.f, wt));
        OKR(new_ten(m, out, {Bc, (int64_t)Lin * r, Co}));
        const float* add = (j <= n - 2) ? feats[n - 2 - j].p : nullptr;
        OKR(egr_col2im_convtr1d(Y.p, m->ptr("voc.ups." + std::to_string(j) + ".bias"), add, out.p, Bc, Lin, Lin * r, Co, kt, r, (kt - r) / 2, m->st));
        Y.release();
        if (j <= n - 2) feats[n - 2 - j].release();
        Ten nx;
        OKR(amp(m, nx, std::move(out), j));
        h = std::move(nx);
    }
    Ten s;
    OKR(snake(m, s, h, "voc.post.alpha", "voc.post.beta"));
    OKR(conv1d(m, y, s, "voc.conv_post", 7, 1, 1, 3, ACT_TANH));
    y.view({B, y.d[1]});
    return EGR_OK;
}

int copy_out(M* m, float* dst, const Ten& t) {
    if (!dst) return EGR_OK;
    EGR_HIP(hipMemcpyAsync(dst, t.p, (size_t)t.numel() * 4, hipMemcpyDeviceToDevice, m->st));
    return EGR_OK;
}

// x [R][chunk], noise [R][h][w][z] (channels-last) -> y [R][chunk]; stages (optional): mel, z_cond, v, z0, mel_hat, y_full
int forward(M* m, const float* x_in, const float* noise, int R, int lowpass_on, float* y_out, float* const* stages) {
    const egr_flashsr_config& c = m->cfg;
    const float* x = x_in;
    m->R = R;
    if (m->h2_mode == 1) {                         // per-row operand maxima of this forward: one zeroed pool per context
        FsrCtx* cx = m->cx;
        size_t slices = (size_t)(2 * m->h2_nweights + 64);
        if (const char* e = getenv("EGR_FSR_RS_POOL_SLICES")) { const int v = atoi(e); if (v >= 1) slices = (size_t)v; }   // test knob: start small, exercise the growth in rs_take
        const size_t need = slices * (size_t)R * EGR_ROW_AMAX_STRIDE;
        if (!cx->rs_retired.empty()) {                 // pools a previous forward outgrew (rs_take): its kernels are done once the stream is
            hipStreamSynchronize(m->st);
            for (unsigned* q : cx->rs_retired) hipFree(q);
            cx->rs_retired.clear();
        }
        if (cx->rs_cap < need) {
            if (cx->rs_pool) { hipStreamSynchronize(m->st); hipFree(cx->rs_pool); cx->rs_pool = nullptr; cx->rs_cap = 0; }
            if (hipMalloc((void**)&cx->rs_pool, need * sizeof(unsigned)) != hipSuccess) { set_error("hipMalloc(row maxima pool) failed"); return EGR_ERR_ALLOC; }
            cx->rs_cap = need;
        }
        EGR_HIP(hipMemsetAsync(cx->rs_pool, 0, cx->rs_cap * sizeof(unsigned), m->st));
        cx->rs_used = 0;
        if (cx->rs_ones_rows < R) {
            if (cx->rs_ones) { hipStreamSynchronize(m->st); hipFree(cx->rs_ones); cx->rs_ones = nullptr; }
            const size_t n = (size_t)R * EGR_ROW_AMAX_STRIDE;
            if (hipMalloc((void**)&cx->rs_ones, n * sizeof(unsigned)) != hipSuccess) { set_error("hipMalloc(row maxima constants) failed"); return EGR_ERR_ALLOC; }
            std::vector<unsigned> ones(n, 0x3f800000u);
            EGR_HIP(hipMemcpy(cx->rs_ones, ones.data(), n * sizeof(unsigned), hipMemcpyHostToDevice));
            cx->rs_ones_rows = R;
        }
    }
    // EGR_FSR_TRACE=2 (diagnosis of first-call costs): synchronise after every stage and print host / device-complete times
    static const int trace_lv = getenv("EGR_FSR_TRACE") ? atoi(getenv("EGR_FSR_TRACE")) : 0;
    const bool trace2 = trace_lv >= 2;
    const double t_fwd = now_ms();
    auto stage_mark = [&](const char* what) {
        if (trace_lv == 1) fprintf(stderr, "[egr_flashsr forward R=%d] %-10s enqueued at %8.1f ms\n", R, what, now_ms() - t_fwd);
        if (!trace2) return;
        const double t_host = now_ms() - t_fwd;
        hipStreamSynchronize(m->st);
        fprintf(stderr, "[egr_flashsr forward R=%d] %-10s enqueued at %8.1f ms, complete at %8.1f ms (arena hipMalloc so far %.1f ms)\n", R, what, t_host,
                now_ms() - t_fwd, 1e-3 * (double)g_arena_malloc_us.load());
    };
    Ten xl;
    if (lowpass_on) { OKR(lowpass(m, xl, x_in, R, c.chunk)); x = xl.p; }
    Ten mel, z_c, v, z0, mel_hat, y;
    OKR(log_mel(m, mel, x, R, c.chunk));
    stage_mark("log_mel");
    OKR(vae_encode(m, z_c, mel));
    stage_mark("vae_encode");
    {
        Ten nz, cat;                               // non-owning view of the caller's noise
        nz.view({R, m->lat_h, m->lat_w, c.z_ch});
        nz.p = const_cast<float*>(noise);
        OKR(concat(m, cat, nz, z_c));
        OKR(unet(m, v, std::move(cat)));
        OKR(eltwise(m, z0, nz, v.p, EW_AXPBY, m->alpha, -m->sigma));
    }
    stage_mark("unet");
    OKR(vae_decode(m, mel_hat, z0));
    stage_mark("vae_decode");
    OKR(vocoder(m, y, mel_hat, x, R));
    stage_mark("vocoder");
    if (stages) {
        OKR(copy_out(m, stages[0], mel)); OKR(copy_out(m, stages[1], z_c)); OKR(copy_out(m, stages[2], v));
        OKR(copy_out(m, stages[3], z0)); OKR(copy_out(m, stages[4], mel_hat)); OKR(copy_out(m, stages[5], y));
    }
    const int64_t Ly = y.d[1];
    EGR_CHECK(Ly >= c.chunk, EGR_ERR_UNSUPPORTED, "FlashSR: vocoder output %lld shorter than the chunk %d", (long long)Ly, c.chunk);
    EGR_HIP(hipMemcpy2DAsync(y_out, (size_t)c.chunk * 4, y.p, (size_t)Ly * 4, (size_t)c.chunk * 4, (size_t)R, hipMemcpyDeviceToDevice, m->st));
    return EGR_OK;
}

// ------------------------------------------------------------------------------------------------ one forward at a time per device
// A handle serves ONE caller stream at a time: its contexts' scratch arenas are reused in stream order, so forwards that arrive on
// different streams of one device are chained by an event (whatever streams the host -- ComfyUI nodes, threads -- uses).  Inside
// a call the handle runs up to max_groups row groups concurrently on its own verified side streams (egr_flashsr_infer).
// History: the chain was introduced in round 1 against wrong STFT bins next to a foreign k_conv_s3; that was root-caused in
// round 2 to a gfx950 packed-fp32 op_sel erratum and removed at the source (csrc/Makefile NOPK, tests/test_isa_audit.py,
// DESIGN.md section 4.4a), so the guard is about arena ownership only.  EGR_FSR_NO_STREAM_GUARD=1 removes the chain (probes).
// The lock is PER DEVICE: host threads that drive different GPUs of one process (flashsr_engine.infer_spans_devices) enqueue
// their forwards side by side.
struct ForwardGuard {
    struct Dev { std::mutex fwd; hipStream_t last_st = nullptr; hipEvent_t ev = nullptr; };
    static Dev& of(int device) {
        static std::mutex m;
        static std::map<int, std::unique_ptr<Dev>> devs;
        std::lock_guard<std::mutex> g(m);
        auto& p = devs[device];
        if (!p) p.reset(new Dev());
        return *p;
    }
    Dev& d;
    std::unique_lock<std::mutex> lk;
    hipStream_t st; bool on;
    ForwardGuard(int device, hipStream_t s) : d(of(device)), lk(d.fwd), st(s) {
        static const bool off = getenv("EGR_FSR_NO_STREAM_GUARD") && atoi(getenv("EGR_FSR_NO_STREAM_GUARD")) != 0;
        on = !off;
        if (!on) return;
        if (d.ev && d.last_st != st) hipStreamWaitEvent(st, d.ev, 0);
    }
    ~ForwardGuard() {
        if (!on) return;
        if (!d.ev) hipEventCreateWithFlags(&d.ev, hipEventDisableTiming);
        hipEventRecord(d.ev, st);
        d.last_st = st;
    }
};

const egr_tensor_desc* find(const egr_tensor_desc* ts, int n, const char* name) {
    for (int i = 0; i < n; ++i) if (ts[i].name && strcmp(ts[i].name, name) == 0) return &ts[i];
    return nullptr;
}

}  // namespace

// ==================================================================================================== C ABI
extern "C" int egr_flashsr_default_config(egr_flashsr_config* c) {
    EGR_CHECK(c != nullptr, EGR_ERR_ARG, "config is null");
    memset(c, 0, sizeof(*c));
    c->struct_bytes = (int)sizeof(*c);
    c->sr = 48000; c->chunk = 245760; c->n_fft = 2048; c->hop = 480; c->n_mels = 256; c->n_frames = 512;
    c->fmin = 20.f; c->fmax = 24000.f; c->log_floor = 1e-5f;
    c->vae_ch = 128; c->vae_levels = 4; c->vae_mult[0] = 1; c->vae_mult[1] = 2; c->vae_mult[2] = 4; c->vae_mult[3] = 8; c->vae_res = 2;
    c->z_ch = 16; c->gn_groups = 32;
    c->unet_ch = 128; c->unet_levels = 4; c->unet_mult[0] = 1; c->unet_mult[1] = 2; c->unet_mult[2] = 3; c->unet_mult[3] = 5; c->unet_res = 2;
    c->unet_n_attn = 3; c->unet_attn_ds[0] = 2; c->unet_attn_ds[1] = 4; c->unet_attn_ds[2] = 8; c->head_dim = 32; c->t_steps = 1000;
    c->voc_ch = 512; c->voc_n_rates = 5; const int r[5] = {6, 5, 4, 2, 2}; memcpy(c->voc_rates, r, sizeof(r));
    c->voc_n_kernels = 3; c->voc_kernels[0] = 3; c->voc_kernels[1] = 7; c->voc_kernels[2] = 11;
    c->voc_n_dils = 3; c->voc_dils[0] = 1; c->voc_dils[1] = 3; c->voc_dils[2] = 5; c->aa_taps = 12;
    return EGR_OK;
}

extern "C" int egr_flashsr_destroy(egr_flashsr* m) {
    if (!m) return EGR_OK;
    hipDeviceSynchronize();
    for (void* p : m->owned) hipFree(p);
    for (size_t i = 0; i < m->ctxs.size(); ++i) {
        FsrCtx* c = m->ctxs[i].get();
        for (auto& kv : c->lp_plans) egr_fatllama_plan_destroy(kv.second);
        if (c->gn_ws) hipFree(c->gn_ws);
        if (c->rs_pool) hipFree(c->rs_pool);
        for (unsigned* q : c->rs_retired) hipFree(q);
        if (c->rs_ones) hipFree(c->rs_ones);
        if (c->done) hipEventDestroy(c->done);
        if (i > 0 && c->st) hipStreamDestroy(c->st);
    }
    if (m->ev_fork) hipEventDestroy(m->ev_fork);
    for (auto& r : m->prof) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
    if (m->d_wmax) hipFree(m->d_wmax);
    delete m;
    return EGR_OK;
}

extern "C" int egr_flashsr_create(egr_flashsr** out, const egr_flashsr_config* cfg, const egr_tensor_desc* tensors, int n_tensors,
                                  unsigned flags, void* stream) {
    EGR_CHECK(out && cfg && tensors && n_tensors > 0, EGR_ERR_ARG, "egr_flashsr_create: null argument");
    *out = nullptr;
    EGR_CHECK(cfg->struct_bytes == (int)sizeof(egr_flashsr_config), EGR_ERR_ARG, "egr_flashsr_config size %d != %d (ABI mismatch)",
              cfg->struct_bytes, (int)sizeof(egr_flashsr_config));
    EGR_CHECK(cfg->vae_levels >= 1 && cfg->vae_levels <= EGR_FSR_MAX && cfg->unet_levels >= 1 && cfg->unet_levels <= EGR_FSR_MAX &&
              cfg->voc_n_rates >= 1 && cfg->voc_n_rates <= EGR_FSR_MAX && cfg->voc_n_kernels >= 1 && cfg->voc_n_kernels <= EGR_FSR_MAX &&
              cfg->voc_n_dils >= 1 && cfg->voc_n_dils <= EGR_FSR_MAX && cfg->unet_n_attn >= 0 && cfg->unet_n_attn <= EGR_FSR_MAX,
              EGR_ERR_ARG, "egr_flashsr_config: level / rate counts out of range");
    egr_flashsr* m = new egr_flashsr();
    m->cfg = *cfg;
    m->flags = flags;
    m->ctxs.emplace_back(new FsrCtx());
    m->ctxs[0]->st = (hipStream_t)stream;
    m->use(m->ctxs[0].get());
    hipGetDevice(&m->device);
    if (const char* e = getenv("EGREGORA_FLASHSR_STREAMS")) { const int g = atoi(e); if (g >= 1 && g <= 4) m->max_groups = g; }
    if (const char* e = getenv("EGREGORA_FLASHSR_WINOGRAD_MIN_CH")) m->wino_min_ch = atoi(e);
    m->h2 = !(flags & (EGR_FSR_F32_MFMA | EGR_FSR_SPLIT_BF16X3));
    if (const char* e = getenv("EGREGORA_FLASHSR_SPLIT")) { if (!strcmp(e, "bf16x3")) m->h2 = false; }
    if (const char* e = getenv("EGREGORA_FLASHSR_ROWS")) { const int r = atoi(e); if (r >= 1) m->rows_per_pass = r; }
    if (const char* e = getenv("EGREGORA_FLASHSR_OUT_AMAX")) m->out_amax_on = atoi(e) != 0;
    if (const char* e = getenv("EGREGORA_FLASHSR_DIRECT3X3_MAX_COUT")) m->direct3x3_max_cout = atoi(e);
    if (const char* e = getenv("EGREGORA_FLASHSR_FUSED_AMP")) m->fused_amp = atoi(e) != 0;
    if (const char* e = getenv("EGREGORA_FLASHSR_ARENA_GB")) { const double gb = atof(e); if (gb > 0.0) m->arena_cap = gb * 1e9; }
    build_blocks(m);
    const int down = 1 << (cfg->vae_levels - 1);
    m->lat_h = cfg->n_frames / down; m->lat_w = cfg->n_mels / down;
    const int nb = cfg->n_fft / 2 + 1;
    m->ldm = ((nb + 15) / 16) * 16;
    auto fail = [&](int rc) { egr_flashsr_destroy(m); return rc; };
    int rc = pack_all(m, tensors, n_tensors);
    if (rc) return fail(rc);
    // derived constants travel with the weights (the host computes them exactly as the oracle does): Hann window, mel filterbank,
    // anti-alias FIR, sinusoidal embedding of t = T-1
    const egr_tensor_desc* tw = find(tensors, n_tensors, "const.window");
    const egr_tensor_desc* tf = find(tensors, n_tensors, "const.aa_filter");
    const egr_tensor_desc* tm = find(tensors, n_tensors, "const.mel_fb");
    const egr_tensor_desc* te = find(tensors, n_tensors, "const.time_emb");
    if (!(tw && tf && tm && te)) { set_error("egr_flashsr_create: const.window / const.aa_filter / const.mel_fb / const.time_emb must accompany the weights"); return fail(EGR_ERR_ARG); }
    if (tw->shape[0] != cfg->n_fft || tf->shape[0] != cfg->aa_taps || tm->ndim != 2 || tm->shape[0] != cfg->n_mels || tm->shape[1] != nb ||
        te->shape[te->ndim - 1] != cfg->unet_ch) { set_error("egr_flashsr_create: constant tensor shapes do not match the config"); return fail(EGR_ERR_ARG); }
    void* p = nullptr;
    if ((rc = dev_alloc(m, (size_t)cfg->n_fft * 4, &p))) return fail(rc);
    m->window = (float*)p;
    hipMemcpyAsync(m->window, tw->data, (size_t)cfg->n_fft * 4, hipMemcpyDeviceToDevice, m->st);
    if ((rc = dev_alloc(m, (size_t)cfg->aa_taps * 4, &p))) return fail(rc);
    m->filt = (float*)p;
    hipMemcpyAsync(m->filt, tf->data, (size_t)cfg->aa_taps * 4, hipMemcpyDeviceToDevice, m->st);
    // mel filterbank [n_mels][nb] as a contraction weight: K = nb (padded to ldm), N = n_mels
    if ((rc = add_packed(m, "mel_fb", tm->data, 0, nb, cfg->n_mels, nb, cfg->n_mels, 1, 1, 1, 1, m->ldm))) return fail(rc);
    {   // cosine schedule at t = T-1 (Nichol & Dhariwal), alpha^2 + sigma^2 = 1
        const double s = 0.008, T = (double)cfg->t_steps, t = T - 1.0;
        auto f = [&](double u) { const double cc = cos((u / T + s) / (1 + s) * M_PI / 2); return cc * cc; };
        double abar = f(t + 1) / f(0);
        abar = abar < 1e-5 ? 1e-5 : (abar > 0.99999 ? 0.99999 : abar);
        m->alpha = (float)sqrt(abar); m->sigma = (float)sqrt(1.0 - abar);
    }
    if ((rc = fold_time_embedding(m, te->data))) return fail(rc);
    if (hipStreamSynchronize(m->st) != hipSuccess || hipGetLastError() != hipSuccess) { set_error("egr_flashsr_create: device work failed"); return fail(EGR_ERR_HIP); }
    *out = m;
    return EGR_OK;
}

extern "C" int egr_flashsr_set_rows_per_pass(egr_flashsr* m, int rows) {
    EGR_CHECK(m && rows >= 1, EGR_ERR_ARG, "bad argument");
    m->rows_per_pass = rows;
    return EGR_OK;
}

extern "C" int egr_flashsr_forward(egr_flashsr* m, const float* x, const float* noise, int rows, int lowpass, float* y, float* const* stages,
                                   void* stream) {
    EGR_CHECK(m && x && noise && y && rows >= 1, EGR_ERR_ARG, "egr_flashsr_forward: null / empty argument");
    ForwardGuard guard(m->device, (hipStream_t)stream);
    m->ctxs[0]->st = (hipStream_t)stream;
    m->use(m->ctxs[0].get());
    // the introspection walk stays on the three-term kernels (bit-equal to the operator API) unless egr_flashsr_set_split(h, 2) asked
    // for the fp16 operand terms here too (stage taps for the tests)
    m->h2_mode = (m->h2 && m->h2_fwd && m->h2_nweights > 0) ? 1 : -1;
    const int rc = forward(m, x, noise, rows, lowpass, y, stages);
    m->h2_mode = -1;
    return rc;
}

// x [rows][chunk] -> y [rows][chunk]; rows = chunks x channels ride the batch dimension (reference :366-368) and are processed
// rows_per_pass at a time; the noise of row r is a function of (seed, row_ids[r] or r) only, so results do not depend on how the
// rows are spread over passes or ranks.
extern "C" int egr_streams_overlap_us(void* stream_a, void* stream_b, int spin_us, double* elapsed_us);

// Side streams for concurrent row groups.  HIP multiplexes its streams onto a few hardware queues and two streams on one queue
// run in order, so candidates are CHECKED: two 300 us spin kernels take ~300 us together on different queues, ~600 us on one.
// Verified against the caller's stream and against each other; re-checked when the caller's stream changes.
static int ensure_side_streams(egr_flashsr* m, hipStream_t caller, int want) {
    bool known = false;
    for (hipStream_t s : m->verified_for) known = known || s == caller;
    auto overlaps = [&](hipStream_t a, hipStream_t b) {
        double best = 1e9, us = 0;
        for (int i = 0; i < 2; ++i) { if (egr_streams_overlap_us(a, b, 300, &us) != EGR_OK) return false; best = std::min(best, us); }
        return best < 450.0;
    };
    if (!known) {                                    // drop side contexts that do not overlap with THIS caller stream
        for (size_t i = 1; i < m->ctxs.size();) {
            if (overlaps(caller, m->ctxs[i]->st)) { ++i; continue; }
            hipStreamSynchronize(m->ctxs[i]->st);
            for (auto& kv : m->ctxs[i]->lp_plans) egr_fatllama_plan_destroy(kv.second);
            if (m->ctxs[i]->gn_ws) hipFree(m->ctxs[i]->gn_ws);
            if (m->ctxs[i]->rs_pool) hipFree(m->ctxs[i]->rs_pool);
            for (unsigned* q : m->ctxs[i]->rs_retired) hipFree(q);
            if (m->ctxs[i]->rs_ones) hipFree(m->ctxs[i]->rs_ones);
            if (m->ctxs[i]->done) hipEventDestroy(m->ctxs[i]->done);
            hipStreamDestroy(m->ctxs[i]->st);
            m->ctxs.erase(m->ctxs.begin() + i);
        }
        m->verified_for.assign(1, caller);
    }
    for (int tries = 0; (int)m->ctxs.size() - 1 < want && tries < 12; ++tries) {
        hipStream_t s = nullptr;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) break;
        bool ok = overlaps(caller, s);
        for (size_t i = 1; ok && i < m->ctxs.size(); ++i) ok = overlaps(m->ctxs[i]->st, s);
        if (!ok) { hipStreamDestroy(s); continue; }
        m->ctxs.emplace_back(new FsrCtx());
        m->ctxs.back()->st = s;
        hipEventCreateWithFlags(&m->ctxs.back()->done, hipEventDisableTiming);
    }
    return (int)m->ctxs.size() - 1;
}

// x [rows][chunk] -> y [rows][chunk]; rows = chunks x channels ride the batch dimension (reference :366-368) and are processed
// rows_per_pass at a time; the noise of row r is a function of (seed, row_ids[r] or r) only, so results do not depend on how the
// rows are spread over passes, groups or ranks (beyond fp32 round-off: tile choices follow the row count of a forward).
// A pass of >= 2 * min_group_rows rows is split into up to max_groups contiguous ROW GROUPS that run as concurrent forwards on the
// handle's verified side streams, each with its own scratch arena (fork / join by events around the pass): one group's
// matrix-bound kernels overlap another's HBM-bound ones and fill each other's tails (26 rows: 260 ms in one forward).
// h2_mode 0: the handle's scheme, decided under the device's forward lock; count_call: the call shows in egr_flashsr_split_info and in
// the profile records (false for the warm-up pass)
static int infer_once(egr_flashsr* m, const float* x, int rows, int lowpass, uint64_t seed, const int64_t* row_ids, int64_t id_base, float* y,
                      void* stream, int h2_mode, bool count_call);

extern "C" int egr_flashsr_infer(egr_flashsr* m, const float* x, int rows, int lowpass, uint64_t seed, const int64_t* row_ids, float* y,
                                 void* stream) {
    EGR_CHECK(m && x && y && rows >= 1, EGR_ERR_ARG, "egr_flashsr_infer: null / empty argument");
    // operand scheme of this call's forwards: two fp16 terms with per-row device-side scales (see the h2 fields of the handle), or
    // three bf16 terms.  Either way the call only enqueues work: no read-back, no host synchronisation.
    // (the scheme is handed to infer_once, which publishes it in the handle only while it holds the device's forward lock)
    return infer_once(m, x, rows, lowpass, seed, row_ids, 0, y, stream, 0, true);
}

// enabled: egr_flashsr_infer runs the fp16 operand terms on this handle; weights: contraction weights that hold fp16 terms;
// calls: egr_flashsr_infer calls made on the scheme.  (ABI 3 reported a calibration state and re-run count here: the scheme has
// neither any more -- scales are per row and per call, derived on the device.)
extern "C" int egr_flashsr_split_info(egr_flashsr* m, int* enabled, int* weights, int64_t* calls) {
    EGR_CHECK(m != nullptr, EGR_ERR_ARG, "handle is null");
    if (enabled) *enabled = m->h2 && m->h2_nweights > 0;
    if (weights) *weights = m->h2_nweights;
    if (calls) *calls = m->h2_calls;
    return EGR_OK;
}

// scheme: 0 = three bf16 terms always, 1 = two fp16 terms with per-row device-side scales in egr_flashsr_infer (needs a handle
// created with the scheme available), 2 = as 1 and egr_flashsr_forward runs the fp16 terms as well (stage taps for tests)
extern "C" int egr_flashsr_set_split(egr_flashsr* m, int scheme) {
    EGR_CHECK(m && scheme >= 0 && scheme <= 2, EGR_ERR_ARG, "bad argument");
    EGR_CHECK(scheme == 0 || m->h2_nweights > 0, EGR_ERR_UNSUPPORTED, "this handle holds no fp16 weight terms");
    m->h2 = scheme >= 1;
    m->h2_fwd = scheme == 2;
    return EGR_OK;
}

// rows of one call; row_ids NULL: implicit ids id_base .. id_base + rows - 1
static int infer_once(egr_flashsr* m, const float* x, int rows, int lowpass, uint64_t seed, const int64_t* row_ids, int64_t id_base, float* y,
                      void* stream, int h2_mode, bool count_call) {
    hipStream_t st0 = (hipStream_t)stream;
    ForwardGuard guard(m->device, st0);                  // everything below touches per-handle state: inside the device's forward lock
    if (h2_mode == 0) h2_mode = (m->h2 && m->h2_nweights > 0 && !m->count_flops) ? 1 : -1;
    struct ModeScope {                                   // the operand scheme of THIS call, visible to the operators while the lock is held
        egr_flashsr* m; bool prof;
        ModeScope(egr_flashsr* mm, int mode, bool count) : m(mm), prof(mm->profiling) {
            m->h2_mode = mode;
            if (mode == 1 && count) ++m->h2_calls;
            if (!count) m->profiling = false;            // a warm-up pass leaves no profile records behind
        }
        ~ModeScope() { m->h2_mode = -1; m->profiling = prof; }
    } mode_scope(m, h2_mode, count_call);
    m->ctxs[0]->st = st0;
    m->use(m->ctxs[0].get());
    const egr_flashsr_config& c = m->cfg;
    const int64_t per_row = (int64_t)m->lat_h * m->lat_w * c.z_ch;
    // scratch budget: fewer rows per pass instead of an allocation failure next to the host's other models
    int rpp = m->rows_per_pass;
    if (m->arena_cap > 0.0) {
        // before any pass has run: ~24 live tensors of the widest activation (n_frames x n_mels x vae_ch) per row, measured 1.5 GB at the full size
        const double est = 24.0 * 4.0 * (double)c.n_frames * c.n_mels * c.vae_ch;
        const double per_row_bytes = m->arena_row_bytes > 0.0 ? m->arena_row_bytes : est;
        rpp = std::max(1, std::min(rpp, (int)(m->arena_cap / per_row_bytes)));
    }
    static const bool trace = getenv("EGR_FSR_TRACE") && atoi(getenv("EGR_FSR_TRACE")) != 0;
    const double t_enter = now_ms(), malloc0 = 1e-3 * (double)g_arena_malloc_us.load();
    double t_side = 0.0;
    int groups_max = m->profiling ? 1 : m->max_groups;          // per-kernel timing wants the kernels alone on the chip
    if (groups_max > 1 && std::min(rows, rpp) >= 2 * m->min_group_rows) {
        const double t0 = now_ms();
        groups_max = std::min(m->max_groups, 1 + ensure_side_streams(m, st0, groups_max - 1));   // side contexts of an earlier, wider setting stay idle
        t_side = now_ms() - t0;
    } else {
        groups_max = 1;
    }
    if (groups_max > 1 && !m->ev_fork) EGR_HIP(hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming));
    int rc = EGR_OK;
    // passes of equal size (260 rows at 32 per pass: nine passes of 29 / 28 rows instead of eight of 32 and one of 4)
    const int npass = (rows + rpp - 1) / rpp;
    const int per = (rows + npass - 1) / npass;
    for (int lo = 0; lo < rows && rc == EGR_OK; lo += per) {
        const int n = std::min(per, rows - lo);
        const int G = std::max(1, std::min(groups_max, n / m->min_group_rows));
        if (G > 1) EGR_HIP(hipEventRecord(m->ev_fork, st0));
        int done_rows = 0;
        for (int g = 0; g < G && rc == EGR_OK; ++g) {
            const int ng = (n - done_rows + (G - g) - 1) / (G - g);
            const int glo = lo + done_rows;
            FsrCtx* cx = m->ctxs[g].get();
            if (g > 0) EGR_HIP(hipStreamWaitEvent(cx->st, m->ev_fork, 0));
            m->use(cx);
            Ten nz;
            rc = new_ten(m, nz, {ng, per_row});
            if (rc == EGR_OK) rc = egr_randn_base(nz.p, per_row, ng, seed, row_ids ? row_ids + glo : nullptr, id_base + glo, m->st);
            if (rc == EGR_OK) rc = forward(m, x + (size_t)glo * c.chunk, nz.p, ng, lowpass, y + (size_t)glo * c.chunk, nullptr);
            if (g > 0) {                                          // join even after an error: nothing may outlive the call unordered
                hipEventRecord(cx->done, cx->st);
                hipStreamWaitEvent(st0, cx->done, 0);
            }
            done_rows += ng;
        }
        m->use(m->ctxs[0].get());
    }
    if (rc == EGR_OK && per > 0) {                       // what a row of a pass of this size costs in scratch (the arenas never shrink)
        double tot = 0.0;
        for (auto& cx : m->ctxs) tot += (double)cx->arena.total;
        m->arena_row_bytes = std::max(m->arena_row_bytes, tot / std::min(per, rows));
    }
    if (trace) fprintf(stderr, "[egr_flashsr_infer] rows %d: host %.1f ms (side-stream check %.1f, arena hipMalloc %.1f), arenas %.2f GB\n", rows,
                       now_ms() - t_enter, t_side, 1e-3 * (double)g_arena_malloc_us.load() - malloc0, (double)egr_flashsr_scratch_bytes(m) / 1e9);
    return rc;
}

// Scratch budget in bytes for the handle's arenas (0 = none; EGREGORA_FLASHSR_ARENA_GB sets it at creation): egr_flashsr_infer
// lowers its rows per pass until a pass is expected to fit -- slower passes instead of an out-of-memory next to the host's other
// models.  Arenas already allocated are not returned.
extern "C" int egr_flashsr_set_arena_cap(egr_flashsr* m, double bytes) {
    EGR_CHECK(m && bytes >= 0.0, EGR_ERR_ARG, "bad argument");
    m->arena_cap = bytes;
    return EGR_OK;
}

extern "C" int egr_flashsr_warmup(egr_flashsr* m, int rows, void* stream) {
    EGR_CHECK(m && rows >= 1, EGR_ERR_ARG, "bad argument");
    const size_t n = (size_t)rows * m->cfg.chunk;
    float* buf = nullptr;
    if (hipMalloc((void**)&buf, 2 * n * sizeof(float)) != hipSuccess) { set_error("egr_flashsr_warmup: hipMalloc(%zu) failed", 2 * n * sizeof(float)); return EGR_ERR_ALLOC; }
    int rc = EGR_OK;
    if (hipMemsetAsync(buf, 0, n * sizeof(float), (hipStream_t)stream) != hipSuccess) rc = EGR_ERR_HIP;
    if (rc == EGR_OK) rc = infer_once(m, buf, rows, 0, 0, nullptr, 0, buf + n, stream, 0, false);      // (split_info counts the host's calls, not this one)
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess && rc == EGR_OK) { set_error("egr_flashsr_warmup: device work failed"); rc = EGR_ERR_HIP; }
    hipFree(buf);
    return rc;
}

extern "C" int egr_flashsr_set_streams(egr_flashsr* m, int max_groups, int min_group_rows) {
    EGR_CHECK(m && max_groups >= 1 && max_groups <= 4 && min_group_rows >= 1, EGR_ERR_ARG, "bad argument");
    m->max_groups = max_groups;
    m->min_group_rows = min_group_rows;
    return EGR_OK;
}

extern "C" int egr_flashsr_set_profiling(egr_flashsr* m, int enable) {
    EGR_CHECK(m != nullptr, EGR_ERR_ARG, "handle is null");
    for (auto& r : m->prof) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
    m->prof.clear();
    m->profiling = enable != 0;
    return EGR_OK;
}

// Aggregated HIP-event timing of the MFMA contraction launches since profiling was switched on: entry i of the distinct kernel
// instantiations -> name, launches, flops, milliseconds.  Returns the number of distinct kinds through *count when kind_buf is null.
extern "C" int egr_flashsr_profile(egr_flashsr* m, int index, char* kind_buf, size_t buflen, int64_t* launches, double* flops, double* ms,
                                   int* count) {
    EGR_CHECK(m != nullptr, EGR_ERR_ARG, "handle is null");
    EGR_HIP(hipDeviceSynchronize());
    std::map<std::string, std::tuple<int64_t, double, double>> agg;
    static const bool dump = getenv("EGR_FSR_PROFILE_DUMP") && atoi(getenv("EGR_FSR_PROFILE_DUMP")) != 0;
    for (auto& r : m->prof) {
        float t = 0.f;
        hipEventElapsedTime(&t, r.a, r.b);
        if (dump && count && !kind_buf)              // dev: every timed launch with its layer and shape (once per read-out)
            fprintf(stderr, "[egr_flashsr profile] %-36s %9.3f ms %8.1f GFLOP %7.1f TF/s-eq  %s\n", r.kind.c_str(), t, r.flops / 1e9, t > 0 ? r.flops / t / 1e9 : 0.0,
                    r.detail.c_str());
        auto& e = agg[r.kind];
        std::get<0>(e) += 1; std::get<1>(e) += r.flops; std::get<2>(e) += t;
    }
    if (count) *count = (int)agg.size();
    if (!kind_buf) return EGR_OK;
    EGR_CHECK(index >= 0 && index < (int)agg.size(), EGR_ERR_ARG, "profile index out of range");
    auto it = agg.begin();
    std::advance(it, index);
    snprintf(kind_buf, buflen, "%s", it->first.c_str());
    if (launches) *launches = std::get<0>(it->second);
    if (flops) *flops = std::get<1>(it->second);
    if (ms) *ms = std::get<2>(it->second);
    return EGR_OK;
}

// Dense-contraction flops of one forward over `rows` rows (dry run with counting on; 2 Cin Cout Kh Kw Hout Wout per conv as executed).
extern "C" int egr_flashsr_flop_count(egr_flashsr* m, int rows, double* flops, void* stream) {
    EGR_CHECK(m && flops && rows >= 1, EGR_ERR_ARG, "bad argument");
    m->ctxs[0]->st = (hipStream_t)stream;
    m->use(m->ctxs[0].get());
    const egr_flashsr_config& c = m->cfg;
    Ten x, nz, y;
    OKR(new_ten(m, x, {rows, c.chunk}));
    OKR(new_ten(m, nz, {rows, (int64_t)m->lat_h * m->lat_w * c.z_ch}));
    OKR(new_ten(m, y, {rows, c.chunk}));
    EGR_HIP(hipMemsetAsync(x.p, 0, x.bytes, m->st));
    OKR(egr_randn(nz.p, (int64_t)m->lat_h * m->lat_w * c.z_ch, rows, 0, nullptr, m->st));
    m->count_flops = true; m->flops = 0.0;
    m->h2_mode = -1;
    const int rc = forward(m, x.p, nz.p, rows, 0, y.p, nullptr);
    m->count_flops = false;
    EGR_HIP(hipStreamSynchronize(m->st));
    *flops = m->flops;
    return rc;
}

extern "C" int64_t egr_flashsr_scratch_bytes(egr_flashsr* m) {
    int64_t t = 0;
    if (m) for (auto& c : m->ctxs) t += (int64_t)c->arena.total;
    return t;
}

// ---------------------------------------------------------------------------------------------------- weight blob files
// "EGRW" container: what `egr_flashsr_create` takes, on disk -- written once by the host that owns checkpoint I/O
// (flashsr_weights.write_blob) and loadable from plain C:
//   char magic[8] = "EGRW0001"; int32 config_bytes; egr_flashsr_config; int32 n_tensors;
//   n x { int32 name_len; char name[name_len]; int32 ndim; int64 shape[4]; int64 offset (bytes from file start); }
//   ... fp32 data, each tensor 64-byte aligned
extern "C" int egr_flashsr_create_from_file(egr_flashsr** out, const char* path, unsigned flags, void* stream) {
    EGR_CHECK(out && path, EGR_ERR_ARG, "null argument");
    *out = nullptr;
    // nothing may throw across the C boundary: allocation failures of the host buffers become EGR_ERR_ALLOC
    try {
        FILE* f = fopen(path, "rb");
        EGR_CHECK(f != nullptr, EGR_ERR_ARG, "cannot open %s", path);
        std::vector<char> buf;
        long sz = -1;
        if (fseek(f, 0, SEEK_END) == 0) sz = ftell(f);
        if (sz <= 16 || fseek(f, 0, SEEK_SET) != 0) {
            fclose(f);
            set_error("%s is not an EGRW0001 weight blob (size %ld)", path, sz);
            return EGR_ERR_ARG;
        }
        buf.resize((size_t)sz);
        const size_t got = fread(buf.data(), 1, (size_t)sz, f);
        fclose(f);
        EGR_CHECK(got == (size_t)sz && memcmp(buf.data(), "EGRW0001", 8) == 0, EGR_ERR_ARG, "%s is not an EGRW0001 weight blob", path);
        size_t pos = 8;
        auto rd = [&](void* dst, size_t n) -> bool { if (n > buf.size() - pos) return false; memcpy(dst, buf.data() + pos, n); pos += n; return true; };
        int32_t cb = 0, nt = 0;
        egr_flashsr_config cfg;
        constexpr int32_t kMaxTensors = 1 << 20;
        bool ok = rd(&cb, 4) && cb == (int32_t)sizeof(cfg) && rd(&cfg, sizeof(cfg)) && rd(&nt, 4) && nt > 0 && nt <= kMaxTensors;
        EGR_CHECK(ok, EGR_ERR_ARG, "%s: header does not match this library's egr_flashsr_config (%d bytes) or tensor count out of range", path, (int)sizeof(cfg));
        std::vector<std::string> names(nt);
        std::vector<egr_tensor_desc> ds(nt);
        std::vector<int64_t> offs(nt);
        std::vector<size_t> bytes(nt);
        size_t total = 0;
        for (int i = 0; i < nt && ok; ++i) {
            int32_t nl = 0, nd = 0;
            ok = rd(&nl, 4) && nl > 0 && nl < 512;
            if (ok) { names[i].resize(nl); ok = rd(&names[i][0], nl); }
            ok = ok && rd(&nd, 4) && nd >= 0 && nd <= 4 && rd(ds[i].shape, 32) && rd(&offs[i], 8);
            if (!ok) break;
            ds[i].ndim = nd;
            // element count without overflow: every extent non-negative, the running product bounded by what the file can hold
            const size_t cap = buf.size() / 4;
            size_t ne = 1;
            for (int d = 0; d < nd && ok; ++d) {
                const int64_t e = ds[i].shape[d];
                ok = e >= 0 && (e == 0 || ne <= cap / (size_t)e);
                if (ok) ne *= (size_t)e;
            }
            ok = ok && offs[i] >= 0 && (size_t)offs[i] <= buf.size() && ne <= (buf.size() - (size_t)offs[i]) / 4;
            bytes[i] = ne * 4;
            total += (bytes[i] + 255) & ~(size_t)255;
        }
        EGR_CHECK(ok, EGR_ERR_ARG, "%s: corrupt tensor index", path);
        char* dev = nullptr;
        if (hipMalloc((void**)&dev, total + 256) != hipSuccess) {
            set_error("hipMalloc of %zu bytes for the weight blob failed", total + 256);
            return EGR_ERR_ALLOC;
        }
        size_t dpos = 0;
        for (int i = 0; i < nt; ++i) {
            if (bytes[i]) {
                const hipError_t ce = hipMemcpy(dev + dpos, buf.data() + offs[i], bytes[i], hipMemcpyHostToDevice);
                if (ce != hipSuccess) {
                    hipFree(dev);
                    set_error("hipMemcpy of tensor %s -> %s", names[i].c_str(), hipGetErrorString(ce));
                    return EGR_ERR_HIP;
                }
            }
            ds[i].name = names[i].c_str();
            ds[i].data = (const float*)(dev + dpos);
            dpos += (bytes[i] + 255) & ~(size_t)255;
        }
        const int rc = egr_flashsr_create(out, &cfg, ds.data(), nt, flags, stream);
        hipFree(dev);
        return rc;
    } catch (const std::bad_alloc&) {
        set_error("%s: out of host memory while reading the weight blob", path);
        return EGR_ERR_ALLOC;
    } catch (...) {
        set_error("%s: unexpected failure while reading the weight blob", path);
        return EGR_ERR_ARG;
    }
}
