// Dev harness: times k_conv_s3 standalone (no torch).  Build (from the repo root):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I comfyui-egregora-audio-super-resolution_amd/csrc \
//         tools/ubench/conv_s3_harness.hip -o tools/ubench/conv_s3_<tag>
// The round-3 ablation study (global loads / operand split / LDS stores / barrier / epilogue switched off one at a time through
// -DS3_ABL_* branches INSIDE the kernel; results in profiles/r03/conv_s3_ablation.log) lived in the product translation unit; the
// branches were removed from it in round 4 (git show 4c5e402:comfyui-egregora-audio-super-resolution_amd/csrc/egr_nn_gemm_s3.hip
// has them).  run_conv_s3_ablation.sh / run_conv_s3_quick.sh therefore time the shipped kernel only.
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "egr_nn_gemm_s3.hip"
static float* g_zero_page = nullptr;
namespace egr { int zero_page(const float** out) { *out = g_zero_page; return 0; }
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); } }
int main(int argc, char** argv) {
    int B = 26, H = 128, W = 64, Ci = 512, Co = 512, k = 3, reps = getenv("S3_REPS") ? atoi(getenv("S3_REPS")) : 5;
    if (argc >= 7) { B = atoi(argv[1]); H = atoi(argv[2]); W = atoi(argv[3]); Ci = atoi(argv[4]); Co = atoi(argv[5]); k = atoi(argv[6]); }
    const long long M = (long long)B * H * W, K = (long long)k * k * Ci, ns = K / 16;
    float *x, *wp, *y, *zeros; void* w3;
    hipMalloc(&x, M * Ci * 4); hipMalloc(&wp, ns * Co * 16 * 4); hipMalloc(&y, M * Co * 4); hipMalloc(&w3, ns * 3 * Co * 32); hipMalloc(&zeros, 4096);
    hipMemset(zeros, 0, 4096); g_zero_page = zeros;
    std::vector<float> h(M * Ci); const bool zero = getenv("S3_ZERO") != nullptr;
    for (auto& v : h) v = zero ? 0.f : (float)rand() / RAND_MAX - 0.5f; hipMemcpy(x, h.data(), M * Ci * 4, hipMemcpyHostToDevice);
    std::vector<float> hw(ns * Co * 16); for (auto& v : hw) v = zero ? 0.f : ((float)rand() / RAND_MAX - 0.5f) * 0.05f; hipMemcpy(wp, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    const int sch = getenv("S3_SCH") ? atoi(getenv("S3_SCH")) : 0;
    if (sch) egr_split2h_pack(wp, w3, ns, Co, 8192.f, nullptr); else
    egr_split3_pack(wp, w3, ns, Co, nullptr);
    egr::ConvP p; memset(&p, 0, sizeof(p));
    p.x = x; p.w3 = (const uint4*)w3; p.y = y; p.B = B; p.H = H; p.W = W; p.Cin = Ci; p.OH = H; p.OW = W; p.Cout = Co; p.KH = k; p.KW = k;
    p.stride = 1; p.dil = 1; p.pad_t = k / 2; p.pad_l = k / 2; p.M = (int)M; p.K = (int)K; p.osy = p.osx = 1; p.OHF = H; p.OWF = W;
    p.ksplit = 1; p.kt_per = (int)ns; p.zeros = zeros;
    // scheme 1: per-batch-row operand maxima on the device (here: 4.0 for every row, one float per 128-byte line)
    unsigned* ra = nullptr;
    hipMalloc((void**)&ra, (size_t)B * EGR_ROW_AMAX_STRIDE * 4);
    {
        std::vector<float> h((size_t)B * EGR_ROW_AMAX_STRIDE, 4.0f);
        hipMemcpy(ra, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    }
    p.sch = sch; p.out_scale = sch ? 1.f / 8192.f : 1.f; p.row_amax = sch ? ra : nullptr; p.rows_div = sch ? H * W : 1;
    p.out_amax = nullptr; p.gn_part = nullptr;
    if (getenv("S3_XCD")) p.xcd_remap = 1;
    const int bn = getenv("S3_BN") ? atoi(getenv("S3_BN")) : (Co > 64 ? 128 : (Co > 32 ? 64 : 32));
    const int bm = getenv("S3_BM") ? atoi(getenv("S3_BM")) : egr::s3_bm(M, Co, bn);
    dim3 grid((unsigned)((M + bm - 1) / bm), (unsigned)((Co + bn - 1) / bn));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    egr::launch_conv_s3(bm, bn, grid, 0, p); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) egr::launch_conv_s3(bm, bn, grid, 0, p);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    if (getenv("S3_CHECK")) {          // determinism: a second launch into another buffer must match bit for bit
        float* y2; hipMalloc(&y2, M * Co * 4); hipMemset(y2, 0xff, M * Co * 4);
        egr::ConvP q = p; q.y = y2; egr::launch_conv_s3(bm, bn, grid, 0, q); hipDeviceSynchronize();
        std::vector<float> ha(M * Co), hb(M * Co);
        hipMemcpy(ha.data(), y, M * Co * 4, hipMemcpyDeviceToHost); hipMemcpy(hb.data(), y2, M * Co * 4, hipMemcpyDeviceToHost);
        long long bad = 0; for (long long i = 0; i < M * Co; ++i) bad += memcmp(&ha[i], &hb[i], 4) != 0;
        double ref = 0; for (int kk = 0; kk < K; ++kk) ref += (double)h[(M / 2) * Ci + kk] * hw[((kk / 16) * Co + 0) * 16 + (kk % 16)];   // k = 1 only
        printf("mismatching elements between two launches: %lld of %lld; y[M/2][0]=%.7f ref(k=1)=%.7f x0=%.6f\n", bad, M * Co, ha[(M / 2) * Co], ref, h[0]);
    }
    std::vector<float> hy(4096); hipMemcpy(hy.data(), y + (M / 2) * Co, 4096 * 4, hipMemcpyDeviceToHost);
    double cs = 0; for (float v : hy) cs += (double)v * v;
    printf("cs=%.9e ", cs);
    printf("bm%d B%d %dx%d Ci%d Co%d k%d: %.3f ms  %.1f TF/s (fp32-equivalent)  err=%s\n", bm, B, H, W, Ci, Co, k, ms, 2.0 * M * Co * K / ms / 1e9,
           hipGetErrorString(hipGetLastError()));
    return 0;
}
