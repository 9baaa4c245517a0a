/* libegregora_amd.so -- C ABI of the MI355X-native (gfx950) backend for the two compute hot paths of
 * ComfyUI-Egregora-Audio-Super-Resolution.
 *
 * Conventions
 *   - every function returns an int status (EGR_OK == 0); egr_last_error() returns a thread-local
 *     message for the last failure.  No C++ exceptions cross this boundary.
 *   - all data pointers are DEVICE pointers owned by the caller (the Python host passes
 *     torch.Tensor.data_ptr()); `stream` is a hipStream_t passed as void* (0 = default stream); work is
 *     enqueued on that stream and the call returns without synchronising unless stated.
 *   - plans own their twiddle tables / workspaces on the device that was current at creation time.
 *     A plan is not thread-safe; use one per device per thread.
 *
 * Each entry point cites the reference interface it stands in for (paths relative to the reference
 * repository root).
 */
#ifndef EGREGORA_AMD_H
#define EGREGORA_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EGR_ABI_VERSION 1

#define EGR_OK 0
#define EGR_ERR_ARG 1          /* bad argument */
#define EGR_ERR_HIP 2          /* a HIP runtime call failed */
#define EGR_ERR_UNSUPPORTED 3  /* length / configuration not supported by this build */
#define EGR_ERR_ALLOC 4

const char* egr_last_error(void);
int egr_abi_version(void);
/* number of visible HIP devices, or a negative status; never throws (used by the host's loud
 * "no MI355X" check that replaces egregora_fat_llama_gpu.py:132-159 _ensure_gpu_stack). */
int egr_device_count(void);
/* name/arch string of device `dev` (e.g. "gfx950:sramecc+:xnack-") into buf. */
int egr_device_arch(int dev, char* buf, size_t buflen);

/* ------------------------------------------------------------------------------------------------
 * Fat-Llama iterative spectral enhancer.
 * Replaces: fat_llama.audio_fattener.feed.upscale(...) as called at
 *   egregora_fat_llama_gpu.py:213-224 (GPU node) and egregora_fat_llama_cpu.py:126-134 (CPU node),
 * plus the arithmetic of the two disk hops around it (temp WAV hand-over egregora_fat_llama_gpu.py:34-37,
 * write patch :191-205, read-back :291-294) -- no file is ever written.
 * ---------------------------------------------------------------------------------------------- */
typedef struct egr_fatllama_plan egr_fatllama_plan;

/* flags for egr_fatllama_enhance */
#define EGR_FL_NORMALIZE 0x1u  /* upstream toggle_normalize: out / max|out| (joint over channels)  */
#define EGR_FL_AUTOSCALE 0x2u  /* upstream toggle_autoscale: per channel out *= max|in| / max|out| */
#define EGR_FL_PCM_IN    0x4u  /* x is unit-scale float; quantise like the temp-WAV write (PCM_16)  */
#define EGR_FL_NODE_POST 0x8u  /* apply write patch (/32768 iff max>1) + PCM_16 write + float read  */

/* Host-only planning query (no GPU needed): fills info[] =
 *   {supported, N, M, M1, M2, TC, nst1, nst2, radix1[0..13], radix2[0..13], lds_col, lds_row}.
 * m1_hint <= 0 lets the planner choose. */
#define EGR_FL_INFO_LEN 40
int egr_fatllama_plan_query(int64_t n_in, int factor, int m1_hint, int64_t info[EGR_FL_INFO_LEN]);

int egr_fatllama_plan_create(egr_fatllama_plan** out, int64_t n_in, int channels, int factor, int m1_hint,
                             int tc_hint);
int egr_fatllama_plan_destroy(egr_fatllama_plan* plan);

/* x: [channels][n_in] float32, out: [channels][n_in*factor] float32 (both device).
 * Runs: (PCM_IN quantise) -> linear up-rate by `factor` -> d0 = |y|>thr ? y : 0 ->
 *       max_iter x { X = rfft(d); X = |X|>thr ? X : 0; d = irfft(X) } -> out = y + d ->
 *       (autoscale) -> (normalise) -> (NODE_POST).
 * All max_iter iterations are executed (no fixed-point early exit). */
int egr_fatllama_enhance(egr_fatllama_plan* plan, const float* x, float* out, int max_iter, float threshold,
                         unsigned flags, void* stream);

/* Debug / roofline helpers: the forward transposed half-spectrum after `iters` iterations is not
 * exposed; these run single stages of the loop so tests can bisect. which: 0 = peaks {pin[C], pout[C]}
 * of the last enhance call copied to host (synchronises the stream). */
int egr_fatllama_last_peaks(egr_fatllama_plan* plan, float* host_pin, float* host_pout, void* stream);
/* Average HIP-event duration (ms) and launch count of the two loop kernels over the last enhance
 * call when profiling was enabled with egr_fatllama_set_profiling(plan, 1). Synchronises. */
int egr_fatllama_set_profiling(egr_fatllama_plan* plan, int enable);
int egr_fatllama_kernel_times(egr_fatllama_plan* plan, double* row_ms_avg, double* col_ms_avg, int64_t* row_launches,
                              int64_t* col_launches);

/* ------------------------------------------------------------------------------------------------
 * Host-glue arithmetic moved on device (FlashSR node path and the parity yardstick).
 * ---------------------------------------------------------------------------------------------- */

/* y[i] = wrap_int16(rint(x[i]*write_scale)) / read_div      (read_div = 1 keeps the integer scale)
 * Replaces the libsndfile float->PCM_16->float hops at egregora_fat_llama_gpu.py:36 and :291. */
int egr_pcm16_roundtrip(const float* x, float* y, int64_t n, float write_scale, float read_div, void* stream);

/* STFT magnitude, Hann-windowed frames, no centring, mono downmix = mean over channels:
 *   frames = 1 + max(0,(n - n_fft)/hop); out: [frames][n_fft/2+1] float32 (frame-major; the reference's
 *   array is the transpose).  window: device float[n_fft] (caller supplies np.hanning(n_fft) so the
 *   window is bit-identical to the reference's).  n_fft must be even and {2,3,5,7,11,13}-smooth, <= 8192.
 * Replaces: _stft_mag at egregora_audio_eval_pack.py:389-402 (twin egregora_null_test_suite.py:167-180). */
int egr_stft_mag(const float* x, int channels, int64_t n, int n_fft, int hop, const float* window, float* out,
                 void* stream);

/* Hann-weighted overlap-add of uniformly spaced chunk predictions and division by the weight sum,
 * evaluated as a gather (each output sample sums its covering chunks in chunk order, float32,
 * unfused mul/add => bit-identical to the reference's scatter loop).
 *   preds: [n_chunks][channels][lp] ; chunk k starts at k*hop with valid length min(win, total-k*hop, lp)
 *   window: device float[win] ; out: [channels][total]
 * Replaces: _wola_stitch at egregora_audio_super_resolution.py:227-251. */
int egr_wola_stitch(const float* preds, int n_chunks, int channels, int64_t lp, int64_t total, int64_t win,
                    int64_t hop, const float* window, float* out, void* stream);

/* Slice + zero-pad: chunks[k][c][0..win) = x[c][k*hop .. k*hop+win) (zero beyond total).
 * Replaces the slice/pad at egregora_audio_super_resolution.py:411-416. */
int egr_chunk_gather(const float* x, int channels, int64_t total, int64_t win, int64_t hop, int chunk_begin,
                     int n_chunks, float* chunks, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EGREGORA_AMD_H */
