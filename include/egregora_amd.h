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

/* 2: paired chirp-z plans (egr_fatllama_plan_create_chirpz, info[40..41], egr_fatllama_kernel_times3), model handle, DFN / null-test entry points
 * 3: two-term fp16 operand scheme (egr_split2h_pack, egr_absmax, egr_conv_h2, egr_flashsr_set_split / _split_info)
 * 4: the fp16 scheme's scales are per batch row and derived on the device: egr_conv_h2 takes (w_scale, row_amax, batch_rows)
 *    instead of (a_scale, w_scale, amax); egr_absmax_rows, egr_winograd4_input_ra / _output_ra, egr_snake_aa_ra, egr_randn_base added; egr_flashsr_split_info
 *    reports (enabled, weights, calls); egr_flashsr_infer no longer measures, verifies, re-runs or synchronises
 * 5: egr_fatllama_joint_peak / egr_fatllama_finalize + EGR_FL_DEFER_FINALIZE (channel-parallel Fat-Llama: one 4-byte all-reduce(MAX));
 *    EGR_FL_THR_RELATIVE carries the spectrum maximum from iteration to iteration (EGR_FL_THR_RECOMPUTE: the per-iteration pass) */
#define EGR_ABI_VERSION 5

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
/* Threshold semantics of the loop -- upstream's are unverified (SPEC.md section 3 names each variant); default = none set */
#define EGR_FL_THR_RELATIVE 0x10u /* level = threshold * max|.| of the array judged (per channel, every iteration)  */
#define EGR_FL_THR_SOFT     0x20u /* spectrum: X max(0, 1 - level/|X|) instead of X [|X| > level]                   */
#define EGR_FL_NO_INIT_THR  0x40u /* d0 = y (no time-domain threshold before the first transform)                  */
#define EGR_FL_ZERO_STUFF   0x80u /* up-rate by zero insertion (y[i*f] = x[i]) instead of linear interpolation      */
#define EGR_FL_INTERP_LINSPACE 0x100u /* up-rate as numpy.interp(linspace(0, n-1, n_out), arange(n), x): endpoint-inclusive grid, no zero tail */
#define EGR_FL_DEFER_FINALIZE 0x400u /* stop behind out = y + d: autoscale / normalise / NODE_POST wait for egr_fatllama_finalize (channel-parallel runs) */
#define EGR_FL_THR_RECOMPUTE 0x200u /* with THR_RELATIVE: max|X| of EVERY iteration from a read-only pass of its own (the form of rounds 1-5:
                                     * twice the row passes) instead of the maximum the previous iteration's hook carried forward */

/* Host-only planning query (no GPU needed): fills info[] =
 *   {supported (1 = packed real plan, 2 = chirp-z over M = P complex points per state), N, M, M1, M2, TC, nst1, nst2,
 *    radix1[0..13], radix2[0..13], lds_col, lds_row, M3, levels,
 *    [40] chirp-z kind (1 = even/odd packing, one state per channel; 2 = one state per channel pair), [41] transform length D}.
 * m1_hint <= 0 lets the planner choose. */
#define EGR_FL_INFO_LEN 48
int egr_fatllama_plan_query(int64_t n_in, int factor, int m1_hint, int64_t info[EGR_FL_INFO_LEN]);

int egr_fatllama_plan_create(egr_fatllama_plan** out, int64_t n_in, int channels, int factor, int m1_hint,
                             int tc_hint);
/* Output length given explicitly (n_out >= n_in, any ratio): SPEC.md factor_mode "ratio_then_int"; enhance then needs
 * EGR_FL_INTERP_LINSPACE and writes [channels][n_out]. */
int egr_fatllama_plan_create_n(egr_fatllama_plan** out, int64_t n_in, int64_t n_out, int channels);
/* Explicit factorisation N/2 = m1*m2*m3 (m3 = 1: two levels); used by tests and tuning. */
int egr_fatllama_plan_create_ex(egr_fatllama_plan** out, int64_t n_in, int channels, int factor, int m1, int m2, int m3,
                                int tc_hint);
/* Force the chirp-z (Bluestein) path, which egr_fatllama_plan_create selects by itself for lengths the packed real
 * transform cannot take (odd, or N/2 with a prime factor > 13): exact length-N DFTs as length-P convolutions. */
int egr_fatllama_plan_create_bluestein(egr_fatllama_plan** out, int64_t n_in, int channels, int factor);
/* Chirp-z on ANY length (tests, A/B runs).  Lengths without a packed plan -- odd N, N/2 with a prime factor above 13: most real
 * files; the reference transforms the whole file whatever its length, egregora_fat_llama_gpu.py:272-288 -- get the PAIRED form
 * from egr_fatllama_plan_create by themselves: the real signal is first reduced to a complex sequence of length D (kind 1, N even:
 * even/odd samples as re/im, D = N/2, one state per channel; kind 2, N odd: two channels as re/im, D = N, one state per channel
 * pair) and the length-D DFT runs as a cyclic convolution of P >= 2D - 1 points (csrc/egr_fatllama_pz.hip).  kind 0 = that
 * choice, 3 = the legacy full-complex form (P >= 2N - 1 per channel). */
int egr_fatllama_plan_create_chirpz(egr_fatllama_plan** out, int64_t n_in, int channels, int factor, int kind);
int egr_fatllama_plan_destroy(egr_fatllama_plan* plan);

/* x: [channels][n_in] float32, out: [channels][n_in*factor] float32 (both device).
 * Runs: (PCM_IN quantise) -> linear up-rate by `factor` -> d0 = |y|>thr ? y : 0 ->
 *       max_iter x { X = rfft(d); X = |X|>thr ? X : 0; d = irfft(X) } -> out = y + d ->
 *       (autoscale) -> (normalise) -> (NODE_POST).
 * All max_iter iterations are executed (no fixed-point early exit).  The EGR_FL_THR_* / NO_INIT_THR / ZERO_STUFF flags select
 * the other readings of upstream's threshold and interpolation (SPEC.md section 3; oracle: FatLlamaSpec fields of the same
 * names).  EGR_FL_THR_RELATIVE costs ONE extra read-only pass per call, not per iteration: the shrink keeps the spectrum's Hermitian
 * symmetry, so the spectrum iteration i + 1 transforms to IS iteration i's shrunk spectrum (fft(real(ifft(S(X)))) = S(X) up to one
 * float32 transform pair's round-off) and every hook leaves max |S(X)|^2 for its successor; only iteration 0, whose input is the
 * time-domain d0, needs the maximum from a pass of its own (EGR_FL_THR_RECOMPUTE keeps the pass in every iteration). */
int egr_fatllama_enhance(egr_fatllama_plan* plan, const float* x, float* out, int max_iter, float threshold,
                         unsigned flags, void* stream);

/* Channel-parallel runs (this path shards over CHANNELS only -- the reference transforms each channel of the
 * file as one whole signal, egregora_fat_llama_gpu.py:272-288 -- and the channels meet in upstream's joint normalise alone): every rank
 * holds a plan for ITS channels, calls egr_fatllama_enhance with EGR_FL_DEFER_FINALIZE (the loop, out = y + d, per-channel peaks; no
 * autoscale / normalise / write patch / PCM_16 yet), egr_fatllama_joint_peak (this plan's max over channels of the peak after autoscale ->
 * one device float), all-reduces that float with MAX (4 bytes: the only exchange), and egr_fatllama_finalize(flags, the all-reduced
 * float).  max is exact: the ranks' outputs are the single-plan outputs bit for bit.  joint_dev == NULL: the plan's own channels only. */
int egr_fatllama_joint_peak(egr_fatllama_plan* plan, unsigned flags, float* joint_dev, void* stream);
int egr_fatllama_finalize(egr_fatllama_plan* plan, float* out, unsigned flags, const float* joint_dev, void* stream);

/* y = irfft(rfft(x) * gain) per channel on a packed-real plan with factor 1: x, y [channels][N], gain
 * [channels][N/2+1] (real, one value per half-spectrum bin).  Used by the FlashSR input low-pass. */
int egr_spectral_gain(egr_fatllama_plan* plan, const float* x, const float* gain, float* y, void* stream);

/* Per-channel peaks of the last enhance call copied to host: host_pin[c] = max|x_c| on the integer scale (after
 * the optional PCM_16 quantisation), host_pout[c] = max|y_c + d_c| before autoscale / normalise.  Synchronises `stream`. */
/* Hand the plan the stream its second channel pipeline runs on (not owned; one verified with egr_streams_overlap_us to sit on
 * another hardware queue than the caller's stream), and switch the captured-graph replay of the loop on / off (same bits either
 * way; the host may time both and keep the faster). */
int egr_fatllama_set_side_stream(egr_fatllama_plan* plan, void* stream);
int egr_fatllama_set_graph(egr_fatllama_plan* plan, int enable);
int egr_fatllama_last_peaks(egr_fatllama_plan* plan, float* host_pin, float* host_pout, void* stream);
/* Average HIP-event duration (ms) and launch count of the two loop kernels over the last enhance
 * call when profiling was enabled with egr_fatllama_set_profiling(plan, 1). Synchronises. */
int egr_fatllama_set_profiling(egr_fatllama_plan* plan, int enable);
int egr_fatllama_kernel_times(egr_fatllama_plan* plan, double* row_ms_avg, double* col_ms_avg, int64_t* row_launches,
                              int64_t* col_launches);
/* The same by event kind: [0] row pass (packed: k_row; chirp-z: k_pz_rowconv), [1] outer column pass (k_col<1>; k_pzpair),
 * [2] second column pass (inner pass of a three-level plan; chirp-z: the crop pass k_pzcol<1>). */
int egr_fatllama_kernel_times3(egr_fatllama_plan* plan, double ms_avg[3], int64_t launches[3]);

/* Debug builds only (make -C csrc CANARY=1 -> -DEGR_LDS_CANARY): every dynamic LDS region of the Fat-Llama kernels carries a 64-byte
 * guard band on either side, armed at kernel start and checked at kernel exit.  egr_lds_canary_failures: workgroups that found a
 * guard overwritten since the library was loaded (-1: this build has no canaries).  egr_lds_canary_selftest(where): one launch that
 * writes the last payload element (0), one element past the payload (1) or one before it (2); returns the failures it caused. */
long long egr_lds_canary_failures(void);
long long egr_lds_canary_selftest(int where);

/* ------------------------------------------------------------------------------------------------
 * Host-glue arithmetic moved on device (FlashSR node path and the parity yardstick).
 * ---------------------------------------------------------------------------------------------- */

/* y[i] = wrap_int16(rint(x[i]*write_scale)) / read_div      (read_div = 1 keeps the integer scale)
 * Replaces the libsndfile float->PCM_16->float hops at egregora_fat_llama_gpu.py:36 and :291. */
int egr_pcm16_roundtrip(const float* x, float* y, int64_t n, float write_scale, float read_div, void* stream);

/* The stage around the DeepFilterNet model of the reference's Egregora_DeepFilterNet_Denoise (the model itself is upstream's):
 *   egr_dfn_vad_gains : 48 kHz dry signal [C][n48] -> per-10-ms-frame wet / dry gains [C][ceil(n48/480)]: frame RMS, its
 *                       95th percentile (numpy's linear method), clip, one-pole smoothing (smooth_ms > 0), adaptive
 *                       strength (mode 0 off, 1 more_on_noise, 2 more_on_speech, 3 gate_on_noise), equal-power or linear
 *                       curve; float32 roundings exactly where numpy makes them   (egregora_audio_enhance_extras.py:548-606)
 *   egr_dfn_mix       : y = clamp(limit(clip(g_dry[f] dry + g_wet[f] wet, -1, 1) * post_gain)), f = i / hop, limiter =
 *                       ceiling / max|y| over all channels when that peak exceeds the ceiling            (:655-704)
 * workspace: egr_dfn_workspace_bytes(C, n48) bytes; peak_ws: 4 bytes of device memory. */
size_t egr_dfn_workspace_bytes(int channels, int64_t n48);
int egr_dfn_vad_gains(const float* dry48, int channels, int64_t n48, double smooth_ms, int mode, double strength, double amount,
                      double vad_threshold, int linear_curve, void* workspace, float* g_dry, float* g_wet, void* stream);
int egr_dfn_mix(const float* dry, const float* wet, const float* g_dry, const float* g_wet, int channels, int64_t n,
                int64_t n_frames, int hop, float post_gain, int use_gain, int limit, double ceiling, float* y, void* peak_ws,
                void* stream);

/* Linear-interpolation resampler of the "Resample Audio (HQ)" node's fallback branch: y[c][j] = np.interp at
 * j * n_in / n_out input samples, clamped to the last sample (egregora_audio_eval_pack.py:515-519). */
int egr_resample_linear(const float* x, int channels, int64_t n_in, float* y, int64_t n_out, void* stream);

/* GCC-PHAT delay estimate (the null-test suite's _xcorr_delay, egregora_null_test_suite.py:213-237) on the Fat-Llama transform
 * passes: z = a + i b zero-padded to n = 2^k >= na + nb is the packed state of a ONE-channel plan created for 2 n samples
 * (egr_fatllama_plan_create(&plan, 2 n, 1, 1, 0, 0)); the row pass turns it into B conj(A) / (|B conj(A)| + 1e-12), the inverse
 * passes give the correlation.  work: 4 n floats; out4 (device): {first-maximum index of the centred correlation within
 * +-max_shift, relative to n/2, as int bits; its two neighbours and itself} for the host's parabolic refinement. */
int egr_gcc_phat(egr_fatllama_plan* plan, const float* a, int64_t na, const float* b, int64_t nb, int64_t max_shift, float* work,
                 float* out4, void* stream);

/* y = irfft(rfft(x) * [k >= band_lo]) per channel at the plan's own length (packed-real or chirp-z plan, factor 1): the high band
 * of the null-test suite's _band_energy_hi_db (egregora_null_test_suite.py:192-199); with egr_band_sums the one-sided band energies
 * sum_{k >= band_lo} |X[k]|^2 and sum_k |X[k]|^2 of the length-n rfft follow by Parseval:
 *   E = (n * sum v^2 + (sum v)^2 + [n even] (sum (-1)^i v)^2) / 2   for v = y (high band) and v = x (all). */
int egr_band_filter(egr_fatllama_plan* plan, const float* x, int64_t band_lo, float* y, void* stream);
/* out6 (device, double) = {sum x^2, sum x, sum (-1)^i x, sum y^2, sum y, sum (-1)^i y} */
int egr_band_sums(const float* x, const float* y, int64_t n, double* out6, void* stream);

/* K-weighting approximation of the suite's loudness meter (_k_weight, egregora_null_test_suite.py:125-140): one-pole high-pass
 * z = (1-k) x + k z, y = x - z in float32 exactly as numpy's scalar loop rounds, then y[i] += 0.02 (y[i] - y[i-1]).
 * one_minus_k and k are the float32 roundings of the reference's Python floats. */
int egr_kweight(const float* x, int channels, int64_t n, float one_minus_k, float k, float* y, void* stream);
/* y[i] = float32 mean over channels of x[c * stride + i] (numpy .mean(axis=0) order) */
int egr_mono_mean(const float* x, int channels, int64_t stride, int64_t n, float* y, void* stream);
/* out[f] (device, double) = mean over [f hop, min(f hop + block, n)) of mono(x)^2: the 400 ms / 100 ms blocks of integrated_lufs
 * (:143-165) and, with one block of n samples, _rms_db (:119-122) */
int egr_frame_meansq(const float* x, int channels, int64_t n, int64_t block, int64_t hop, int64_t frames, double* out, void* stream);
/* out5 (device, double) = {sum a, sum b, sum ab, sum aa, sum bb} over the mono downmixes of a[ca][n] and b[cb][n] (b scaled by the
 * float32 k first when use_k): least-squares scale and correlation coefficient of Audio Null Test (:431-447) */
int egr_pair_stats(const float* a, int ca, int64_t stride_a, const float* b, int cb, int64_t stride_b, int64_t n, float k, int use_k,
                   double* out5, void* stream);
/* null[c][i] = a[c][i] +- fl32(b[c][i] * k) (minus when invert_b); out2 (device, double) = {sum mono(null)^2, count |null| > 1}
 * (:437-441, 449, 463) */
int egr_null_mix(const float* a, int64_t stride_a, const float* b, int64_t stride_b, int channels, int64_t n, float k, int use_k,
                 int invert_b, float* null_out, double* out2, void* stream);

/* Delay compensation of the null-test suite's aligner: y[c][i] = sum_k h[k] s[i + (taps-1)/2 - k] with s[i] = x[c][i - shift]
 * (zero outside [0, n_in)), i.e. an integer shift followed by np.convolve(., h, "same"); taps = 0 skips the FIR; n_out pads with
 * zeros or crops (_apply_frac_delay_CN + _pad_or_crop_CN, egregora_null_test_suite.py:203-266). */
int egr_shift_fir(const float* x, int channels, int64_t n_in, int64_t shift, const float* h, int taps, float* y, int64_t n_out,
                  void* stream);

/* Evaluation metrics on the device (the parity yardstick of this pack and the reference's "Metrics (LSD + SI-SDR)" node):
 *   egr_lsd_frames  : per[f] = sqrt(mean_k (20 log10(SA[f][k]+1e-12) - 20 log10(SB[f][k]+1e-12))^2 + 1e-12) from two
 *                     frame-major magnitude arrays of egr_stft_mag          (_lsd, egregora_audio_eval_pack.py:405-411)
 *   egr_sum_f64     : *out = sum v[i] in double (the mean over frames)
 *   egr_order_stats2: out2[0..1] = the k_lo-th and k_hi-th smallest of v (0-based; radix selection) -- the two order
 *                     statistics numpy's linear-interpolated percentile needs (p95 of the per-frame LSD)
 *   egr_si_sdr_terms: out4 = {<s_hat,s>, <s,s>, |alpha s|^2, |s_hat - alpha s|^2} on the mono downmixes (mean over the
 *                     cs / csh channels, rows strided by stride_*), alpha = out4[0]/(out4[1]+1e-20)   (_si_sdr, :414-429) */
int egr_lsd_frames(const float* SA, const float* SB, int64_t frames, int nb, float* per, void* stream);
int egr_sum_f64(const float* v, int64_t n, double* out, void* stream);
int egr_order_stats2(const float* v, int64_t n, int64_t k_lo, int64_t k_hi, float* out2, void* stream);
int egr_si_sdr_terms(const float* s, int cs, int64_t stride_s, const float* s_hat, int csh, int64_t stride_sh, int64_t n,
                     double* out4, void* stream);

/* STFT magnitude, Hann-windowed frames, no centring, mono downmix = mean over channels:
 *   frames = 1 + max(0,(n - n_fft)/hop); out: [frames][n_fft/2+1] float32 (frame-major; the reference's
 *   array is the transpose).  window: device float[n_fft] (caller supplies np.hanning(n_fft) so the
 *   window is bit-identical to the reference's).  n_fft even, <= 8192; prime factors of n_fft/2 up to 13 run as register
 *   butterflies, larger ones (e.g. 2176 = 2^7 * 17, any value of the reference widget's 512..8192 step-128 grid) as one generic stage.
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

/* Rational-rate polyphase FIR resampler with the definition and float32 accumulation order of
 * scipy.signal.resample_poly(x, up, down) (zero extension): y[m] = sum_k h[m*down - k*up + half] x[k].
 *   x [channels][n_in], y [channels][n_out], n_out = ceil(n_in*up/down); h: device float[2*half+1], already
 *   multiplied by `up` (the host designs it: firwin(2*half+1, 1/max(up,down), kaiser 5.0), half = 10*max(up,down)).
 * Replaces the scipy branch of _resample_hq at egregora_audio_super_resolution.py:178-187. */
int egr_resample_poly(const float* x, int channels, int64_t n_in, int up, int down, const float* h, int half, float* y,
                      int64_t n_out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * FlashSR network operators (channels-last activations, exact fp32).  Together they stand in for the
 * upstream call `self._model(x, lowpass_input=...)` at egregora_audio_super_resolution.py:366-369; the
 * graph that strings them together is flashsr_engine.py (layer table: flashsr_arch.py).
 * ---------------------------------------------------------------------------------------------- */
#define EGR_ACT_NONE 0
#define EGR_ACT_SILU 1
#define EGR_ACT_TANH 2
#define EGR_ACT_LEAKY01 3
#define EGR_ACT_LOGCLAMP 4 /* log(max(v, act_param)) */

/* Implicit-GEMM convolution on the matrix cores (v_mfma_f32_32x32x2_f32).
 *   x [B][H][W][Cin], y [B][OH][OW][Cout]; w is PRE-PACKED as [ceil(K/16)][Cout][16] with K = KH*KW*Cin ordered
 *   (ky, kx, ci) and zero padded (flashsr_engine.FlashSREngine.pack_matrix);
 *   input coordinate = o*stride + k*(dil for W, 1 for H) - pad; out-of-range reads are zero;
 *   up2 != 0: the logical input is the nearest-neighbour 2x upsampling of x (read at coord>>1);
 *   epilogue: + bias[Cout] + bias_b[B][Cout] + res[B][OH][OW][Cout] (each optional), then activation.
 * 1-D convolutions use H = KH = 1; Linear layers use H = W = KH = KW = 1 with B = rows. */
int egr_conv_nhwc(const float* x, const float* w, const float* bias, const float* bias_b, const float* res, float* y,
                  int B, int H, int W, int Cin, int OH, int OW, int Cout, int KH, int KW, int stride, int dil,
                  int pad_t, int pad_l, int up2, int act, float act_param, void* stream);

/* Same, with output placement: pixel (b, oy, ox) lands at (b, oy*osy + ooy, ox*osx + oox) of an OHF x OWF image
 * (res is read at the same place).  Four calls with 2x2 phase kernels and osy = osx = 2 evaluate
 * "nearest-2x upsample, then 3x3 conv" with 4/9 of the multiplies (flashsr_engine.FlashSREngine.conv3). */
int egr_conv_nhwc_placed(const float* x, const float* w, const float* bias, const float* bias_b, const float* res, float* y,
                         int B, int H, int W, int Cin, int OH, int OW, int Cout, int KH, int KW, int stride, int dil,
                         int pad_t, int pad_l, int up2, int act, float act_param, int osy, int osx, int ooy, int oox,
                         int OHF, int OWF, void* stream);

/* The same convolution on the bf16 matrix pipe with fp32-grade results (csrc/egr_nn_gemm_s3.hip): every fp32 operand is
 * split exactly into three bf16 terms and the six leading partial products are accumulated in fp32 by
 * v_mfma_f32_32x32x16_bf16 (dropped terms < 3 * 2^-24 of each product; measured error vs float64 <= the f32-MFMA kernel's).
 *   egr_split3_pack : packed fp32 weights [nslabs][Cout][16] -> w3 [nslabs][3][Cout][16] bf16 (6 * nslabs * Cout * 16 bytes)
 *   egr_conv_s3     : egr_conv_nhwc_placed on w3; requires Cin % 16 == 0.  nz > 1 runs nz independent problems along
 *                     blockIdx.z with offsets zx / zy (floats) and zw3 (16-byte units of w3), as egr_gemm_zbatched does. */
int egr_split3_pack(const float* w_packed, void* w3, int64_t nslabs, int Cout, void* stream);

/* The same contraction on TWO fp16 terms per operand (scheme 1 of csrc/egr_nn_gemm_s3.hip): x s = h0 + h1 with h0 = f16(x s),
 * h1 = f16(x s - h0); three exact products h0 h0 + h0 h1 + h1 h0 accumulated in fp32 by v_mfma_f32_32x32x16_f16 -- half the matrix
 * instructions of the three-term bf16 scheme; measured error vs float64 <= the bf16 scheme's (tests/test_gpu_split_h2.py).
 * s is a power of two PER BATCH ROW, derived inside the kernel from that row's max |x| so that the row's scaled maximum lies in
 * [2^14, 2^15): no operand can leave fp16's range, every element within 2^-29 of ITS ROW's maximum keeps 22 significand bits
 * (absolute error 2^-40 of the row maximum below that), and a row's result does not depend on the other rows of the batch.
 *   egr_split2h_pack : packed fp32 weights [nslabs][Cout][16] -> w2 [nslabs][2][Cout][16] f16 terms of w * w_scale
 *   egr_absmax       : raises *slot (a float the caller zeroed) to max |x[i]|, i < n  (x 16-byte aligned)
 *   egr_absmax_rows  : row_amax[r] (floats the caller zeroed) raised to max |x| over batch row r: `rows` contiguous runs of per_row
 *                      floats; nz > 1: nz such blocks zx floats apart (the Winograd V layout [nz][rows * tiles][C]), one maximum
 *                      per row over all of them.  per_row % 4 == 0, x 16-byte aligned.
 *   egr_conv_h2      : egr_conv_s3 on w2.  The B * OH * OW GEMM rows form batch_rows equal consecutive groups; group g is scaled
 *                      by the power of two derived from row_amax[g] (device floats: egr_absmax_rows of x, or what the producer
 *                      of x left: egr_winograd4_input_ra) and the epilogue undoes it together with w_scale.  zw2 counts 16-byte
 *                      units of w2.  out_amax (optional, [batch_rows] floats the caller zeroed; nz == 1): raised to max |y| per
 *                      batch row -- the row_amax of the NEXT contraction that reads y.  Nothing about the scales passes through
 *                      the host. */
/* Layout of every row_amax / out_amax array: the maximum of batch row r is the float at index r * EGR_ROW_AMAX_STRIDE (one 128-byte
 * line per row: atomic commits to different rows do not serialise in the L2; the other floats of a line are unused). */
#define EGR_ROW_AMAX_STRIDE 32
int egr_split2h_pack(const float* w_packed, void* w2, int64_t nslabs, int Cout, float w_scale, void* stream);
int egr_absmax(const float* x, int64_t n, float* slot, void* stream);
int egr_absmax_rows(const float* x, int rows, int64_t per_row, int nz, int64_t zx, float* row_amax, void* stream);
int egr_conv_h2(const float* x, const void* w2, const float* bias, const float* bias_b, const float* res, float* y, int B,
                int H, int W, int Cin, int OH, int OW, int Cout, int KH, int KW, int stride, int dil, int pad_t, int pad_l,
                int up2, int act, float act_param, int osy, int osx, int ooy, int oox, int OHF, int OWF, int nz, int64_t zx,
                int64_t zw2, int64_t zy, float w_scale, const float* row_amax, int batch_rows, float* out_amax, void* stream);
/* 3x3 stride-1 pad-1 convolution on two fp16 terms with the producer's GroupNorm (+ SiLU) applied to x while it is loaded, on the
 * input-stationary kernel (a workgroup splits the halo patch of its 4 x 32 pixel tile once and reads the nine taps from it; x is
 * read once, no Winograd intermediates): H % 4 == 0, W % 32 == 0, Cin % 32 == 0, >= 512 tiles, else EGR_ERR_UNSUPPORTED with
 * nothing launched.  row_amax[b] bounds max |GroupNorm(x)| of image b from above (egr_gn_operand_bound: from the coefficients and
 * the row maxima of x, no pass over x); the scale derived from it only has to place the row's maximum inside [1, 2^15).
 * gn_part (optional, [B * H * W / 32][Cout / 4] float2): (sum, sum of squares) of y per 32-pixel row segment and channel quad, for
 * egr_groupnorm_stats_from_partials(part, B, H * W / 32, Cout, G). */
int egr_conv_h2_gn(const float* x, const float* gn_scale, const float* gn_shift, int gn_silu, const void* w2, const float* bias,
                   const float* res, float* y, int B, int H, int W, int Cin, int Cout, int act, float w_scale, const float* row_amax,
                   float* out_amax, void* gn_part, void* stream);
int egr_gn_operand_bound(const float* scale, const float* shift, int B, int C, const float* x_row_amax, float* bound, void* stream);
int egr_conv_s3(const float* x, const void* w3, const float* bias, const float* bias_b, const float* res, float* y, int B,
                int H, int W, int Cin, int OH, int OW, int Cout, int KH, int KW, int stride, int dil, int pad_t, int pad_l,
                int up2, int act, float act_param, int osy, int osx, int ooy, int oox, int OHF, int OWF, int nz, int64_t zx,
                int64_t zw3, int64_t zy, void* stream);

/* Winograd F(2x2,3x3) for stride-1 pad-1 3x3 convolutions with many channels (2.25x fewer multiplies):
 *   egr_winograd_input : x [B][H][W][C] -> V [16][P][C], P = B*ceil(H/2)*ceil(W/2) tiles, V[4i+j] = (B^T d B)[i][j]
 *   egr_gemm_zbatched  : M[xi] = V[xi] ([P][Cin]) x U[xi] ([Cin][Cout], each packed like egr_conv_nhwc weights), xi < nz
 *   egr_winograd_output: y [B][H][W][N] = act(A^T M A + bias + res)        (act: 0 none, 1 SiLU)
 * U = G g G^T is prepared by the host once per layer (flashsr_engine.FlashSREngine.add_winograd). */
int egr_winograd_input(const float* x, const float* gn_scale, const float* gn_shift, int gn_silu, int B, int H, int W, int C,
                       float* V, void* stream);   /* gn_*: optional fused producer GroupNorm (+SiLU), [B][C] each */
int egr_gemm_zbatched(const float* x, const float* w, float* y, int nz, int rows, int Cin, int Cout, int64_t zx, int64_t zw,
                      int64_t zy, void* stream);
int egr_winograd_output(const float* M, const float* bias, const float* res, float* y, int B, int H, int W, int N, int act,
                        void* stream);

/* Strided batched C[b1][b2] = alpha * A (M x K, lda) * B^T (B is N x K, ldb) with BOTH fp32 operands split into three bf16
 * terms by the loader (attention: Q K^T, and P V after egr_transpose_batched of V).  K % 16 == 0, 16-byte aligned rows;
 * otherwise EGR_ERR_UNSUPPORTED (callers fall back to egr_bgemm). */
int egr_bgemm_nt_s3(const float* a, const float* b, float* c, int nb1, int nb2, int M, int N, int K, int lda, int ldb, int ldc,
                    int64_t sa1, int64_t sa2, int64_t sb1, int64_t sb2, int64_t sc1, int64_t sc2, float alpha, void* stream);
/* The same product on TWO fp16 terms per operand (ABI 4; both operands are activations): the operands of outer batch index b1 -- the
 * batch row -- are scaled by powers of two derived in the kernel from a_amax[b1] / b_amax[b1] (row_amax layout: egr_absmax_rows or a
 * producer's out_amax); out_amax (optional, zeroed by the caller): max |C| per b1. */
int egr_bgemm_nt_h2(const float* a, const float* b, float* c, int nb1, int nb2, int M, int N, int K, int lda, int ldb, int ldc,
                    int64_t sa1, int64_t sa2, int64_t sb1, int64_t sb2, int64_t sc1, int64_t sc2, float alpha, const float* a_amax,
                    const float* b_amax, float* out_amax, void* stream);

/* Winograd F(4x4,3x3) (4x fewer multiplies; H and W multiples of 4): same three steps with 36 components,
 *   V [36][P][C], P = B*(H/4)*(W/4) tiles, V[6i+j] = (B^T d B)[i][j] of the 6x6 input tile at (4ty-1, 4tx-1);
 *   36 z-batched GEMMs (egr_conv_s3 / egr_gemm_zbatched with nz = 36); y = act(A^T M A + bias + res) per 4x4 output tile.
 * Cook-Toom points 0, +-3/4, +-3/2, inf (smaller error amplification than 0, +-1, +-2; csrc/egr_nn_wino4.hip);
 * egr_winograd4_g returns the matching G (6x3, row-major, double) and U = G g G^T (6x6 per channel pair) is prepared by
 * the host in float64 (flashsr_engine.FlashSREngine.add_winograd). */
int egr_winograd4_g(double* g18);
int egr_winograd4_input(const float* x, const float* gn_scale, const float* gn_shift, int gn_silu, int B, int H, int W, int C,
                        float* V, void* stream);
/* same, and row_amax[b] (optional; floats the caller zeroed) is raised to max |V| over image b -- the operand maxima egr_conv_h2
 * wants for the 36 GEMMs that read V, without another pass over V */
int egr_winograd4_input_ra(const float* x, const float* gn_scale, const float* gn_shift, int gn_silu, int B, int H, int W, int C,
                           float* V, float* row_amax, void* stream);
int egr_winograd4_output(const float* M, const float* bias, const float* res, float* y, int B, int H, int W, int N, int act,
                         void* stream);
/* GroupNorm statistics of y without another pass over it: egr_winograd4_output_stats also writes every thread's (sum, sum of
 * squares) over its 4x4 pixels x 4 channels to part [B*(H/4)*(W/4)][N/4] float2 (1/32 of y's bytes);
 * egr_groupnorm_stats_from_partials reduces them to stats [B][G][2] double in a fixed order (no atomics; (N/G) % 4 == 0);
 * egr_groupnorm_coeff_from_stats turns statistics into the per-(b, c) scale / shift the consumer applies. */
int egr_winograd4_output_stats(const float* M, const float* bias, const float* res, float* y, int B, int H, int W, int N,
                               int act, void* part, void* stream);
/* either output transform (part may be NULL) with row_amax[b] (optional; floats the caller zeroed) raised to max |y| of image b:
 * the operand maxima egr_conv_h2 wants when y feeds a 1x1 / strided / phase convolution directly */
int egr_winograd4_output_ra(const float* M, const float* bias, const float* res, float* y, int B, int H, int W, int N, int act,
                            void* part, float* row_amax, void* stream);
int egr_groupnorm_stats_from_partials(const void* part, int B, int tiles_per_image, int C, int G, double* stats, void* stream);
int egr_groupnorm_coeff_from_stats(const double* stats, const float* gamma, const float* beta, int B, int HW, int C, int G,
                                   float eps, float* scale, float* shift, void* stream);

/* GroupNorm split in two: egr_groupnorm_coeff computes the statistics and the per-(b, c) scale/shift ([B][C] each);
 * egr_conv_nhwc_gn is egr_conv_nhwc (no dilation / upsample / placement) with x*scale + shift (+SiLU) applied to the
 * input while it is loaded (zero padding after it), so the normalised tensor is never materialised. */
int egr_groupnorm_coeff(const float* x, const float* gamma, const float* beta, int B, int HW, int C, int G, float eps,
                        void* workspace, float* scale, float* shift, void* stream);
int egr_conv_nhwc_gn(const float* x, const float* gn_scale, const float* gn_shift, int gn_silu, const float* w,
                     const float* bias, const float* res, float* y, int B, int H, int W, int Cin, int OH, int OW, int Cout,
                     int KH, int KW, int stride, int pad_t, int pad_l, int act, void* stream);

/* Strided batched GEMM for attention: C[b1][b2] = alpha * A[b1][b2] (MxK) * (transB ? B^T : B). */
int egr_bgemm(const float* a, const float* b, float* c, int nb1, int nb2, int M, int N, int K, int lda, int ldb,
              int ldc, int64_t sa1, int64_t sa2, int64_t sb1, int64_t sb2, int64_t sc1, int64_t sc2, int transB,
              float alpha, void* stream);

/* GroupNorm over [B][HW][C] with G groups (+ optional SiLU); workspace >= egr_groupnorm_workspace_bytes. */
size_t egr_groupnorm_workspace_bytes(int B, int C, int G);
int egr_groupnorm_nhwc(const float* x, const float* gamma, const float* beta, float* y, int B, int HW, int C, int G,
                       float eps, int silu, void* workspace, void* stream);
int egr_layernorm_rows(const float* x, const float* gamma, const float* beta, float* y, int64_t rows, int C, float eps,
                       void* stream);
int egr_softmax_rows(float* x, int64_t rows, int cols, void* stream);
/* op: 0 a+b, 1 s0*a+s1*b, 2 silu(a), 3 s0*a, 4 copy, 5 s0*(a+b) */
int egr_eltwise(const float* a, const float* b, float* y, int64_t n, int op, float s0, float s1, void* stream);
int egr_geglu(const float* u, float* y, int64_t rows, int D, void* stream);
/* The `_ra` forms of the element-wise operators (ABI 4): same outputs; the tensor is `batch_rows` equal consecutive runs (the rows of
 * the batch) and row_amax[b] (optional; floats the caller zeroed) is raised to max |y| over run b -- the operand maxima of the
 * split contraction that reads y (egr_conv_h2) without a pass over y. */
int egr_groupnorm_nhwc_ra(const float* x, const float* gamma, const float* beta, float* y, int B, int HW, int C, int G, float eps,
                          int silu, void* workspace, float* row_amax, void* stream);
int egr_layernorm_rows_ra(const float* x, const float* gamma, const float* beta, float* y, int64_t rows, int C, float eps,
                          int batch_rows, float* row_amax, void* stream);
int egr_eltwise_ra(const float* a, const float* b, float* y, int64_t n, int op, float s0, float s1, int batch_rows, float* row_amax,
                   void* stream);
int egr_geglu_ra(const float* u, float* y, int64_t rows, int D, int batch_rows, float* row_amax, void* stream);
int egr_concat_channels_ra(const float* a, const float* b, float* y, int64_t M, int C1, int C2, int batch_rows, float* row_amax,
                           void* stream);
/* Wall time (us) of two spin_us-long busy kernels launched back to back on streams a and b: about spin_us when the two streams
 * run concurrently (different hardware queues), about twice that when the runtime multiplexes them onto one queue.  The FlashSR
 * engine uses it once per process to pick side streams that really overlap with the caller's stream. */
int egr_streams_overlap_us(void* stream_a, void* stream_b, int spin_us, double* elapsed_us);

/* y[b,oy,ox,co] = bias[co] + sum_{ky,kx} P[b, oy+ky-pad_t, ox+kx-pad_l][(ky*KW+kx)*Cout + co] over in-image source pixels: the second
 * half of a kh x kw convolution with very few outputs computed as a 1x1 contraction onto per-tap partial products (VAE conv_out). */
int egr_tap_gather(const float* P, const float* bias, float* y, int B, int H, int W, int KH, int KW, int Cout, int pad_t, int pad_l,
                   void* stream);
/* Stride-1 convolution of a ONE-channel image, fp32 FMAs on the vector ALU (VAE conv_in); w_packed: [Cout][16], k = ky*KW+kx first. */
int egr_conv_cin1(const float* x, const float* w_packed, const float* bias, float* y, int B, int H, int W, int Cout, int KH, int KW,
                  int pad_t, int pad_l, void* stream);
int egr_concat_channels(const float* a, const float* b, float* y, int64_t M, int C1, int C2, void* stream);
int egr_transpose_batched(const float* x, float* y, int batch, int R, int Cc, void* stream);
/* Anti-aliased snake over [B][L][C]: 2x up (K-tap FIR, replicate pad) . x + sin^2(e^alpha x)/(e^beta+1e-9) . 2x down. */
int egr_snake_aa(const float* x, const float* alpha, const float* beta, const float* filt, float* y, int B, int L, int C,
                 int K, void* stream);
/* same, and row_amax[b] (optional; floats the caller zeroed) is raised to max |y| of batch row b */
int egr_snake_aa_ra(const float* x, const float* alpha, const float* beta, const float* filt, float* y, int B, int L, int C,
                    int K, float* row_amax, void* stream);
/* ConvTranspose1d as GEMM + gather: Y [B][Lin][K][Co] -> out [B][Lout][Co] (+bias, + optional add). */
int egr_col2im_convtr1d(const float* Y, const float* bias, const float* add, float* out, int B, int Lin, int Lout, int Co,
                        int K, int r, int pad, void* stream);
/* Mel front-end STFT: x [B][L] -> |STFT| frames [B][T][ldm], reflect padding rpad, frames >= t_valid zeroed. */
int egr_stft_frames(const float* x, int B, int L, int n_fft, int hop, int rpad, int T, int t_valid, int ldm,
                    const float* window, float* mag, void* stream);
/* FlashSR `lowpass_input=True`: per row, cutoff bin from STFT magnitudes mag [B][T][ldm] (highest bin whose
 * cumulative time-summed energy is below pct of the total, + 1) -> cut_out[B]; then the zero-phase Chebyshev-I
 * amplitude gain 1/(1 + eps^2 T_order^2(tan(pi f/sr)/tan(pi fc/sr))) on `nbins` bins 0..sr/2 -> gain [B][nbins]. */
int egr_lowpass_gain(const float* mag, int B, int T, int ldm, int nb, float pct, float sr, int order, float ripple_db,
                     int64_t nbins, int* cut_out, float* gain, void* stream);
/* Standard normals: element e of row r is a function of (seed, row_ids[r] or r, e) only (Philox4x32-10). */
int egr_randn(float* out, int64_t per_row, int rows, uint64_t seed, const int64_t* row_ids, void* stream);
/* same; without row_ids the id of row r is id_base + r */
int egr_randn_base(float* out, int64_t per_row, int rows, uint64_t seed, const int64_t* row_ids, int64_t id_base, void* stream);

/* ------------------------------------------------------------------------------------------------
 * FlashSR model handle -- the inner boundary of SURVEY.md section 8(b).
 * Replaces the two upstream calls the reference makes:
 *   egr_flashsr_create  <->  FlashSR(student_ldm.pth, sr_vocoder.pth, vae.pth); model.eval(); model.to(dev)
 *                            (egregora_audio_super_resolution.py:346-359; built once, not on every run() as at :393)
 *   egr_flashsr_infer   <->  model(x[C, 245760] on the device, lowpass_input=bool) under torch.inference_mode()  (:361-369)
 * The graph walk (layer table -> operator launches) and its scratch memory live in the library (csrc/egr_flashsr.cpp); the host
 * keeps checkpoint I/O: it hands over named fp32 tensors in torch layouts, using the layer-table names of flashsr_arch.py
 * (conv weights [Co][Ci][kh][kw], ConvTranspose1d [Ci][Co][k], linear [Co][Ci], ...) plus four derived constants
 * "const.window" [n_fft], "const.mel_fb" [n_mels][n_fft/2+1], "const.aa_filter" [aa_taps], "const.time_emb" [unet_ch].
 * Handles are not thread-safe; one per device.  All work is enqueued on the stream given to each call.
 * ---------------------------------------------------------------------------------------------- */
typedef struct egr_flashsr egr_flashsr;
#define EGR_FSR_MAX 8
typedef struct egr_flashsr_config {          /* mirrors flashsr_arch.FlashSRConfig field by field */
    int struct_bytes;                        /* sizeof(egr_flashsr_config): ABI check */
    int sr, chunk, n_fft, hop, n_mels, n_frames;
    float fmin, fmax, log_floor;
    int vae_ch, vae_levels, vae_mult[EGR_FSR_MAX], vae_res, z_ch, gn_groups;
    int unet_ch, unet_levels, unet_mult[EGR_FSR_MAX], unet_res, unet_n_attn, unet_attn_ds[EGR_FSR_MAX], head_dim, t_steps;
    int voc_ch, voc_n_rates, voc_rates[EGR_FSR_MAX], voc_n_kernels, voc_kernels[EGR_FSR_MAX], voc_n_dils, voc_dils[EGR_FSR_MAX], aa_taps;
} egr_flashsr_config;
typedef struct egr_tensor_desc {
    const char* name;
    const float* data;                       /* DEVICE pointer, fp32, contiguous */
    int ndim;                                /* 0..4 */
    int64_t shape[4];
} egr_tensor_desc;
/* create flags (default 0 = the shipped configuration) */
#define EGR_FSR_F32_MFMA       0x01u /* every contraction on v_mfma_f32_32x32x2_f32 instead of the exact 3-way bf16 split */
#define EGR_FSR_NO_WINOGRAD    0x02u
#define EGR_FSR_NO_WINO_F4     0x04u
#define EGR_FSR_NO_GN_PARTIALS 0x08u
#define EGR_FSR_NO_THIN_ENDS   0x10u
#define EGR_FSR_NO_FUSE_GN     0x20u
#define EGR_FSR_SPLIT_BF16X3   0x40u /* egr_flashsr_infer never uses the two-term fp16 operand scheme (no fp16 weight terms are kept) */
int egr_flashsr_default_config(egr_flashsr_config* cfg);       /* the declared full-size table */
int egr_flashsr_create(egr_flashsr** out, const egr_flashsr_config* cfg, const egr_tensor_desc* tensors, int n_tensors,
                       unsigned flags, void* stream);           /* repacks on `stream`, synchronises before returning */
/* The same from an "EGRW0001" weight-blob file (config + named tensors; written by flashsr_weights.write_blob from the three
 * upstream checkpoints): the form a host without Python links against. */
int egr_flashsr_create_from_file(egr_flashsr** out, const char* path, unsigned flags, void* stream);
int egr_flashsr_destroy(egr_flashsr* h);
/* x, y: [rows][chunk] fp32 device; rows = chunks x channels ride the batch dimension (reference :366-368) and are processed
 * rows-per-pass at a time.  The 1-step diffusion noise of row r depends only on (seed, row_ids[r]) -- row_ids: device int64[rows],
 * or NULL for 0..rows-1 -- so the result is independent of pass and rank boundaries (the reference never seeds). */
int egr_flashsr_infer(egr_flashsr* h, const float* x, int rows, int lowpass_input, uint64_t seed, const int64_t* row_ids, float* y,
                      void* stream);
/* One pass with caller-supplied noise [rows][lat_h][lat_w][z_ch] (channels-last); stages: NULL or 6 device pointers (each may be
 * NULL) receiving mel [R][T][n_mels], z_cond, v, z0 [R][h][w][z], mel_hat [R][T][n_mels], and the vocoder's full output. */
int egr_flashsr_forward(egr_flashsr* h, const float* x, const float* noise, int rows, int lowpass_input, float* y, float* const* stages,
                        void* stream);
int egr_flashsr_set_rows_per_pass(egr_flashsr* h, int rows);
/* Operand scheme of egr_flashsr_infer's split contractions.  Default (scheme 1): two fp16 terms per operand (egr_conv_h2) with
 * one power-of-two scale per (tensor, batch row), derived on the device from that row's own maximum (left by the tensor's
 * producer, or measured by egr_absmax_rows when the tensor first feeds a split contraction).  Consequences: the output row r is a
 * function of (weights, x[r], seed, id of r) -- not of the handle's history, of the other rows of the call, or of how rows are
 * spread over ranks (tile / split-K choices still follow the row count of a forward: fp32 round-off only); no operand can leave
 * fp16's range, so nothing is verified or run twice; a quiet row next to a loud one keeps its own precision.
 * egr_flashsr_set_split(h, 0) or creation flag EGR_FSR_SPLIT_BF16X3 or EGREGORA_FLASHSR_SPLIT=bf16x3 keep every call on the
 * three-term bf16 kernels; egr_flashsr_forward always is, unless egr_flashsr_set_split(h, 2) (stage taps of the fp16 path for
 * the tests).  The mel projection and the attention products always run on bf16 terms.
 * egr_flashsr_split_info: enabled, contraction weights holding fp16 terms, egr_flashsr_infer calls made on the scheme.
 *
 * WHEN egr_flashsr_infer BLOCKS THE HOST: it does not, in the steady state -- all work is enqueued on `stream` (and on the
 * handle's side streams, forked from and joined to `stream` by events).  The FIRST call with a given (row count, lowpass) shape
 * allocates scratch (hipMalloc, which synchronises the device), creates the low-pass plans and, once per caller stream, times
 * two spin kernels to verify its side streams (~1 ms).  x and y must stay valid until the work has run, as for any kernel. */
int egr_flashsr_set_split(egr_flashsr* h, int scheme);
int egr_flashsr_split_info(egr_flashsr* h, int* enabled, int* weights, int64_t* calls);
/* Concurrent row groups inside egr_flashsr_infer: a pass of >= 2 * min_group_rows rows is split into up to max_groups (1..4,
 * default 2; EGREGORA_FLASHSR_STREAMS) contiguous groups run as simultaneous forwards on side streams the handle creates and
 * verifies to sit on other hardware queues than the caller's; fork / join by events, so the call still looks single-stream. */
int egr_flashsr_set_streams(egr_flashsr* h, int max_groups, int min_group_rows);
/* HIP-event timing of the MFMA contraction launches, aggregated per kernel instantiation (bench.py's roofline): switch on, run,
 * then read entry `index` (kind_buf NULL: only *count).  egr_flashsr_flop_count: executed dense flops of one pass over `rows`. */
int egr_flashsr_set_profiling(egr_flashsr* h, int enable);
int egr_flashsr_profile(egr_flashsr* h, int index, char* kind_buf, size_t buflen, int64_t* launches, double* flops, double* ms, int* count);
int egr_flashsr_flop_count(egr_flashsr* h, int rows, double* flops, void* stream);
int64_t egr_flashsr_scratch_bytes(egr_flashsr* h);
/* Scratch budget in bytes (0 = none; EGREGORA_FLASHSR_ARENA_GB sets it at creation).  A 26-row pass of the full-size table in two
 * row groups holds ~39 GB of activations; under a budget egr_flashsr_infer runs fewer rows per pass (same results, rows are
 * independent) instead of failing an allocation next to the host's other models.  Allocated arenas are not returned. */
int egr_flashsr_set_arena_cap(egr_flashsr* h, double bytes);
/* One throw-away egr_flashsr_infer over `rows` rows of silence, synchronised before it returns: the handle's kernels are loaded
 * (HIP loads a kernel's code object at its first launch: ~200 ms of host time for the graph's ~150 instantiations), the scratch
 * arenas and row-maxima pools for that row count exist and -- for rows >= 2 * min_group_rows -- the side streams are verified, so
 * the host's first real call costs what every later call costs.  Replaces nothing in the reference (it rebuilds the model per call,
 * egregora_audio_super_resolution.py:393); the engine calls it once when the handle is built (flashsr_engine.FlashSREngine.warmup). */
int egr_flashsr_warmup(egr_flashsr* h, int rows, void* stream);
/* One (kernel size k, dilation d) unit of a 16-channel AMP block of the vocoder as ONE kernel (csrc/egr_nn_amp.hip):
 *   y = conv2(snake2(conv1(snake1(x)))) + x,  x, y [B][L][C = 16] channels-last fp32, y != x
 * snake = the anti-aliased activation of egr_snake_aa (12-tap FIR), conv1 = k taps at dilation d, conv2 = k taps at dilation 1, both
 * 'same'-padded with zeros; w?_h2 = egr_split2h_pack(slab-major pack of the conv weight, w?_scale).  Stands in for the four launches
 * (snake, conv, snake, conv + residual) of the upstream vocoder's AMP unit as the reference reaches it through FlashSR.__call__
 * (egregora_audio_super_resolution.py:361-369).  EGR_ERR_UNSUPPORTED (nothing launched) outside C = 16, odd k <= 11, d (k - 1) / 2 <= 25,
 * aa_taps = 12; EGR_ERR_ARG for L < 16 or B outside 1 .. 65535; x and y 16-byte aligned. */
int egr_amp_unit_h2(const float* x, float* y, int B, int L, int C, int k, int d, const float* alpha1, const float* beta1, const void* w1_h2,
                    float w1_scale, const float* bias1, const float* alpha2, const float* beta2, const void* w2_h2, float w2_scale,
                    const float* bias2, const float* filt, int aa_taps, void* stream);
/* Weight repacking shared by the handle and the Python graph driver (csrc/egr_flashsr_pack.hip):
 *   egr_pack_weight     : torch layout -> slab-major [ceil(K/16)][N][16]; layout 0 conv/linear [N][Ci][KH][KW] (k = (ky KW + kx) Ci + ci),
 *                         1 ConvTranspose1d [K=Ci][Co][KW] (n = kk Co + co), 2 per-tap products [Co][K=Ci][KH][KW] (n = tap Co + co)
 *   egr_phase_weights   : [Co][Ci][3][3] -> four [Co][Ci][2][2] phase kernels of "nearest-2x then 3x3" (dst [4][Co][Ci][2][2])
 *   egr_winograd_pack_u : U = G g G^T in double per (co, ci), np = 4 (F(2x2,3x3)) or 6 (F(4x4,3x3)); G_dev: np x 3 doubles on the
 *                         device; dst [np*np][ceil(Ci/16)][Co][16] */
int egr_pack_weight(const float* src, float* dst, int layout, int K, int N, int Ci, int Co, int KH, int KW, void* stream);
int egr_phase_weights(const float* w_oihw, float* dst4, int Co, int Ci, void* stream);
int egr_winograd_pack_u(const float* w_oihw, float* dst, const double* G_dev, int np, int Co, int Ci, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EGREGORA_AMD_H */
