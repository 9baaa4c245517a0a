"""The reference's null-test suite (egregora_null_test_suite.py, fixtures G13 / G14) on this pack's kernels: `Audio Align (XCorr)`,
`Audio Gain Match`, `Audio Null Test`, `Audio Plotter`, `Null Test (Full)` with the reference's mapping keys, INPUT_TYPES /
RETURN_TYPES / RETURN_NAMES / FUNCTION / CATEGORY and `execute` signatures.

Where the work runs: the GCC-PHAT delay estimate and the band energy on the Fat-Llama whole-signal transform passes (egr_gcc_phat,
egr_band_filter); shift + fractional FIR, rate match, K-weighting, block energies, correlation / least-squares sums, the null mix,
STFT magnitudes and the log-spectral distance in egr_glue.hip kernels.  The host keeps what is O(frames) or O(1): the loudness gate
over the 400 ms blocks, dB conversions, the metric dictionary, and matplotlib rendering of the plots.  `Null Test (Full)` keeps the
signals on the device between its stages.

Reference behaviours kept on purpose (the fixtures pin them): the delay is the true lag minus one (Q8, device_ops.xcorr_delay);
the fractional FIR always DELAYS by frac = |d| - floor(|d|) whatever the sign of d, and an even tap count adds the half sample of
np.convolve(..., "same") (Q9); the K-weighting recurrence rounds to float32 at every step (numpy >= 2 scalar rules).  Not
reproduced: the aligner's debug IMAGE is the reference's own fallback (a blank 8x8 image) instead of a correlation plot.
"""
import io
import math

import numpy as np
import torch

from . import device_ops, native
from .audio_glue import eval_audio, eval_package


def _opt(default, lo, hi, step):
    return dict(zip(("default", "min", "max", "step"), (default, lo, hi, step)))


_AUDIO, _FLAG = ("AUDIO", {}), lambda d: ("BOOLEAN", {"default": d})
_STFT = {"n_fft": ("INT", _opt(2048, 512, 8192, 128)), "hop": ("INT", _opt(512, 64, 4096, 64))}
_METRIC_FLAGS = {"compute_corr": _FLAG(True), "compute_null_rms": _FLAG(True), "compute_null_lufs": _FLAG(True), "compute_lsd": _FLAG(True),
                 "compute_hf_residual": _FLAG(False)}
_DRAW_FLAGS = {"draw_waveforms": _FLAG(True), "draw_spectrograms": _FLAG(True), "draw_diffspec": _FLAG(True)}


def blank_image(h=8, w=8):
    return torch.zeros((1, h, w, 3), dtype=torch.float32)


def _device_audio(x):
    """AUDIO -> (package, [C,N] float32 CUDA tensor)."""
    pkg = eval_audio(x)
    return pkg, torch.from_numpy(np.ascontiguousarray(pkg["samples"])).cuda()


def _at_rate(x, sr_from, sr_to):
    """np.interp rate match both the aligner and the gain matcher apply to the second input (:299-308, :365-373)."""
    if sr_from == sr_to:
        return x
    return device_ops.resample_linear(x, int(round(x.shape[1] * sr_to / sr_from)))


# ------------------------------------------------------------------------------------------------ stages on device tensors
def frac_delay_taps(frac: float, taps: int) -> np.ndarray:
    """Hann-windowed sinc centred (taps-1)/2 + frac, unit DC gain, float32 -- the reference's FIR (:255-262)."""
    m = max(16, int(taps))
    n = np.arange(m)
    h = (np.sinc(n - (m - 1) / 2.0 - frac) * np.hanning(m)).astype(np.float32)
    return h / np.sum(h)


def apply_delay(x: torch.Tensor, delay_samples: float, taps: int, n_out: int) -> torch.Tensor:
    """[C,N] CUDA -> [C,n_out]: reference _apply_frac_delay_CN followed by _pad_or_crop_CN."""
    C, N = x.shape
    y = torch.empty((C, n_out), dtype=torch.float32, device=x.device)
    shift, h, m = 0, None, 0
    if abs(delay_samples) >= 1e-6:
        int_d = int(math.floor(abs(delay_samples)))
        frac = abs(delay_samples) - int_d
        shift = int_d if delay_samples >= 0 else -int_d
        if int_d >= N:
            shift = N if delay_samples >= 0 else -N          # everything shifted out: zeros, as the reference's guards leave them
        if frac > 1e-6:
            hh = frac_delay_taps(frac, taps)
            h, m = torch.from_numpy(hh).to(x.device), hh.size
    native.check(native.lib().egr_shift_fir(native.ptr(x), C, N, shift, native.ptr(h) if h is not None else None, m,
                                            native.ptr(y), n_out, native.stream_ptr()), "egr_shift_fir")
    return y


def align_stage(xr, xp, sr, max_shift_ms=200, fractional=True, fir_len=64):
    """-> (xp aligned to xr's length, delay in samples); xp already at rate sr."""
    n = min(xr.shape[1], xp.shape[1])
    lag = float(device_ops.xcorr_delay(device_ops.mono_mean(xr, n), device_ops.mono_mean(xp, n), sr, int(sr * (max_shift_ms / 1000.0))))
    comp = -lag if fractional else -round(lag)
    return apply_delay(xp.contiguous(), float(comp), fir_len, xr.shape[1]), lag


def gain_stage(xr, xi, sr, mode="LUFS-I", max_gain_db=12.0):
    """-> (xi * gain, gain_db, ref_level, in_level) (:375-388)."""
    if str(mode).upper().startswith("LUFS"):
        ref_level, in_level = device_ops.integrated_lufs(xr, sr), device_ops.integrated_lufs(xi, sr)
    else:
        ref_level, in_level = device_ops.rms_db(xr), device_ops.rms_db(xi)
    gain_db = float(np.clip(ref_level - in_level, -abs(max_gain_db), abs(max_gain_db)))
    return device_ops.scale(xi, 10 ** (gain_db / 20.0)), gain_db, float(ref_level), float(in_level)


def null_stage(A, B, sr, invert_b=True, least_squares_scale=False, compute_corr=True, compute_null_rms=True, compute_null_lufs=True,
               compute_lsd=True, compute_hf_residual=False, n_fft=2048, hop=512, hf_band_hz=8000):
    """-> (null [C,n] on the device, metrics in the reference's key order) (:426-470)."""
    n = min(A.shape[1], B.shape[1])
    k = None
    if least_squares_scale:
        _, _, ab, _, bb = device_ops.pair_stats(A, B, n)
        k = float(ab / float(bb + 1e-20))
    null, null_ss, overs = device_ops.null_mix(A, B, n, k, invert_b)
    m = {}
    if compute_corr:
        sa, sb, ab, aa, bb = device_ops.pair_stats(A, B, n, k)
        cov = ab - sa * sb / n
        den = math.sqrt(max(aa - sa * sa / n, 0.0)) * math.sqrt(max(bb - sb * sb / n, 0.0)) + 1e-20
        m["corr_coef"] = float(np.float32((cov if invert_b else -cov) / den))
    if compute_null_rms:
        m["null_rms_dbfs"] = 10.0 * math.log10(null_ss / n + 1e-20)
    if compute_null_lufs:
        m["null_lufs"] = device_ops.integrated_lufs(null, sr)
    if compute_lsd:
        Bs = B[:, :n].contiguous()
        m["lsd_mean_db"], m["lsd_p95_db"] = device_ops.lsd(A[:, :n].contiguous(), Bs if k is None else device_ops.scale(Bs, k), n_fft, hop)
    if compute_hf_residual:
        m["hf_residual_db"] = device_ops.band_energy_hi_db(null, sr, hf_band_hz)
    m["overshoot_count"] = int(overs)
    m["clipped_pct"] = float(100.0 * overs / null.numel())
    m["scale_k"] = float(1.0 if k is None else k)
    return null, m


# ------------------------------------------------------------------------------------------------ plots (host rendering)
def figure_image(fig) -> torch.Tensor:
    """matplotlib figure -> IMAGE [1,H,W,3] in 0..1: tight bounding box at 110 dpi through a PNG, as the reference's (:97-113)."""
    from PIL import Image
    png = io.BytesIO()
    fig.savefig(png, format="png", bbox_inches="tight", dpi=110)
    fig.clf()
    png.seek(0)
    return torch.from_numpy(np.asarray(Image.open(png).convert("RGB"), dtype=np.float32) / 255.0)[None]


def plot_stage(a, b, null, draw_waveforms=True, draw_spectrograms=True, draw_diffspec=True, n_fft=2048, hop=512):
    """a, b, null: mono CUDA signals of one length.  STFT magnitudes come from egr_stft_mag; drawing is matplotlib's (:499-566)."""
    if not (draw_waveforms or draw_spectrograms or draw_diffspec):
        return blank_image(1, 1), blank_image(1, 1), blank_image(1, 1)
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    spec_db = lambda y: 20 * np.log10(device_ops.stft_mag(y, n_fft, hop).cpu().numpy() + 1e-9)
    waves = specs = diff = None
    if draw_waveforms:
        fig, axes = plt.subplots(3, 1, figsize=(10, 6), sharex=True)
        t = np.arange(a.numel())
        for ax, y, title in zip(axes, (a, b, null), ("A: original", "B: processed", "Null: A−B")):
            ax.plot(t, y.cpu().numpy(), linewidth=0.7)
            ax.set_ylim(-1.05, 1.05)
            ax.set_title(title)
            ax.grid(alpha=0.25)
        axes[-1].set_xlabel("samples")
        fig.tight_layout()
        waves = figure_image(fig)
    if draw_spectrograms:
        fig, axes = plt.subplots(3, 1, figsize=(10, 7))
        for ax, y, title in zip(axes, (a, b, null), ("A: spec", "B: spec", "Null: spec")):
            ax.imshow(spec_db(y), origin="lower", aspect="auto")
            ax.set_title(title)
        fig.tight_layout()
        specs = figure_image(fig)
    if draw_diffspec:
        d = np.abs(10 ** (spec_db(a) / 20.0) - 10 ** (spec_db(b) / 20.0))
        fig = plt.figure(figsize=(10, 3))
        plt.imshow(20 * np.log10(d + 1e-9), origin="lower", aspect="auto")
        plt.title("|Spec(A) − Spec(B)| (dB)")
        plt.tight_layout()
        diff = figure_image(fig)
    none = blank_image(1, 1)
    return (waves if waves is not None else none, specs if specs is not None else none, diff if diff is not None else none)


def _plot_inputs(xr, xp, xn):
    n = min(xr.shape[1], xp.shape[1], xn.shape[1])
    return device_ops.mono_mean(xr, n), device_ops.mono_mean(xp, n), device_ops.mono_mean(xn, n)


# ------------------------------------------------------------------------------------------------ nodes
class Audio_Align_XCorr:
    CATEGORY = "Egregora/Analysis"
    RETURN_TYPES = ("AUDIO", "FLOAT", "FLOAT", "FLOAT", "IMAGE")
    RETURN_NAMES = ("audio_proc_aligned", "delay_samples", "delay_ms", "peak_corr", "debug_image")
    FUNCTION = "execute"

    @classmethod
    def INPUT_TYPES(cls):
        return {"required": {"audio_ref": _AUDIO, "audio_proc": _AUDIO},
                "optional": {"max_shift_ms": ("INT", _opt(200, 0, 5000, 1)), "align_method": (["gcc-phat"], {}),
                             "fractional": _FLAG(True), "fir_len": ("INT", _opt(64, 16, 256, 1))}}

    def execute(self, audio_ref, audio_proc, max_shift_ms=200, align_method="gcc-phat", fractional=True, fir_len=64):
        native.require_device()
        (ref, xr), (proc, xp) = _device_audio(audio_ref), _device_audio(audio_proc)
        sr = ref["sample_rate"]
        aligned, lag = align_stage(xr, _at_rate(xp, proc["sample_rate"], sr), sr, max_shift_ms, fractional, fir_len)
        out = eval_package(sr, aligned.cpu().numpy(), proc.get("meta", {}))
        return (out, lag, 1000.0 * lag / sr, 0.0, blank_image())


class Audio_Gain_Match:
    CATEGORY = "Egregora/Analysis"
    RETURN_TYPES = ("AUDIO", "FLOAT", "FLOAT", "FLOAT")
    RETURN_NAMES = ("audio_matched", "gain_db", "ref_level", "in_level")
    FUNCTION = "execute"

    @classmethod
    def INPUT_TYPES(cls):
        return {"required": {"audio_ref": _AUDIO, "audio_in": _AUDIO},
                "optional": {"mode": (["LUFS-I", "RMS"], {}), "max_gain_db": ("FLOAT", _opt(12.0, -48.0, 48.0, 0.1))}}

    def execute(self, audio_ref, audio_in, mode="LUFS-I", max_gain_db=12.0):
        native.require_device()
        (ref, xr), (inn, xi) = _device_audio(audio_ref), _device_audio(audio_in)
        sr = ref["sample_rate"]
        y, gain_db, ref_level, in_level = gain_stage(xr, _at_rate(xi, inn["sample_rate"], sr), sr, mode, max_gain_db)
        return (eval_package(sr, y.cpu().numpy(), inn.get("meta", {})), gain_db, ref_level, in_level)


class Audio_Null_Test:
    CATEGORY = "Egregora/Analysis"
    RETURN_TYPES = ("AUDIO", "DICT")
    RETURN_NAMES = ("audio_null", "metrics")
    FUNCTION = "execute"

    @classmethod
    def INPUT_TYPES(cls):
        return {"required": {"audio_ref": _AUDIO, "audio_proc_aligned_matched": _AUDIO},
                "optional": {"invert_b": _FLAG(True), "least_squares_scale": _FLAG(False), **_METRIC_FLAGS, **_STFT,
                             "hf_band_hz": ("INT", _opt(8000, 1000, 20000, 100))}}

    def execute(self, audio_ref, audio_proc_aligned_matched, invert_b=True, least_squares_scale=False,
                compute_corr=True, compute_null_rms=True, compute_null_lufs=True,
                compute_lsd=True, compute_hf_residual=False, n_fft=2048, hop=512, hf_band_hz=8000):
        native.require_device()
        (ref, A), (pro, B) = _device_audio(audio_ref), _device_audio(audio_proc_aligned_matched)
        if pro["sample_rate"] != ref["sample_rate"]:
            raise ValueError("Sample rate mismatch after alignment stage")
        null, metrics = null_stage(A, B, ref["sample_rate"], invert_b, least_squares_scale, compute_corr, compute_null_rms,
                                   compute_null_lufs, compute_lsd, compute_hf_residual, n_fft, hop, hf_band_hz)
        return eval_package(ref["sample_rate"], null.cpu().numpy(), {}), metrics


class Audio_Plotter:
    CATEGORY = "Egregora/Visualization"
    RETURN_TYPES = ("IMAGE", "IMAGE", "IMAGE")
    RETURN_NAMES = ("image_waveforms", "image_spectrograms", "image_diffspec")
    FUNCTION = "execute"

    @classmethod
    def INPUT_TYPES(cls):
        return {"required": {"audio_ref": _AUDIO, "audio_proc": _AUDIO, "audio_null": _AUDIO}, "optional": {**_DRAW_FLAGS, **_STFT}}

    def execute(self, audio_ref, audio_proc, audio_null, draw_waveforms=True, draw_spectrograms=True, draw_diffspec=True, n_fft=2048, hop=512):
        native.require_device()
        a, b, null = _plot_inputs(_device_audio(audio_ref)[1], _device_audio(audio_proc)[1], _device_audio(audio_null)[1])
        return plot_stage(a, b, null, draw_waveforms, draw_spectrograms, draw_diffspec, n_fft, hop)


class Null_Test_Full:
    CATEGORY = "Egregora/Analysis"
    RETURN_TYPES = ("AUDIO", "AUDIO", "FLOAT", "FLOAT", "DICT", "IMAGE", "IMAGE", "IMAGE")
    RETURN_NAMES = ("audio_proc_aligned_matched", "audio_null", "delay_ms", "gain_db", "metrics", "image_waveforms",
                    "image_spectrograms", "image_diffspec")
    FUNCTION = "execute"

    @classmethod
    def INPUT_TYPES(cls):
        return {"required": {"audio_ref": _AUDIO, "audio_proc": _AUDIO},
                "optional": {"align_max_shift_ms": ("INT", _opt(200, 0, 5000, 1)), "align_method": (["gcc-phat"], {}),
                             "fractional": _FLAG(True), "fir_len": ("INT", _opt(64, 16, 256, 1)), "match_mode": (["LUFS-I", "RMS"], {}),
                             "least_squares_scale": _FLAG(False), **_METRIC_FLAGS, **_DRAW_FLAGS, **_STFT}}

    def execute(self, audio_ref, audio_proc, align_max_shift_ms=200, align_method="gcc-phat", fractional=True,
                fir_len=64, match_mode="LUFS-I", least_squares_scale=False,
                compute_corr=True, compute_null_rms=True, compute_null_lufs=True,
                compute_lsd=True, compute_hf_residual=False,
                draw_waveforms=True, draw_spectrograms=True, draw_diffspec=True,
                n_fft=2048, hop=512):
        native.require_device()
        (ref, xr), (proc, xp) = _device_audio(audio_ref), _device_audio(audio_proc)
        sr = ref["sample_rate"]
        aligned, lag = align_stage(xr, _at_rate(xp, proc["sample_rate"], sr), sr, align_max_shift_ms, fractional, fir_len)
        matched, gain_db, _, _ = gain_stage(xr, aligned, sr, match_mode)
        null, metrics = null_stage(xr, matched, sr, True, least_squares_scale, compute_corr, compute_null_rms, compute_null_lufs,
                                   compute_lsd, compute_hf_residual, n_fft, hop)
        images = plot_stage(*_plot_inputs(xr, matched, null), draw_waveforms, draw_spectrograms, draw_diffspec, n_fft, hop)
        return (eval_package(sr, matched.cpu().numpy(), proc.get("meta", {})), eval_package(sr, null.cpu().numpy(), {}),
                float(1000.0 * lag / sr), float(gain_db), metrics, *images)


NODE_CLASS_MAPPINGS = {"Audio Align (XCorr)": Audio_Align_XCorr, "Audio Gain Match": Audio_Gain_Match, "Audio Null Test": Audio_Null_Test,
                       "Audio Plotter": Audio_Plotter, "Null Test (Full)": Null_Test_Full}
NODE_DISPLAY_NAME_MAPPINGS = {k: k for k in NODE_CLASS_MAPPINGS}
