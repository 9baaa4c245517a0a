"""`Audio Align (XCorr)` of the reference's null-test suite (egregora_null_test_suite.py:272-340, fixture G13) on this pack's
kernels: the GCC-PHAT delay estimate runs on the Fat-Llama whole-signal transform passes (egr_gcc_phat), the compensation
(integer shift + windowed-sinc fractional FIR + pad / crop) in egr_shift_fir, the rate match of the processed signal in
egr_resample_linear.  Same mapping key, INPUT_TYPES / RETURN_TYPES / RETURN_NAMES / FUNCTION / CATEGORY and `execute` signature.

Reference behaviours kept on purpose (the fixture pins them): the delay is the true lag minus one (Q8, see
device_ops.xcorr_delay); the fractional FIR always DELAYS by frac = |d| - floor(|d|) whatever the sign of d, and an even tap
count adds the half sample of np.convolve(..., "same") (Q9).  Not reproduced: the debug IMAGE is the reference's own fallback
(a blank 8x8 image) instead of a matplotlib plot.  The suite's other nodes (gain match, null test, plotter) are not built.
"""
import math

import numpy as np
import torch

from . import device_ops, native
from .audio_glue import eval_audio, eval_package


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


class Audio_Align_XCorr:
    CATEGORY = "Egregora/Analysis"
    RETURN_TYPES = ("AUDIO", "FLOAT", "FLOAT", "FLOAT", "IMAGE")
    RETURN_NAMES = ("audio_proc_aligned", "delay_samples", "delay_ms", "peak_corr", "debug_image")
    FUNCTION = "execute"

    @classmethod
    def INPUT_TYPES(cls):
        rng = lambda d, lo, hi, st: dict(zip(("default", "min", "max", "step"), (d, lo, hi, st)))
        return {"required": {"audio_ref": ("AUDIO", {}), "audio_proc": ("AUDIO", {})},
                "optional": {"max_shift_ms": ("INT", rng(200, 0, 5000, 1)), "align_method": (["gcc-phat"], {}),
                             "fractional": ("BOOLEAN", {"default": True}), "fir_len": ("INT", rng(64, 16, 256, 1))}}

    def execute(self, audio_ref, audio_proc, max_shift_ms=200, align_method="gcc-phat", fractional=True, fir_len=64):
        native.require_device()
        ref, proc = eval_audio(audio_ref), eval_audio(audio_proc)
        sr = ref["sample_rate"]
        xr = torch.from_numpy(np.ascontiguousarray(ref["samples"])).cuda()
        xp = torch.from_numpy(np.ascontiguousarray(proc["samples"])).cuda()
        if proc["sample_rate"] != sr:                       # linear interpolation is what the reference uses here (:299-308)
            xp = device_ops.resample_linear(xp, int(round(xp.shape[1] * sr / proc["sample_rate"])))
        n = min(xr.shape[1], xp.shape[1])
        a = xr[:, :n].mean(dim=0).contiguous()
        b = xp[:, :n].mean(dim=0).contiguous()
        lag = device_ops.xcorr_delay(a, b, sr, int(sr * (max_shift_ms / 1000.0)))
        delay_samples = float(lag)
        comp = -delay_samples if fractional else -round(delay_samples)
        aligned = apply_delay(xp.contiguous(), float(comp), fir_len, xr.shape[1])
        out = eval_package(sr, aligned.cpu().numpy(), proc.get("meta", {}))
        return (out, delay_samples, 1000.0 * delay_samples / sr, 0.0, torch.zeros((1, 8, 8, 3), dtype=torch.float32))


NODE_CLASS_MAPPINGS = {"Audio Align (XCorr)": Audio_Align_XCorr}
NODE_DISPLAY_NAME_MAPPINGS = {"Audio Align (XCorr)": "Audio Align (XCorr)"}
