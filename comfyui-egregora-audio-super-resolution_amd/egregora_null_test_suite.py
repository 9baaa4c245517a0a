"""`Audio Null Test` of the reference's null-test suite (egregora_null_test_suite.py:391-470, fixture G14) on this pack's kernels --
SURVEY.md section 8(f) row 3 admits the null-test METRICS and the GCC-PHAT delay estimator (device_ops.xcorr_delay, fixture G12);
the suite's aligner, gain matcher, plotter and one-shot node are analysis utilities outside the hot-path scope and are not part
of this pack (use the reference's file for them).

Where the work runs: band energy on the Fat-Llama whole-signal transform passes (egr_band_filter); K-weighting, block energies,
correlation / least-squares sums, the null mix, STFT magnitudes and the log-spectral distance in egr_glue.hip kernels.  The host
keeps what is O(frames) or O(1): the loudness gate over the 400 ms blocks, dB conversions, the metric dictionary.
"""
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


def blank_image(h=8, w=8):
    return torch.zeros((1, h, w, 3), dtype=torch.float32)


def _device_audio(x):
    """AUDIO -> (package, [C,N] float32 CUDA tensor)."""
    pkg = eval_audio(x)
    return pkg, torch.from_numpy(np.ascontiguousarray(pkg["samples"])).cuda()


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


# ------------------------------------------------------------------------------------------------ nodes
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


NODE_CLASS_MAPPINGS = {"Audio Null Test": Audio_Null_Test}
NODE_DISPLAY_NAME_MAPPINGS = {k: k for k in NODE_CLASS_MAPPINGS}
