"""Evaluation-pack nodes that sit on this pack's device kernels: the reference's
`Metrics (LSD + SI-SDR)` and `Resample Audio (HQ)` (egregora_audio_eval_pack.py:432-522) with the same mapping keys,
INPUT_TYPES / RETURN_TYPES / RETURN_NAMES / FUNCTION / CATEGORY and `execute` signatures (fixture G10).

STFT, per-frame LSD, the percentile's order statistics, the SI-SDR sums and both resamplers run in libegregora_amd.so
(csrc/egr_glue.hip); the host keeps only the AUDIO coercion rules of the reference (`to_internal_audio`,
`_normalize_CN`, `make_audio`, :60-103).  No CPU fallback: without the library or a gfx950 device `execute` raises.
The reference's other evaluation nodes (ABX, BS.1770 loudness) are host-side bookkeeping and are not part of this pack.
"""
from typing import Any, Dict, Optional

import numpy as np
import torch

from . import device_ops, native, resample


def _to_numpy(x: Any) -> np.ndarray:
    if isinstance(x, np.ndarray):
        return x
    if hasattr(x, "detach") and hasattr(x, "cpu"):
        return x.detach().cpu().numpy()
    return np.asarray(x)


def normalize_cn(arr) -> np.ndarray:
    """Reference _normalize_CN (:60-74): squeeze; 1-D -> [1,N]; 2-D transposed when rows > columns; more dimensions: the
    longest axis becomes time and the rest folds into channels.  float32."""
    a = np.squeeze(np.asarray(arr))
    if a.ndim == 1:
        a = a[None, :]
    elif a.ndim == 2:
        if a.shape[0] > a.shape[1]:
            a = a.T
    else:
        a = np.moveaxis(a, int(np.argmax(a.shape)), -1)
        a = a.reshape(int(np.prod(a.shape[:-1])), a.shape[-1])
    return a.astype(np.float32)


def make_audio(sr: int, samples_cn, meta: Optional[dict] = None) -> Dict[str, Any]:
    """Reference make_audio (:76-86): the eval pack's AUDIO dict carries both spellings of the rate and both views."""
    s = normalize_cn(samples_cn)
    return {"sr": int(sr), "sample_rate": int(sr), "samples": s, "waveform": torch.from_numpy(s).unsqueeze(0),
            "meta": dict(meta or {})}


def to_internal_audio(x: Any) -> Dict[str, Any]:
    """Reference to_internal_audio (:89-103), including its error text."""
    if isinstance(x, dict) and "waveform" in x and ("sample_rate" in x or "sr" in x or "rate" in x):
        sr = int(x.get("sample_rate") or x.get("sr") or x.get("rate"))
        wf = _to_numpy(x["waveform"])
        if wf.ndim == 3:
            wf = wf[0]
        return make_audio(sr, wf, x.get("meta", {}))
    if isinstance(x, dict) and ("sr" in x or "sample_rate" in x):
        sr = int(x.get("sr") or x.get("sample_rate"))
        buf = x.get("samples")
        if buf is None:
            buf = x.get("audio")
        if buf is None:
            buf = x.get("array")
        if buf is None:
            raise ValueError("Audio dict missing samples/waveform")
        return make_audio(sr, _to_numpy(buf), x.get("meta", {}))
    raise ValueError("Unsupported AUDIO object for this node")


def _device_cn(a: Dict[str, Any]) -> torch.Tensor:
    native.require_device()
    return torch.from_numpy(np.ascontiguousarray(a["samples"])).cuda()


class Metrics_LSD_SISDR:
    CATEGORY = "Egregora/Analysis"
    RETURN_TYPES = ("DICT",)
    RETURN_NAMES = ("metrics",)
    FUNCTION = "execute"

    @classmethod
    def INPUT_TYPES(cls):
        return {
            "required": {
                "audio_ref": ("AUDIO", {}),
                "audio_proc": ("AUDIO", {}),
            },
            "optional": {
                "n_fft": ("INT", {"default": 2048, "min": 512, "max": 8192, "step": 128}),
                "hop": ("INT", {"default": 512, "min": 64, "max": 4096, "step": 64}),
                "compute_lsd": ("BOOLEAN", {"default": True}),
                "compute_si_sdr": ("BOOLEAN", {"default": True}),
            },
        }

    def execute(self, audio_ref, audio_proc, n_fft=2048, hop=512, compute_lsd=True, compute_si_sdr=True):
        A = _device_cn(to_internal_audio(audio_ref))
        B = _device_cn(to_internal_audio(audio_proc))
        n = min(A.shape[1], B.shape[1])            # both are mono-downmixed and cut to the common length (:455-459)
        a, b = A[:, :n].contiguous(), B[:, :n].contiguous()
        out: Dict[str, Any] = {}
        if compute_lsd:
            lsd_mean, lsd_p95 = device_ops.lsd(a, b, int(n_fft), int(hop))
            out["lsd_mean_db"] = float(lsd_mean)
            out["lsd_p95_db"] = float(lsd_p95)
        if compute_si_sdr:
            out["si_sdr_db"] = float(device_ops.si_sdr(a, b))
        return (out,)


class Resample_Audio_HQ:
    CATEGORY = "Egregora/Utils"
    RETURN_TYPES = ("AUDIO",)
    RETURN_NAMES = ("audio_out",)
    FUNCTION = "execute"

    @classmethod
    def INPUT_TYPES(cls):
        modes = ["auto", "scipy_polyphase", "torchaudio", "linear"]
        return {
            "required": {
                "audio": ("AUDIO", {}),
                "target_sr": ("INT", {"default": 48000, "min": 4000, "max": 384000, "step": 1}),
            },
            "optional": {
                "mode": (modes, {}),
                "kaiser_beta": ("FLOAT", {"default": 14.769, "min": 5.0, "max": 20.0, "step": 0.1}),
            },
        }

    def execute(self, audio, target_sr=48000, mode="auto", kaiser_beta=14.769):
        a = to_internal_audio(audio)
        src_sr = int(a["sample_rate"])
        if src_sr == int(target_sr):
            return (a,)
        x = _device_cn(a)
        if mode in ("auto", "scipy_polyphase"):
            # scipy.signal.resample_poly(x[c], up, down) per channel on the device (egr_resample_poly); the reference's
            # kaiser_beta widget only feeds its torchaudio branch (:509-512)
            y = resample.resample_hq(x, src_sr, int(target_sr))
        else:
            # "linear", and "torchaudio" where torchaudio is absent (this pack does not depend on it): the reference's
            # np.interp fallback (:514-519)
            new_n = int(round(x.shape[1] * (int(target_sr) / src_sr)))
            y = device_ops.resample_linear(x, new_n)
        return (make_audio(int(target_sr), y.cpu().numpy(), a.get("meta", {})),)


NODE_CLASS_MAPPINGS = {
    "Metrics (LSD + SI-SDR)": Metrics_LSD_SISDR,
    "Resample Audio (HQ)": Resample_Audio_HQ,
}

NODE_DISPLAY_NAME_MAPPINGS = {
    "Metrics (LSD + SI-SDR)": "Egregora Metrics (LSD + SI-SDR)",
    "Resample Audio (HQ)": "Egregora Resample Audio (HQ)",
}
