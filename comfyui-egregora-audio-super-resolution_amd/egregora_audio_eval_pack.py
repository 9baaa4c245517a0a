"""Evaluation-pack nodes that sit on this pack's device kernels: the reference's
`Metrics (LSD + SI-SDR)` and `Resample Audio (HQ)` (egregora_audio_eval_pack.py:432-522) with the same mapping keys,
INPUT_TYPES / RETURN_TYPES / RETURN_NAMES / FUNCTION / CATEGORY and `execute` signatures (fixture G10).

STFT, per-frame LSD, the percentile's order statistics, the SI-SDR sums and both resamplers run in libegregora_amd.so
(csrc/egr_glue.hip); the host keeps only the AUDIO coercion rules of the reference (`to_internal_audio`,
`_normalize_CN`, `make_audio`, :60-103).  No CPU fallback: without the library or a gfx950 device `execute` raises.
The reference's other evaluation nodes (ABX, BS.1770 loudness) are host-side bookkeeping and are not part of this pack.
"""
from typing import Any, Dict

import numpy as np
import torch

from . import device_ops, native, resample


from .audio_glue import eval_audio as to_internal_audio, eval_channels_first as normalize_cn, eval_package as make_audio


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
        ints = dict(n_fft=(2048, 512, 8192, 128), hop=(512, 64, 4096, 64))           # widget: (default, min, max, step)
        opt = {k: ("INT", dict(zip(("default", "min", "max", "step"), v))) for k, v in ints.items()}
        opt.update({k: ("BOOLEAN", {"default": True}) for k in ("compute_lsd", "compute_si_sdr")})
        return {"required": {k: ("AUDIO", {}) for k in ("audio_ref", "audio_proc")}, "optional": opt}

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

    MODES = ("auto", "scipy_polyphase", "torchaudio", "linear")

    @classmethod
    def INPUT_TYPES(cls):
        rng = lambda d, lo, hi, st: dict(zip(("default", "min", "max", "step"), (d, lo, hi, st)))
        return {"required": {"audio": ("AUDIO", {}), "target_sr": ("INT", rng(48000, 4000, 384000, 1))},
                "optional": {"mode": (list(cls.MODES), {}), "kaiser_beta": ("FLOAT", rng(14.769, 5.0, 20.0, 0.1))}}

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
