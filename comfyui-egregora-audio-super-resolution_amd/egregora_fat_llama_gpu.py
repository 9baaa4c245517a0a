"""ComfyUI node "Spectral Enhance (Fat Llama - GPU)" on the MI355X-native engine.

Same plugin surface as the reference node (reference egregora_fat_llama_gpu.py:228-268): mapping key
`EgregoraFatLlamaGPU`, INPUT_TYPES / RETURN_TYPES / FUNCTION / CATEGORY / OUTPUT_NODE and the positional
order of run().  What differs is everything behind it: no CuPy, no temp WAV, no monkey-patching -- the
node hands a device tensor to libegregora_amd.so through fatllama_engine.
"""
from pathlib import Path

import numpy as np
import torch

from . import audio_glue, audio_io, fatllama_engine, native

RETURN_TYPES = ("AUDIO",)
FUNCTION = "run"
CATEGORY = "Egregora/Audio"


def resolve_input(AUDIO=None, audio_path: str = "", audio_url: str = ""):
    """AUDIO dict | (array, sr) | path | url  ->  ([C,T] float tensor, sr).

    Precedence and error text follow _normalize_audio_input (reference egregora_fat_llama_gpu.py:40-80).
    The dict form is passed through un-normalised; the other forms go through the peak>1 rescale of
    _to_cs exactly as the reference does before it writes its temp WAV.
    """
    if audio_glue.is_audio_dict(AUDIO):
        return audio_glue._dict_to_ct(AUDIO, "Unexpected AUDIO tensor shape: {shape} (want [C,T])")
    if isinstance(AUDIO, (list, tuple)) and len(AUDIO) == 2:
        arr, sr = AUDIO
        return torch.from_numpy(audio_glue.channels_first(np.asarray(arr))), int(sr)
    if audio_path:
        p = Path(audio_path)
        if not p.exists():
            raise RuntimeError(f"audio_path not found: {audio_path}")
        y, sr = audio_io.read_audio(str(p))
        return torch.from_numpy(audio_glue.channels_first(y)), int(sr)
    if audio_url:
        import requests
        r = requests.get(audio_url, timeout=60)
        r.raise_for_status()
        y, sr = audio_io.read_audio_bytes(r.content)
        return torch.from_numpy(audio_glue.channels_first(y)), int(sr)
    raise RuntimeError("No AUDIO provided.")


def check_format(target_format):
    # upstream writes wav or flac; both are PCM_16 for float data, so the arithmetic is identical
    if target_format not in ("wav", "flac"):
        raise RuntimeError(f"target_format must be 'wav' or 'flac', got {target_format!r}")


class EgregoraFatLlamaGPU:
    """Iterative FFT-threshold spectral enhancer; runs entirely on the GPU."""

    @classmethod
    def INPUT_TYPES(cls):
        return {
            "required": {
                "target_format": (["wav", "flac"],),
                "max_iterations": ("INT", {"default": 300, "min": 1, "max": 5000}),
                "threshold_value": ("FLOAT", {"default": 0.6, "min": 0.0, "max": 1.0, "step": 0.01}),
                "target_bitrate_kbps": ("INT", {"default": 1411, "min": 64, "max": 5000}),
                "toggle_normalize": ("BOOLEAN", {"default": True}),
                "toggle_autoscale": ("BOOLEAN", {"default": True}),
            },
            "optional": {
                "AUDIO": ("AUDIO",),
                "audio_path": ("STRING", {"default": ""}),
                "audio_url": ("STRING", {"default": ""}),
            },
        }

    RETURN_TYPES = RETURN_TYPES
    FUNCTION = FUNCTION
    CATEGORY = CATEGORY
    OUTPUT_NODE = False

    def run(self, target_format, max_iterations, threshold_value, target_bitrate_kbps, toggle_normalize,
            toggle_autoscale, AUDIO=None, audio_path="", audio_url=""):
        native.require_device()
        check_format(target_format)
        cs, sr = resolve_input(AUDIO, audio_path, audio_url)
        y, out_sr = fatllama_engine.node_run(cs, sr, max_iterations, threshold_value, target_bitrate_kbps,
                                             toggle_normalize, toggle_autoscale)
        return (audio_glue.package(out_sr, y),)


NODE_CLASS_MAPPINGS = {"EgregoraFatLlamaGPU": EgregoraFatLlamaGPU}
NODE_DISPLAY_NAME_MAPPINGS = {"EgregoraFatLlamaGPU": "🎛️ Spectral Enhance (Fat Llama — GPU)"}
