"""ComfyUI custom-node pack: MI355X-native (gfx950) backend for Egregora Audio Super Resolution.

Drop-in for the reference pack's three core nodes (reference __init__.py:33-43): same mapping keys, same
display names, same INPUT_TYPES / RETURN_TYPES / FUNCTION surface -- plus the evaluation-pack nodes and the DeepFilterNet
stage that run on this pack's kernels (merged the way the reference merges its ENHANCE_MAP / EVAL_MAP, __init__.py:9-23,47-53).  Compute runs in libegregora_amd.so
(hand-written HIP, C ABI in include/egregora_amd.h); importing this package needs neither the library
nor a GPU -- the nodes raise at run() time when either is missing.
"""
from .egregora_audio_super_resolution import EgregoraAudioSuperResolution
from .egregora_fat_llama_cpu import EgregoraFatLlamaCPU
from .egregora_fat_llama_gpu import EgregoraFatLlamaGPU
from .egregora_audio_eval_pack import (NODE_CLASS_MAPPINGS as EVAL_MAP, NODE_DISPLAY_NAME_MAPPINGS as EVAL_NAMES)
from .egregora_audio_enhance_extras import (NODE_CLASS_MAPPINGS as ENHANCE_MAP, NODE_DISPLAY_NAME_MAPPINGS as ENHANCE_NAMES)
from .egregora_null_test_suite import (NODE_CLASS_MAPPINGS as NULL_MAP, NODE_DISPLAY_NAME_MAPPINGS as NULL_NAMES)

NODE_CLASS_MAPPINGS = {
    "EgregoraAudioUpscaler": EgregoraAudioSuperResolution,
    "EgregoraFatLlamaGPU": EgregoraFatLlamaGPU,
    "EgregoraFatLlamaCPU": EgregoraFatLlamaCPU,
}

NODE_DISPLAY_NAME_MAPPINGS = {
    "EgregoraAudioUpscaler": "🎧 Audio Super Resolution (FlashSR)",
    "EgregoraFatLlamaGPU": "🎛️ Spectral Enhance (Fat Llama — GPU)",
    "EgregoraFatLlamaCPU": "🎛️ Spectral Enhance (Fat Llama — CPU/FFTW)",
}

NODE_CLASS_MAPPINGS.update(ENHANCE_MAP)
NODE_CLASS_MAPPINGS.update(EVAL_MAP)
NODE_CLASS_MAPPINGS.update(NULL_MAP)
NODE_DISPLAY_NAME_MAPPINGS.update(ENHANCE_NAMES)
NODE_DISPLAY_NAME_MAPPINGS.update(EVAL_NAMES)
NODE_DISPLAY_NAME_MAPPINGS.update(NULL_NAMES)

__all__ = ["NODE_CLASS_MAPPINGS", "NODE_DISPLAY_NAME_MAPPINGS"]
