"""ComfyUI node "Spectral Enhance (Fat Llama - CPU/FFTW)" -- surface kept for drop-in compatibility.

Surface mirrors reference egregora_fat_llama_cpu.py:136-172 (mapping key `EgregoraFatLlamaCPU`, 4 required
widgets, default 800 iterations, 7-kwarg upstream call => upstream defaults for normalise/autoscale).
In this MI355X-native pack the node executes on the same device engine as the GPU node: there is no
host-CPU compute path in the product (the CPU restatement lives in oracle/ and is only a test/bench
baseline).  Without a GPU the node raises, like every compute node of this pack.
"""
from . import audio_glue, fatllama_engine, native
from .egregora_fat_llama_gpu import check_format, resolve_input

RETURN_TYPES = ("AUDIO",)
FUNCTION = "run"
CATEGORY = "Egregora/Audio"

# upstream defaults that the reference's 7-kwarg call leaves in force
UPSTREAM_DEFAULT_NORMALIZE = True
UPSTREAM_DEFAULT_AUTOSCALE = True


class EgregoraFatLlamaCPU:
    @classmethod
    def INPUT_TYPES(cls):
        return {
            "required": {
                "target_format": (["wav", "flac"],),
                "max_iterations": ("INT", {"default": 800, "min": 1, "max": 10000}),
                "threshold_value": ("FLOAT", {"default": 0.6, "min": 0.0, "max": 1.0, "step": 0.01}),
                "target_bitrate_kbps": ("INT", {"default": 1411, "min": 64, "max": 5000}),
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

    def run(self, target_format, max_iterations, threshold_value, target_bitrate_kbps, AUDIO=None, audio_path="",
            audio_url=""):
        native.require_device()
        check_format(target_format)
        cs, sr = resolve_input(AUDIO, audio_path, audio_url)
        y, out_sr = fatllama_engine.node_run(cs, sr, max_iterations, threshold_value, target_bitrate_kbps,
                                             UPSTREAM_DEFAULT_NORMALIZE, UPSTREAM_DEFAULT_AUTOSCALE)
        return (audio_glue.package(out_sr, y),)


NODE_CLASS_MAPPINGS = {"EgregoraFatLlamaCPU": EgregoraFatLlamaCPU}
NODE_DISPLAY_NAME_MAPPINGS = {"EgregoraFatLlamaCPU": "🎛️ Spectral Enhance (Fat Llama — CPU/FFTW)"}
