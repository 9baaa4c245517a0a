"""ComfyUI node "Audio Super Resolution (FlashSR)" on the MI355X-native engine.

Same plugin surface as reference egregora_audio_super_resolution.py:372-388 (mapping key
`EgregoraAudioUpscaler`; widgets audio / lowpass_input / output_sr; run(audio, lowpass_input, output_sr)).

Flow (reference :388-431, re-arranged for the device):
  coerce AUDIO -> [C,T] on the GPU -> (resample to 48 kHz) -> one batched gather of ALL 5.12 s windows
  (zero-padded, hop 4.62 s) -> FlashSR engine on [chunks*C, 245760] rows, sharded over ranks when a
  process group exists -> Hann WOLA gather kernel -> (resample to output_sr) -> AUDIO dict on the CPU.
The reference re-creates its runner (and re-loads three checkpoints) on every call (:393); here the engine
is cached per process.
"""
import torch

from . import audio_glue, device_ops, flashsr_engine, native, resample

FUNCTION = "run"
CATEGORY = "Egregora/Audio"


def upscale_48k(x_ct: torch.Tensor, lowpass_input: bool) -> torch.Tensor:
    """[C,T] float32 CUDA @48 kHz -> [C,T] float32 CUDA @48 kHz (chunked FlashSR + WOLA)."""
    C, total = x_ct.shape
    win, hop = audio_glue.CHUNK_SAMPLES, audio_glue.HOP_SAMPLES
    sp = audio_glue.spans(total, win, hop)
    if not sp:
        return torch.zeros((1, max(1, total)), dtype=torch.float32, device=x_ct.device)   # reference :234-235
    preds = flashsr_engine.infer_spans(x_ct, len(sp), win, hop, bool(lowpass_input))      # [n, C, Lp]
    return device_ops.wola_stitch(preds, total, win, hop)


class EgregoraAudioSuperResolution:
    @classmethod
    def INPUT_TYPES(cls):
        return {
            "required": {
                "audio": ("AUDIO",),
                "lowpass_input": ("BOOLEAN", {"default": False}),
                "output_sr": (["48000", "44100", "96000"], {"default": "48000"}),
            }
        }

    RETURN_TYPES = ("AUDIO",)
    FUNCTION = FUNCTION
    CATEGORY = CATEGORY
    OUTPUT_NODE = False

    def run(self, audio=None, lowpass_input=False, output_sr="48000"):
        x, sr = audio_glue.upscaler_input(audio)
        native.require_device()
        flashsr_engine.ensure_ready()
        x = x.to("cuda", torch.float32).contiguous()
        if sr != audio_glue.REQ_SR:
            x = resample.resample_hq(x, sr, audio_glue.REQ_SR)
            sr = audio_glue.REQ_SR
        y = upscale_48k(x, bool(lowpass_input))
        tgt = int(output_sr)
        if tgt != sr:
            y = resample.resample_hq(y, sr, tgt)
            sr = tgt
        return (audio_glue.package(sr, y),)


NODE_CLASS_MAPPINGS = {"EgregoraAudioUpscaler": EgregoraAudioSuperResolution}
NODE_DISPLAY_NAME_MAPPINGS = {"EgregoraAudioUpscaler": "🎧 Audio Super Resolution (FlashSR)"}
