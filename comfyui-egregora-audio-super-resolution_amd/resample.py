"""Sample-rate conversion for the FlashSR node (reference _resample_hq, egregora_audio_super_resolution.py:159-207).

Not built yet on the device: the polyphase-FIR kernel matching scipy.signal.resample_poly (fixture G4) is
the next row of the scope table.  Until then a rate change raises instead of
silently running a CPU resampler.
"""
import torch


def resample_hq(x_ct: torch.Tensor, src_sr: int, dst_sr: int) -> torch.Tensor:
    if int(src_sr) == int(dst_sr):
        return x_ct.to(torch.float32)
    raise RuntimeError(f"on-device resampling {src_sr} -> {dst_sr} Hz is not built yet; feed 48 kHz audio and "
                       "keep output_sr=48000")
