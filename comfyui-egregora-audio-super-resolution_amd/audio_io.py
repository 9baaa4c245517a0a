"""File / URL inputs of the Fat-Llama nodes (reference egregora_fat_llama_gpu.py:61-78: `sf.read(path, dtype="float32",
always_2d=False)`): the container is sniffed from its first bytes -- RIFF/WAVE (wavio.py) and FLAC (flacio.py) are decoded by
this pack's own parsers with libsndfile's float conversion; anything else goes to `soundfile` when that is importable and is
refused otherwise (there is no libsndfile / ffmpeg on the target box)."""
from typing import Tuple

import numpy as np

from . import flacio, wavio


def read_audio_bytes(buf: bytes) -> Tuple[np.ndarray, int]:
    """(frames-first float32 [S] or [S,C], sample_rate)."""
    if buf[:4] == b"RIFF" and buf[8:12] == b"WAVE":
        return wavio.read_wav_bytes(buf)
    if buf[:4] == b"fLaC" or (buf[:3] == b"ID3" and b"fLaC" in buf[:1 << 20]):
        return flacio.read_flac_bytes(buf)
    try:
        import io
        import soundfile as sf
    except Exception:       # noqa: BLE001
        raise RuntimeError("unsupported audio container: this pack decodes RIFF/WAVE and FLAC itself; other formats need the "
                           "`soundfile` package (libsndfile), which is not installed") from None
    y, sr = sf.read(io.BytesIO(buf), dtype="float32", always_2d=False)
    return y, int(sr)


def read_audio(path: str) -> Tuple[np.ndarray, int]:
    with open(path, "rb") as f:
        return read_audio_bytes(f.read())
