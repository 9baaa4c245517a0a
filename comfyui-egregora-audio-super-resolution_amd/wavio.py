"""Minimal RIFF/WAVE reader for the Fat-Llama nodes' `audio_path` / `audio_url` inputs.

The reference reads these with libsndfile (`sf.read(..., dtype="float32")`, egregora_fat_llama_gpu.py:66,76);
neither libsndfile nor ffmpeg exists on the target box, so integer PCM (8/16/24/32 bit) and IEEE float
(32/64 bit) WAV are decoded here with libsndfile's float conversion (int / 2**(bits-1)).  Other containers
raise a RuntimeError naming the limitation.
"""
import struct
from typing import Tuple

import numpy as np


def read_wav_bytes(buf: bytes) -> Tuple[np.ndarray, int]:
    """Returns (frames-first float32 array [S] or [S,C], sample_rate) like sf.read(always_2d=False)."""
    if len(buf) < 12 or buf[:4] != b"RIFF" or buf[8:12] != b"WAVE":
        raise RuntimeError("only RIFF/WAVE input is supported by this build (no libsndfile/ffmpeg on the box)")
    pos, fmt, data = 12, None, None
    while pos + 8 <= len(buf):
        cid, size = buf[pos:pos + 4], struct.unpack("<I", buf[pos + 4:pos + 8])[0]
        body = buf[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            fmt = body
        elif cid == b"data":
            data = body
        pos += 8 + size + (size & 1)
    if fmt is None or data is None or len(fmt) < 16:
        raise RuntimeError("malformed WAV: missing fmt/data chunk")
    tag, ch, sr, _, align, bits = struct.unpack("<HHIIHH", fmt[:16])
    if tag == 0xFFFE and len(fmt) >= 26:       # WAVE_FORMAT_EXTENSIBLE: real tag is the GUID's first 2 bytes
        tag = struct.unpack("<H", fmt[24:26])[0]
    n = len(data) // align * align
    data = data[:n]
    if tag == 1:
        if bits == 8:
            a = (np.frombuffer(data, np.uint8).astype(np.float32) - 128.0) / 128.0
        elif bits == 16:
            a = np.frombuffer(data, "<i2").astype(np.float32) / 32768.0
        elif bits == 24:
            b = np.frombuffer(data, np.uint8).reshape(-1, 3).astype(np.int32)
            v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
            v = np.where(v >= 1 << 23, v - (1 << 24), v)
            a = v.astype(np.float32) / 8388608.0
        elif bits == 32:
            a = (np.frombuffer(data, "<i4").astype(np.float64) / 2147483648.0).astype(np.float32)
        else:
            raise RuntimeError(f"unsupported PCM width {bits}")
    elif tag == 3:
        a = np.frombuffer(data, "<f4" if bits == 32 else "<f8").astype(np.float32)
    else:
        raise RuntimeError(f"unsupported WAV format tag {tag}")
    if ch > 1:
        a = a.reshape(-1, ch)
    return a, int(sr)


def read_wav(path: str) -> Tuple[np.ndarray, int]:
    with open(path, "rb") as f:
        return read_wav_bytes(f.read())


def write_wav_pcm16(path: str, frames_first: np.ndarray, sr: int) -> None:
    """[S] or [S,C] float -> PCM_16 RIFF/WAVE with libsndfile's default float conversion (rint(x*32767), wrap)."""
    a = np.asarray(frames_first, dtype=np.float32)
    if a.ndim == 1:
        a = a[:, None]
    q = np.rint(a * np.float32(32767.0)).astype(np.int64)
    q = ((q + 32768) % 65536 - 32768).astype("<i2")
    ch = q.shape[1]
    body = q.tobytes()
    hdr = (b"RIFF" + struct.pack("<I", 36 + len(body)) + b"WAVEfmt " +
           struct.pack("<IHHIIHH", 16, 1, ch, int(sr), int(sr) * ch * 2, ch * 2, 16) + b"data" + struct.pack("<I", len(body)))
    with open(path, "wb") as f:
        f.write(hdr + body)
