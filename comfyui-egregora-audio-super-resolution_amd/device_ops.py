"""Thin torch-tensor wrappers over the glue entry points of the C ABI (csrc/egr_glue.hip)."""
import numpy as np
import torch

from . import native


def _chk(t, what):
    if not (torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise RuntimeError(f"{what}: want a contiguous float32 CUDA tensor")


_WINDOWS = {}


def hann_window(n: int, device) -> torch.Tensor:
    """np.hanning(n) as float32 on `device` (bit-identical to the reference's _hann, :210-211)."""
    key = (n, str(device))
    w = _WINDOWS.get(key)
    if w is None:
        w = torch.from_numpy(np.hanning(n).astype(np.float32)).to(device)
        _WINDOWS[key] = w
    return w


def pcm16_roundtrip(x: torch.Tensor, write_scale: float = 32767.0, read_div: float = 32768.0) -> torch.Tensor:
    _chk(x, "pcm16_roundtrip")
    y = torch.empty_like(x)
    native.check(native.lib().egr_pcm16_roundtrip(native.ptr(x), native.ptr(y), x.numel(), write_scale, read_div,
                                                   native.stream_ptr()), "egr_pcm16_roundtrip")
    return y


def chunk_gather(x_ct: torch.Tensor, win: int, hop: int, chunk_begin: int, n_chunks: int) -> torch.Tensor:
    """[C,T] -> [n_chunks, C, win] zero-padded windows starting at (chunk_begin+k)*hop."""
    _chk(x_ct, "chunk_gather")
    C, T = x_ct.shape
    out = torch.empty((n_chunks, C, win), dtype=torch.float32, device=x_ct.device)
    native.check(native.lib().egr_chunk_gather(native.ptr(x_ct), C, T, win, hop, chunk_begin, n_chunks,
                                                native.ptr(out), native.stream_ptr()), "egr_chunk_gather")
    return out


def wola_stitch(preds: torch.Tensor, total: int, win: int, hop: int) -> torch.Tensor:
    """preds [n_chunks, C, Lp] -> [C,total]: Hann WOLA with division by the weight sum."""
    _chk(preds, "wola_stitch")
    n, C, lp = preds.shape
    out = torch.empty((C, total), dtype=torch.float32, device=preds.device)
    w = hann_window(win, preds.device)
    native.check(native.lib().egr_wola_stitch(native.ptr(preds), n, C, lp, total, win, hop, native.ptr(w),
                                               native.ptr(out), native.stream_ptr()), "egr_wola_stitch")
    return out


def stft_mag(x: torch.Tensor, n_fft: int = 2048, hop: int = 512) -> torch.Tensor:
    """[T] or [C,T] -> [n_fft/2+1, frames] (a transposed view of the frame-major device buffer), matching
    the reference's _stft_mag layout (egregora_audio_eval_pack.py:389-402)."""
    if x.dim() == 1:
        x = x[None, :]
    x = x.contiguous()
    _chk(x, "stft_mag")
    C, T = x.shape
    frames = 1 + max(0, (T - n_fft) // hop)
    out = torch.empty((frames, n_fft // 2 + 1), dtype=torch.float32, device=x.device)
    w = hann_window(n_fft, x.device)
    native.check(native.lib().egr_stft_mag(native.ptr(x), C, T, n_fft, hop, native.ptr(w), native.ptr(out),
                                            native.stream_ptr()), "egr_stft_mag")
    return out.t()
