"""Host-side bookkeeping for the two compute nodes: AUDIO coercion, chunk spans, result packaging.

Only shapes, integers and error text live here; all per-sample arithmetic runs on the device
(csrc/egr_glue.hip, csrc/egr_fatllama.hip).  Behaviour mirrors the reference functions cited in each
docstring so the node surface stays a drop-in.
"""
from typing import Any, Dict, List, Tuple

import numpy as np
import torch

REQ_SR = 48000            # reference egregora_audio_super_resolution.py:255
CHUNK_S = 5.12            # :256
OVERLAP_S = 0.50          # :257
CHUNK_SAMPLES = int(REQ_SR * CHUNK_S)                     # 245760, :258
HOP_SAMPLES = int((CHUNK_S - OVERLAP_S) * REQ_SR)         # 221760, :401


def is_audio_dict(a: Any) -> bool:
    return isinstance(a, dict) and "waveform" in a and "sample_rate" in a


def _dict_to_ct(audio: Dict[str, Any], msg: str) -> Tuple[torch.Tensor, int]:
    """AUDIO dict -> ([C,T] tensor on its current device, sr).  Batch element 0 only (reference quirk Q4)."""
    wf = audio["waveform"]
    sr = int(audio["sample_rate"])
    if not torch.is_tensor(wf):
        wf = torch.as_tensor(np.asarray(wf))
    if wf.dim() == 3:
        wf = wf[0]
    if wf.dim() != 2:
        raise RuntimeError(msg.format(shape=tuple(wf.shape)))
    return wf.detach().float(), sr


def upscaler_input(audio: Any) -> Tuple[torch.Tensor, int]:
    """Mirror of _from_audio_dict (reference egregora_audio_super_resolution.py:125-156)."""
    if is_audio_dict(audio):
        return _dict_to_ct(audio, "Unexpected AUDIO tensor shape {shape}; expected [C, T].")
    if isinstance(audio, (list, tuple)) and len(audio) == 2:
        arr, sr = audio
        a = np.asarray(arr, dtype=np.float32)
        if a.ndim == 1:
            a = a[None, :]
        elif a.ndim == 2:
            frames_first = a.shape[0] >= a.shape[1] and a.shape[1] <= 8
            a = a.T if frames_first else a
        else:
            a = a.reshape(1, -1)
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)), int(sr)
    raise RuntimeError("No valid AUDIO provided.")


def channels_first(x: Any) -> np.ndarray:
    """Mirror of _to_cs (reference egregora_fat_llama_gpu.py:18-32): [S]/[S,C]/[C,S] -> [C,S] float32,
    peak-normalised by (m + 1e-8) only when the peak exceeds 1."""
    a = np.asarray(x, dtype=np.float32)
    if a.ndim == 1:
        a = a[None, :]
    elif a.ndim == 2:
        rows, cols = a.shape
        if cols <= 8 and rows > cols:
            a = a.T
    else:
        a = a.reshape(-1)[None, :]
    peak = float(np.max(np.abs(a))) if a.size else 0.0
    if peak > 1.0:
        a = a / (peak + 1e-8)
    return np.ascontiguousarray(a, dtype=np.float32)


def spans(total: int, win: int = CHUNK_SAMPLES, hop: int = HOP_SAMPLES) -> List[Tuple[int, int]]:
    """(start, length) windows covering [0,total).  Same sequence as _iter_chunks (reference :213-225):
    starts k*hop, length min(win, total-start), stop at the first window that reaches the end."""
    if total <= 0:
        return []
    n = 1 if total <= win else 1 + -(-(total - win) // hop)
    return [(k * hop, min(win, total - k * hop)) for k in range(n)]


def package(sr: int, ct: torch.Tensor) -> Dict[str, Any]:
    """[C,T] (any device) -> AUDIO dict with float32 CPU contiguous [1,C,T].  Mirror of _make_audio (:116-123)."""
    if ct.dim() == 1:
        ct = ct[None, :]
    wf = ct.detach().to("cpu", torch.float32).unsqueeze(0).contiguous()
    return {"waveform": wf, "sample_rate": int(sr)}


# ---------------------------------------------------------------- evaluation-pack AUDIO conventions
# (reference egregora_audio_eval_pack.py:53-103; behaviour pinned by fixture G10)
def eval_channels_first(arr) -> np.ndarray:
    """[.., N] float32 with time last: size-1 axes dropped; a vector becomes one channel; a matrix is transposed when it has
    more rows than columns; with more axes the longest one is time and the others fold into channels."""
    a = np.squeeze(np.asarray(arr))
    if a.ndim <= 1:
        return a.reshape(1, -1).astype(np.float32)
    if a.ndim == 2:
        return (a.T if a.shape[0] > a.shape[1] else a).astype(np.float32)
    t_axis = int(np.argmax(a.shape))
    a = np.moveaxis(a, t_axis, -1)
    return a.reshape(-1, a.shape[-1]).astype(np.float32)


def eval_package(sr: int, samples, meta=None) -> Dict[str, Any]:
    """The eval pack's AUDIO dict: both spellings of the rate, a [C,N] numpy view and a [1,C,N] tensor view, meta copied."""
    cn = eval_channels_first(samples)
    rate = int(sr)
    return dict(sr=rate, sample_rate=rate, samples=cn, waveform=torch.from_numpy(cn)[None], meta=dict(meta or {}))


def eval_audio(x: Any) -> Dict[str, Any]:
    """ComfyUI AUDIO dict (batch element 0) or the eval pack's own dict -> eval_package(...); error text as the reference's."""
    if isinstance(x, dict):
        rate = x.get("sample_rate") or x.get("sr") or x.get("rate")
        if "waveform" in x and rate is not None:
            wf = x["waveform"]
            wf = wf.detach().cpu().numpy() if hasattr(wf, "detach") else np.asarray(wf)
            return eval_package(int(rate), wf[0] if wf.ndim == 3 else wf, x.get("meta", {}))
        rate = x.get("sr") or x.get("sample_rate")
        if "sr" in x or "sample_rate" in x:
            for key in ("samples", "audio", "array"):
                if x.get(key) is not None:
                    buf = x[key]
                    buf = buf.detach().cpu().numpy() if hasattr(buf, "detach") else np.asarray(buf)
                    return eval_package(int(rate), buf, x.get("meta", {}))
            raise ValueError("Audio dict missing samples/waveform")
    raise ValueError("Unsupported AUDIO object for this node")
