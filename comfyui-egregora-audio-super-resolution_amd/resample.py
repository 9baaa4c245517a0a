"""Sample-rate conversion for the FlashSR node on the device.

Mirrors the branch of the reference's `_resample_hq` that runs when scipy is present and soxr is not
(egregora_audio_super_resolution.py:178-187; requirements.txt lists scipy, not soxr): per channel
`scipy.signal.resample_poly(x, up=dst/g, down=src/g)` with the default Kaiser(5.0) design, float32 data.
The FIR is designed on the host (numpy restatement of scipy.signal.firwin, a few thousand taps at most) and the
polyphase evaluation runs in csrc/egr_glue.hip::k_resample_poly with scipy's accumulation order.
"""
from math import gcd

import numpy as np
import torch

from . import native

_FILTERS = {}


def design_filter(up: int, down: int, beta: float = 5.0):
    """float32 taps h (already multiplied by `up`) and half length, as scipy.signal.resample_poly builds them:
    firwin(2*half+1, 1/max(up,down), window=('kaiser', beta)) with half = 10*max(up,down)."""
    mx = max(up, down)
    half = 10 * mx
    numtaps = 2 * half + 1
    cutoff = 1.0 / mx
    m = np.arange(numtaps, dtype=np.float64) - 0.5 * (numtaps - 1)
    h = cutoff * np.sinc(cutoff * m)            # ideal low-pass (fs = 2): right band edge only
    h *= np.kaiser(numtaps, beta)
    h /= np.sum(h)                              # unit gain at DC (scale=True)
    h32 = h.astype(np.float32)
    h32 *= np.float32(up)                       # scipy: h *= up after the cast to x.dtype
    return h32, half


def rates(src_sr: int, dst_sr: int):
    g = gcd(int(src_sr), int(dst_sr))
    return int(dst_sr) // g, int(src_sr) // g


def resample_hq(x_ct: torch.Tensor, src_sr: int, dst_sr: int) -> torch.Tensor:
    """[C,T] float32 CUDA -> [C, ceil(T*up/down)] float32 CUDA."""
    if int(src_sr) == int(dst_sr):
        return x_ct.to(torch.float32)
    if not (x_ct.is_cuda and x_ct.dim() == 2):
        raise RuntimeError("resample_hq wants a [C,T] tensor on the GPU")
    x = x_ct.to(torch.float32).contiguous()
    up, down = rates(src_sr, dst_sr)
    key = (up, down, str(x.device))
    if key not in _FILTERS:
        h, half = design_filter(up, down)
        _FILTERS[key] = (torch.from_numpy(h).to(x.device), half)
    h, half = _FILTERS[key]
    C, n_in = x.shape
    n_out = (n_in * up + down - 1) // down
    y = torch.empty((C, n_out), dtype=torch.float32, device=x.device)
    native.check(native.lib().egr_resample_poly(native.ptr(x), C, n_in, up, down, native.ptr(h), half, native.ptr(y),
                                                n_out, native.stream_ptr()), "egr_resample_poly")
    return y
