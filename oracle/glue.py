"""ORACLE (test infrastructure, not product): numpy restatement of the reference's host-side glue
for the FlashSR hot path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.

Parity status: PINNED against fixtures G1-G6 captured from the reference itself
(tests/golden/make_golden.py; checked by tests/test_oracle_golden.py).

Every function cites the reference lines it follows (paths relative to /root/reference).
"""
from math import gcd

import numpy as np

WIN = 245760          # egregora_audio_super_resolution.py:258  int(48000 * 5.12)
HOP = 221760          # :401  int((5.12 - 0.50) * 48000)
REQ_SR = 48000        # :255


def chunk_spans(total, win=WIN, hop=HOP):
    """(start, length) spans covering [0,total).  egregora_audio_super_resolution.py:213-225."""
    out, i = [], 0
    while i < total:
        length = win if total - i > win else total - i
        out.append((i, length))
        if i + length >= total:
            break
        i += hop
    return out


def hann_sym(n):
    """Symmetric Hann, float32 (np.hanning): 0.5 - 0.5 cos(2 pi k/(n-1)).  :210-211."""
    if n < 1:
        return np.zeros(0, np.float32)
    if n == 1:
        return np.ones(1, np.float32)
    k = np.arange(n, dtype=np.float64)
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * k / (n - 1))).astype(np.float32)


def wola(preds, total, win=WIN):
    """Hann-weighted overlap-add.  preds: list of (y[C,Lp], start, L_in).  :227-251.

    Weights are the PREFIX w_full[:L] of the full-length window (quirk Q3), accumulation and the
    final divide are float32, wsum==0 -> 1 (quirk Q1).
    """
    if not preds:
        return np.zeros((1, max(1, total)), np.float32)
    C = preds[0][0].shape[0]
    acc = np.zeros((C, total), np.float32)
    ws = np.zeros(total, np.float32)
    wf = hann_sym(win)
    for y, start, lin in preds:
        L = min(lin, y.shape[1])
        w = wf[:L] if L <= win else np.ones(L, np.float32)
        acc[:, start:start + L] += y[:, :L] * w[None, :]
        ws[start:start + L] += w
    ws[ws == 0] = 1.0
    return (acc / ws[None, :]).astype(np.float32)


def resample_poly_design(up, down, beta=5.0):
    """FIR that scipy.signal.resample_poly designs by default (window=('kaiser', 5.0)):
    half_len = 10*max(up,down); h = firwin(2*half_len+1, 1/max(up,down), window) * up.
    Restated from scipy 1.15 signal/_signaltools.py::resample_poly (third-party, pinned by meta.json)."""
    from scipy.signal import firwin
    mx = max(up, down)
    half = 10 * mx
    h = firwin(2 * half + 1, 1.0 / mx, window=("kaiser", beta)).astype(np.float64) * up
    return h, half


def resample_poly_ref(x, up, down):
    """Direct (slow, O(n_out*taps/up)) polyphase evaluation identical in definition to
    scipy.signal.resample_poly(x, up, down) for 1-D x with default padtype='constant'."""
    x = np.asarray(x)
    g = gcd(up, down)
    up, down = up // g, down // g
    if up == down == 1:
        return x.copy()
    n_in = x.shape[0]
    n_out = n_in * up // down + bool(n_in * up % down)
    h, half = resample_poly_design(up, down)
    if x.dtype == np.float32:
        h = h.astype(np.float32)
    # scipy pads h in front so that output sample 0 is centred on input sample 0
    xd = x.astype(np.float64)
    hd = h.astype(np.float64)
    out = np.zeros(n_out, np.float64)
    for m in range(n_out):
        # y[m] = sum_k h[m*down - k*up + half] x[k]
        t = m * down + half
        k_hi = min(n_in - 1, t // up)
        k_lo = max(0, -((len(hd) - 1 - t) // up))
        ks = np.arange(k_lo, k_hi + 1)
        out[m] = np.dot(hd[t - ks * up], xd[ks])
    return out.astype(x.dtype)


def resample_hq(x_cs, src, dst):
    """The branch of _resample_hq that actually runs when scipy is present and soxr is not
    (requirements.txt lists scipy, not soxr).  egregora_audio_super_resolution.py:159-187."""
    if src == dst:
        return x_cs.astype(np.float32)
    from scipy.signal import resample_poly
    g = gcd(src, dst)
    up, down = dst // g, src // g
    out = [resample_poly(x_cs[c], up=up, down=down).astype(np.float32) for c in range(x_cs.shape[0])]
    L = min(map(len, out))
    return np.stack([o[:L] for o in out], axis=0)


def coerce_audio(audio):
    """AUDIO dict or (array, sr) -> ([C,S] float32, int sr).  :125-156 (quirk Q4: batch 0 only)."""
    import torch
    if isinstance(audio, dict) and "waveform" in audio and "sample_rate" in audio:
        wf = audio["waveform"]
        sr = int(audio["sample_rate"])
        if wf.dim() == 3:
            wf = wf[0]
        if wf.dim() != 2:
            raise RuntimeError(f"Unexpected AUDIO tensor shape {tuple(wf.shape)}; expected [C, T].")
        return wf.detach().cpu().float().numpy(), sr
    if isinstance(audio, (list, tuple)) and len(audio) == 2:
        arr, sr = audio
        arr = np.asarray(arr, dtype=np.float32)
        if arr.ndim == 1:
            cs = arr[None, :]
        elif arr.ndim == 2:
            cs = arr.T if (arr.shape[0] >= arr.shape[1] and arr.shape[1] <= 8) else arr
        else:
            cs = arr.reshape(1, -1)
        return cs.astype(np.float32), int(sr)
    raise RuntimeError("No valid AUDIO provided.")


def to_cs(x):
    """Fat-Llama side coercion incl. the m>1 peak normalisation.  egregora_fat_llama_gpu.py:18-32."""
    a = np.asarray(x, dtype=np.float32)
    if a.ndim == 1:
        a = a[None, :]
    elif a.ndim == 2:
        h, w = a.shape
        if w <= 8 and h > w:
            a = a.T
    else:
        a = a.reshape(-1)[None, :]
    m = float(np.max(np.abs(a))) if a.size else 0.0
    if m > 1.0:
        a = a / (m + 1e-8)
    return a.astype(np.float32)


def make_audio(sr, cs):
    """[C,T] -> {"waveform": [1,C,T] float32 contiguous, "sample_rate": int}.  :116-123."""
    import torch
    s = np.asarray(cs, dtype=np.float32)
    if s.ndim == 1:
        s = s[None, :]
    return {"waveform": torch.from_numpy(s).unsqueeze(0).contiguous(), "sample_rate": int(sr)}


def flashsr_node_glue(audio_cs, model_fn, win=WIN, hop=HOP):
    """The chunk loop of EgregoraAudioSuperResolution.run (:406-420) around an arbitrary per-chunk
    model callable model_fn(x[C,win]) -> y[C,Lp]."""
    total = audio_cs.shape[1]
    preds = []
    for start, L in chunk_spans(total, win, hop):
        c = audio_cs[:, start:start + L]
        if L < win:
            c = np.concatenate([c, np.zeros((audio_cs.shape[0], win - L), np.float32)], axis=1)
        preds.append((model_fn(c), start, L))
    return wola(preds, total, win)
