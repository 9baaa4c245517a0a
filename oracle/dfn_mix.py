"""ORACLE (test infrastructure, not product): numpy restatement of the stage AROUND the DeepFilterNet model in the
reference's `Egregora_DeepFilterNet_Denoise.execute` (egregora_audio_enhance_extras.py:548-704): RMS VAD in 10 ms frames,
one-pole smoothing, adaptive strength, wet/dry gains, clip, post-gain, ceiling limiter.

Parity status: PINNED against fixture G11 (tests/golden/g11_dfn.*), captured from the reference run with a documented
stand-in for the absent upstream model (tests/golden/make_golden_dfn.py::fake_wet).  The model itself is out of scope.
Float32 array arithmetic with weak Python scalars, as numpy >= 2 evaluates the reference's expressions.
"""
import math

import numpy as np


def vad_probs_rms_48k(x48):
    """:548-558 -- per 480-sample frame sqrt(mean(x^2)), normalised by its 95th percentile, clipped to [0, 1]."""
    hop = 480
    n = (len(x48) + hop - 1) // hop
    rms = np.asarray([float(np.sqrt(np.mean(x48[i * hop:(i + 1) * hop] ** 2))) for i in range(n)], dtype=np.float32)
    p95 = float(np.percentile(rms, 95)) or 1e-6
    return np.clip(rms / p95, 0.0, 1.0).astype(np.float32)


def smooth_probs(probs, smooth_ms):
    """:560-573 -- y[i] = alpha*y[i-1] + (1-alpha)*p[i], alpha = exp(-10 ms / tau), started at p[0], float32."""
    if probs is None or probs.size == 0 or smooth_ms <= 0:
        return probs
    alpha = math.exp(-10.0 / max(1e-3, float(smooth_ms)))
    y = np.empty_like(probs)
    acc = probs[0]
    for i, p in enumerate(probs):
        acc = alpha * acc + (1.0 - alpha) * p
        y[i] = acc
    return y


def strength_per_frame(s0, v, mode, a, thr):
    """:575-595."""
    s0, a = float(s0), float(a)
    v = np.clip(v, 0.0, 1.0)
    if mode == "more_on_noise":
        s = s0 + a * (1.0 - v) * (1.0 - s0)
    elif mode == "more_on_speech":
        s = s0 + a * v * (1.0 - s0)
    elif mode == "gate_on_noise":
        s = ((s0 + a * (1.0 - s0)) * (v < thr) + (s0 * (1.0 - a)) * (v >= thr)).astype(np.float32)
    else:
        s = np.full_like(v, s0, dtype=np.float32)
    return np.clip(s, 0.0, 1.0).astype(np.float32)


def gains(s, curve):
    """:597-606 -- (g_dry, g_wet)."""
    s = np.clip(s, 0.0, 1.0).astype(np.float32)
    if curve == "equal_power":
        return np.cos(0.5 * math.pi * s, dtype=np.float32), np.sin(0.5 * math.pi * s, dtype=np.float32)
    return (1.0 - s).astype(np.float32), s


def mix_stage(dry, wet, sr, strength=0.65, mix_curve="equal_power", adaptive_mode="more_on_noise", adaptive_amount=0.45,
              vad_threshold=0.90, vad_smooth_ms=60, post_gain_db=0.5, limit_ceiling=True, ceiling=0.98):
    """Steps 5-6 of execute (:655-704) for 48 kHz material with the RMS VAD: dry, wet [C,T] float32 -> [C,T] float32."""
    assert sr == 48000
    hop = int(sr * 0.010)
    out = []
    for ch in range(dry.shape[0]):
        v = smooth_probs(vad_probs_rms_48k(dry[ch]), vad_smooth_ms)
        s_eff = strength_per_frame(strength, v, adaptive_mode, adaptive_amount, vad_threshold)
        s_per = np.repeat(s_eff, max(1, hop))[:dry.shape[1]].astype(np.float32)
        gd, gw = gains(s_per, mix_curve)
        out.append(np.clip(gd * dry[ch] + gw * wet[ch], -1.0, 1.0))
    y = np.stack(out).astype(np.float32)
    if post_gain_db != 0.0:
        y = y * np.float32(float(10.0 ** (post_gain_db / 20.0)))
    if limit_ceiling:
        peak = float(np.max(np.abs(y)))
        if peak > ceiling and peak > 0:
            y = y * np.float32(ceiling / peak)
    return np.clip(y, -1.0, 1.0).astype(np.float32)
