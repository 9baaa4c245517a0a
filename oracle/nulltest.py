"""ORACLE (test infrastructure, not product): numpy restatement of the level meters and the null metrics of the reference's
null-test suite.

Parity status: PINNED against fixture G14 (tests/golden/g14_nulltest.*), captured from the reference's nodes; every float of the
fixture is reproduced exactly (tests/test_nulltest_nodes.py).  Follows egregora_null_test_suite.py:119-165 (levels), :192-199
(band energy), :362-388 (gain match), :420-470 (null test).  Arithmetic notes: the reference's scalar K-weighting loop mixes
Python floats with np.float32 samples, which under numpy >= 2 (NEP 50, the version the fixture was made with) rounds every
product and sum to float32.
"""
import math

import numpy as np

from . import metrics as om


def rms_db(x):
    x = np.asarray(x).astype(np.float64)
    return 10.0 * math.log10(float(np.mean(x * x) + 1e-20))


def k_weight(sr, x):
    """[C,N] float32 -> [C,N] float32: one-pole high-pass at 60 Hz (state in float32) and a 2 % first-difference tilt.  :125-140."""
    k = math.exp(-2 * math.pi * (60.0 / (sr * 0.5)))
    a1, kf = np.float32(1 - k), np.float32(k)
    y = np.zeros(x.shape, np.float32)
    for c in range(x.shape[0]):
        z = np.float32(0.0)
        xs = x[c].astype(np.float32)
        for n in range(xs.shape[0]):
            z = np.float32(a1 * xs[n]) + np.float32(kf * z)
            y[c, n] = xs[n] - z
    y[:, 1:] += np.float32(0.02) * (y[:, 1:] - y[:, :-1])
    return y


def integrated_lufs(sr, x):
    """Gated loudness of [C,N]: 400 ms blocks every 100 ms of the K-weighted mono mean, relative gate at -10 LU.  :143-165."""
    mono = k_weight(sr, x).mean(axis=0)
    blk, hop = max(1, int(round(0.400 * sr))), max(1, int(round(0.100 * sr)))
    frames = 1 + max(0, (mono.shape[0] - blk) // hop)
    ms = np.asarray([float(np.mean(mono[i * hop:i * hop + blk].astype(np.float64) ** 2)) for i in range(frames)]) + 1e-20
    return gate(ms)


def gate(ms):
    """ms: block mean squares (+1e-20 already added)."""
    ungated = -0.691 + 10.0 * np.log10(np.mean(ms))
    keep = (-0.691 + 10.0 * np.log10(ms)) >= ungated - 10.0
    if np.any(keep):
        ms = ms[keep]
    return float(-0.691 + 10.0 * np.log10(np.mean(ms)))


def gain_match(ref, sr, x, mode="LUFS-I", max_gain_db=12.0):
    """-> (matched [C,N] float32, gain_db, ref_level, in_level); x already at the reference rate.  :362-388."""
    if str(mode).upper().startswith("LUFS"):
        rl, il = integrated_lufs(sr, ref), integrated_lufs(sr, x)
    else:
        rl, il = rms_db(ref.mean(axis=0)), rms_db(x.mean(axis=0))
    g = float(np.clip(rl - il, -abs(max_gain_db), abs(max_gain_db)))
    return (x * 10 ** (g / 20.0)).astype(np.float32), g, rl, il


def band_energy_hi_db(x, sr, lo_hz):
    mono = x.mean(axis=0)
    X = np.fft.rfft(mono)
    hi = np.fft.rfftfreq(mono.shape[0], d=1.0 / sr) >= lo_hz
    return 10.0 * math.log10(float(np.sum(np.abs(X[hi]) ** 2)) / float(np.sum(np.abs(X) ** 2) + 1e-20) + 1e-20)


def null_test(A, B, sr, invert_b=True, least_squares_scale=False, compute_corr=True, compute_null_rms=True, compute_null_lufs=True,
              compute_lsd=True, compute_hf_residual=False, n_fft=2048, hop=512, hf_band_hz=8000):
    """-> (null [C,N] float32, metrics dict in the reference's key order).  :420-470."""
    n = min(A.shape[1], B.shape[1])
    A, B = A[:, :n], B[:, :n]
    k = 1.0
    if least_squares_scale:
        a, b = A.mean(axis=0).astype(np.float64), B.mean(axis=0).astype(np.float64)
        k = float(np.dot(a, b) / float(np.dot(b, b) + 1e-20))
        B = (B * k).astype(np.float32)
    if invert_b:
        B = -B
    null = (A + B).astype(np.float32)
    m = {}
    a_m, b_m = A.mean(axis=0), (-B).mean(axis=0)
    if compute_corr:
        am, bm = a_m - np.mean(a_m), b_m - np.mean(b_m)
        m["corr_coef"] = float(np.dot(am, bm) / (np.linalg.norm(am) * np.linalg.norm(bm) + 1e-20))
    if compute_null_rms:
        m["null_rms_dbfs"] = float(rms_db(null.mean(axis=0)))
    if compute_null_lufs:
        m["null_lufs"] = float(integrated_lufs(sr, null))
    if compute_lsd:
        m["lsd_mean_db"], m["lsd_p95_db"] = om.lsd(om.stft_mag(a_m, n_fft, hop), om.stft_mag(b_m, n_fft, hop))
    if compute_hf_residual:
        m["hf_residual_db"] = float(band_energy_hi_db(null, sr, hf_band_hz))
    overs = int(np.sum(np.abs(null) > 1.0))
    m["overshoot_count"], m["clipped_pct"], m["scale_k"] = overs, float(100.0 * overs / null.size), float(k)
    return null, m
